"""Development aid: per-scan workload of the bench stream (L, G, ILPs, births, slowest ILP iterations) in blocks of 20 scans."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 221
sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
births, stats, final, api_s, init_s = bench.prepass(sc, 0, 21)
st = np.array(stats, dtype=np.float64)
nb = np.array([len(b) for b in births])
for b0 in range(1, n, 20):
    s = st[b0:b0 + 20]
    print("scans %3d..%3d: L %.0f G %.0f ilp %.1f branched %.2f iters_max %.1f targets %.0f births/scan %.2f scans with births %d" % (
        b0, b0 + len(s) - 1, s[:, 0].mean(), s[:, 1].mean(), s[:, 3].mean(), s[:, 4].mean(), s[:, 5].mean(), s[:, 6].mean(), nb[b0:b0 + 20].mean(), (nb[b0:b0 + 20] > 0).sum()))
