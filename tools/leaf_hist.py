"""Development aid: leaves per target over the steady state of the bench stream (who sets fgrow_kernel's duration?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=True)
mx, over, grow, tot = [], [], [], []
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    g = 1e6 * trk.toc["Process"]
    nl = np.asarray(trk._tbl_["n_leaves"])
    if k >= 20:
        mx.append(nl.max()); over.append(int((nl > 96).sum())); grow.append(g); tot.append(nl.sum())
mx, over, grow = np.array(mx), np.array(over), np.array(grow)
print("leaves/target after a scan (= input of the next): max per scan mean %.0f p50 %.0f max %d; scans with a target > 96 leaves: %d of %d; > 64: n/a" % (mx.mean(), np.median(mx), mx.max(), (over > 0).sum(), len(over)))
# grow time of scan k+1 vs max leaves after scan k
a, b = mx[:-1], grow[1:]
for lo, hi in ((0, 64), (64, 96), (96, 128), (128, 192), (192, 10000)):
    m = (a > lo) & (a <= hi)
    if m.any(): print("  max leaves in (%d, %d]: %d scans, grow stage mean %.1f us" % (lo, hi, m.sum(), b[m].mean()))
trk.close()
