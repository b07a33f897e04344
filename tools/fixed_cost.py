"""Development aid: fixed cost of a timed region of the replay (sync + K scans + sync) as a function of K."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pymht_amd.utils.scenario import make_config
sc = make_config("cfg3", seed=5446, n_scans=21 + 400, confine=True)
births, stats, final, api_s, init_s = bench.prepass(sc, 0, 21)
rp = bench.Replay(sc, births, 0)
for _ in range(21):
    rp.step()
torch.cuda.synchronize()
for K in (1, 2, 4, 8, 16, 32, 64, 128):
    if rp.k + K > len(rp.M): break
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        rp.step()
    t_issue = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("K=%3d: %.1f us total, %.1f us per scan, host issue loop %.1f us (%.1f per scan)" % (K, 1e6 * (t1 - t0), 1e6 * (t1 - t0) / K, 1e6 * (t_issue - t0), 1e6 * (t_issue - t0) / K))
rp.close()
