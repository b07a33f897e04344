"""Development aid: what a device that OWNS a share of the headline sector's targets (ownership sharding: VERDICT round 5, missing #1) would spend per
scan -- the same stream at the same target and clutter density with 1/2, 1/4, 1/8 of the targets (radius scaled by sqrt), timed like bench.py's replay.
The per-scan time of the share is the floor of a rank's time BEFORE any exchange between the ranks: scans/s(share) / scans/s(whole) bounds the
speed-up ownership sharding can reach on this path.     python tools/ownership_bound.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config

K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W = 40
base = None
for div in (1, 2, 4, 8):
    sc = make_config("cfg3", seed=5446, n_scans=W + K, confine=True, T=500 // div, radius=5000.0 / np.sqrt(div))
    births, stats, _, api_s, _ = bench.prepass(sc, 0, W)
    rp = bench.Replay(sc, births, 0)
    for _ in range(W):
        rp.step()
    rp._lib_mod.check(rp.lib.mht_synchronize(rp.h))
    t0 = time.perf_counter()
    for _ in range(K):
        rp.step()
    rp._lib_mod.check(rp.lib.mht_synchronize(rp.h))
    dt = (time.perf_counter() - t0) / K
    st = stats[W:W + K]
    base = base or dt
    print("1/%d of the sector: %3d targets, %5.0f leaves, %4.0f measurements, %4.1f ILPs per scan: %6.1f us per scan = %6.0f scans/s (x %.2f of the whole sector's time)"
          % (div, 500 // div, st[:, 0].mean(), st[:, 2].mean(), st[:, 3].mean(), 1e6 * dt, 1.0 / dt, dt / base))
    rp.close() if hasattr(rp, "close") else None
