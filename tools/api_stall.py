"""Development aid: streams scans through the drop-in API the way bench.py's prepasses do (main sector with a synchronize in the middle, further
sectors without) and prints every call that took more than 1 ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import parallel
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config
import torch
local = 0
for q in range(4):
    n = 41 if q == 0 else 229
    sc = make_config("cfg3", seed=parallel.sector_seed(5446, 0) + 17 * q, n_scans=n, centre=(parallel.sector_centre(0)[0], 20000.0 * q), confine=True)
    trk = bench.make_tracker(sc, local, deviceTiming=False, logScanStats=True)
    lists = [MeasurementList(float(t), z) for z, t in zip(sc["scans"], sc["times"])]
    t_all = time.perf_counter()
    for k, sl in enumerate(lists):
        t = time.perf_counter()
        trk.addMeasurementList(sl)
        dt = time.perf_counter() - t
        if dt > 1e-3: print("sector %d scan %d: call %.1f ms" % (q, k + 1, dt * 1e3))
        if (q == 0 and k + 1 == 21) or k + 1 == n:
            t = time.perf_counter(); trk.synchronize(); dt = time.perf_counter() - t
            if dt > 1e-3: print("sector %d synchronize behind scan %d: %.1f ms" % (q, k + 1, dt * 1e3))
    print("sector %d: %.1f us/scan" % (q, 1e6 * (time.perf_counter() - t_all) / n))
    trk.close()
print("done")
