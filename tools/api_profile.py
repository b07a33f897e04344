"""Development aid: where the time of the drop-in API path goes on the host (cProfile over a streamed headline run) and how long
the device needs for the same scans (stream idle time = host bound).   python tools/api_profile.py [n_scans]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 416
sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
lists = [MeasurementList(float(t), z) for t, z in zip(sc["times"], sc["scans"])]
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
for sl in lists[:16]:
    trk.addMeasurementList(sl)
trk.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for sl in lists[16:]:
    trk.addMeasurementList(sl)
t1 = time.perf_counter()
trk.synchronize()
pr.disable()
t2 = time.perf_counter()
print("%d scans: issue loop %.1f us/scan, drain %.1f us total -> %.0f scans/s (profiler on)" % (n - 16, 1e6 * (t1 - t0) / (n - 16), 1e6 * (t2 - t1), (n - 16) / (t2 - t0)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
trk.close()

# ---- second pass: time the two C calls of a scan separately (issue vs wait for the report) ---------------------------------------
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
acc = {"scan": 0.0, "get": 0.0}
lib = trk._lib
class Timed:
    def __init__(self, fn, key): self.fn, self.key = fn, key
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); acc[self.key] += time.perf_counter() - t; return r
class LibProxy:
    def __init__(self, lib):
        self._lib = lib
        self.mht_forest_scan = Timed(lib.mht_forest_scan, "scan")
        self.mht_forest_report_get = Timed(lib.mht_forest_report_get, "get")
    def __getattr__(self, k): return getattr(self._lib, k)
trk._lib = LibProxy(lib)
for sl in lists[:16]:
    trk.addMeasurementList(sl)
trk.synchronize()
acc["scan"] = acc["get"] = 0.0
t0 = time.perf_counter()
for sl in lists[16:]:
    trk.addMeasurementList(sl)
trk.synchronize()
dt = time.perf_counter() - t0
m = n - 16
print("per scan: total %.1f us, mht_forest_scan %.1f us, mht_forest_report_get %.1f us (wait for the device included), Python %.1f us" % (
    1e6 * dt / m, 1e6 * acc["scan"] / m, 1e6 * acc["get"] / m, 1e6 * (dt - acc["scan"] - acc["get"]) / m))
trk.close()
