"""Development aid: streaming throughput of the drop-in API (Tracker.addMeasurementList, results read after the last scan)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sc = make_config('cfg3', seed=5446, n_scans=n, confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=False)
scans = [MeasurementList(float(t), z) for z, t in zip(sc['scans'], sc['times'])]
for s in scans[:40]:
    trk.addMeasurementList(s)
trk.synchronize()
t0 = time.perf_counter()
for s in scans[40:]:
    trk.addMeasurementList(s)
trk.synchronize()
dt = time.perf_counter() - t0
print('API streaming: %.1f us/scan = %.0f scans/s; targets %d' % (1e6 * dt / (n - 40), (n - 40) / dt, trk.nTargets))
