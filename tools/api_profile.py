"""Development aid: where the host time of the drop-in Tracker.addMeasurementList() goes on the headline config."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
sc = make_config('cfg3', seed=5446, n_scans=n, confine=True)
trk = bench.make_tracker(sc, 0)
scans = [MeasurementList(float(t), z) for z, t in zip(sc['scans'], sc['times'])]
for s in scans[:20]:
    trk.addMeasurementList(s)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for s in scans[20:]:
    trk.addMeasurementList(s)
pr.disable()
dt = time.perf_counter() - t0
print('API path (profiled): %.1f us/scan' % (1e6 * dt / (n - 20)))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
