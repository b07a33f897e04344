"""Development aid: where the host time of the streamed drop-in path goes (cProfile around bench.prepass).
    python tools/api_profile.py [cfg3|cfg2] [n_scans]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pymht_amd.utils.scenario import make_config

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sc = make_config(name, seed=5446, n_scans=n, confine=True)
bench.prepass(sc, 0, 40)      # (warm: library, allocator, first-use paths)
pr = cProfile.Profile()
pr.enable()
_, stats, _, api_s, _ = bench.prepass(sc, 0, 40)
pr.disable()
print("api %.0f scans/s under the profiler (%.1f us per scan)" % (1.0 / api_s, 1e6 * api_s))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
