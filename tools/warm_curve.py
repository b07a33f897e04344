"""Development aid: stage times by scan number over the start of the bench stream (how long the warm-up transient lasts)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=True)
rows = []
for z, t in zip(sc["scans"], sc["times"]):
    trk.addMeasurementList(MeasurementList(float(t), z))
    rows.append((1e6 * trk.toc["Process"], 1e6 * trk.toc["Cluster"], 1e6 * trk.toc["Optim"]))
r = np.array(rows)
for b0 in range(0, n, 10):
    s = r[b0:b0 + 10]
    print("scans %3d..%3d: grow %.1f cluster %.1f ilp %.1f us" % (b0 + 1, b0 + len(s), s[:, 0].mean(), s[:, 1].mean(), s[:, 2].mean()))
trk.close()
