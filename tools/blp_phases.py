"""Development aid: where an ILP's time goes on the headline stream (setup / iteration 0 / further rounds / epilogue), from the
per-cluster stamps of the forest's debug arrays.   python tools/blp_phases.py [n_scans]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
sc = make_config('cfg3', seed=5446, n_scans=n, confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=True)
def rd(name, k, dt=np.int32):
    a = np.zeros(k, dtype=dt)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
rows = []
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 20: continue
    cnt = rd('cl_counts', 8); nC, nM = cnt[0], cnt[1]
    ml = rd('multi_list', nM); it = rd('cl_iters', nC); tm = rd('cl_time', 8 * nC).reshape(-1, 8) / 100.0
    for c in ml:
        rows.append((it[c], tm[c, 0], tm[c, 4] - tm[c, 0], tm[c, 5] - tm[c, 4], tm[c, 1] - tm[c, 5], tm[c, 1]))
r = np.array(rows)
for its in (0, 1, 2, 3):
    m = r[:, 0] == its
    if m.any():
        print('%d rounds: %6d ILPs: setup %.1f, iteration 0 %.1f, rest of the solve %.1f, epilogue %.1f, total %.1f us' % (
            its, m.sum(), r[m, 1].mean(), r[m, 2].mean(), r[m, 3].mean(), r[m, 4].mean(), r[m, 5].mean()))
