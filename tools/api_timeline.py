"""Development aid: device wall-clock stamps of consecutive STREAMED scans (drop-in API, device initiator): period, grow start / end, ILP start / end."""
import ctypes as C, os, sys
os.environ["MHT_OVL_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymht_amd import _lib
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config
n = 400
sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
lists = [MeasurementList(float(t), z) for t, z in zip(sc["times"], sc["scans"])]
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
rows = []
k = 0
for r in range(30):
    for _ in range(40 if r == 0 else 11):
        trk.addMeasurementList(lists[k]); k += 1
    trk.synchronize()
    a = np.zeros(16, dtype=np.uint64)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"status2", a.ctypes.data_as(C.c_void_p), a.nbytes))
    w = a.reshape(2, 8)[:, 2:].astype(np.int64)
    new, old = w[k & 1], w[(k - 1) & 1]
    rows.append([(new[0] - old[0]) / 100.0, (old[1] - old[0]) / 100.0, (old[4] - old[1]) / 100.0, (new[0] - old[1]) / 100.0, (new[5] - new[0]) / 100.0])
a = np.array(rows[1:])
print('streamed API, median us: period (grow k-1 start -> grow k start) %.1f | grow k-1 start -> ILP k-1 start %.1f | ILP k-1 duration %.1f | ILP k-1 start -> grow k start %.1f | grow k: start -> last target workgroup end %.1f' % tuple(np.median(a, axis=0)))
u = np.zeros(2, dtype=np.int32)
trk._lib.mht_forest_debug_read(trk._ctx.handle, b"uf_ovl", u.ctypes.data_as(C.c_void_p), 8)
print('scans %d: clustered in the grow launch %d, grow launches any-order %d' % (k, u[0], u[1]))
