import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config
from pymht_amd import _lib
sc = make_config("cfg3", seed=5446, n_scans=60, confine=True)
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
for z, t in zip(sc["scans"], sc["times"]):
    trk.addMeasurementList(MeasurementList(float(t), z))
trk.synchronize()
a = np.zeros(64 * 16, np.int32)
_lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"commit_log", a.ctypes.data_as(C.c_void_p), a.nbytes))
a = a.reshape(64, 16)
rows = a[(a[:, 0] > 10) & (a[:, 0] < 59)]
for r in sorted(a[a[:, 0] > 0], key=lambda r: -r[13])[:6]:
    print("scan %d: commit %.1f us, admission + flag + report head %.1f us" % (r[0], r[12] / 100.0, r[13] / 100.0))
print("workgroup 0 of fgrow_adm_kernel over %d scans: commit %.2f us (max %.2f), admission + flag + report patch %.2f us (max %.2f), end of workgroup 0 after its start %.2f us" % (
    len(rows), rows[:, 12].mean() / 100, rows[:, 12].max() / 100, rows[:, 13].mean() / 100, rows[:, 13].max() / 100, (rows[:, 12] + rows[:, 13]).mean() / 100))
