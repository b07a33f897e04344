import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config
from pymht_amd import _lib
sc = make_config("cfg3", seed=5446, n_scans=300, confine=True)
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k in (20, 100, 299):
        trk.synchronize()
        x = np.zeros((128, 4)); P = np.zeros((128, 16), np.float32); m = np.zeros(128, np.int32)
        nb, npre, nseed = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(trk._lib.mht_initiator_born(trk.initiator.handle, 128, p(x), p(P), p(m), C.byref(nb), C.byref(npre), C.byref(nseed)))
        print("scan", k + 1, "M", len(z), "unused", int(trk.lastScanStats["unused"].sum()), "preliminary tracks", npre.value, "initiators (seeds)", nseed.value, "born", nb.value)
