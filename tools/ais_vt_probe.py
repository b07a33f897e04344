"""Development: value-table consumption of an AIS-aided stream at the headline size (ids handed out after every scan, generation switches)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.ais import AisMessage, AisMessageList
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config, make_ais
equipped = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sc = make_config("cfg3", seed=5446, n_scans=n_scans, confine=True)
ais = make_ais(sc, seed=11, equipped=equipped, p_report=0.7)
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=5, eta2=5.99, radarRange=float(sc["radius"]) * 1.2, position=np.asarray(sc["centre"], dtype=float),
              aisAided=True, maxTargets=2048, maxNodes=1 << 19, maxMeasurements=1024)
trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    try:
        trk.addMeasurementList(MeasurementList(float(t), z), AisMessageList([AisMessage(*m) for m in ais[k]]), aisInitialization=False)
        trk.synchronize()
    except Exception as e:
        print("scan", k, "FAILED:", str(e)[:160]); break
    v, r = np.zeros(1, np.uint32), np.zeros(1, np.int32)
    trk._lib.mht_forest_debug_read(trk._ctx.handle, b"vcount", v.ctypes.data_as(C.c_void_p), 4)
    trk._lib.mht_forest_debug_read(trk._ctx.handle, b"vt_rebuilds", r.ctypes.data_as(C.c_void_p), 4)
    st = trk.lastScanStats
    print("scan %2d: L %6d G %6d msgs %3d | value ids %8d rebuilds %d" % (k, st["L"], st["G"], len(ais[k]), int(v[0]), int(r[0])), flush=True)
trk.close()
