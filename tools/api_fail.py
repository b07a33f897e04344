"""Development aid: the bench's API prepass; on a device error prints the timeout bits of the two status words."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import parallel, _lib
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config
sc = make_config("cfg3", seed=parallel.sector_seed(5446, 0), n_scans=41, centre=parallel.sector_centre(0), confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=False, logScanStats=True)
lists = [MeasurementList(float(t), z) for z, t in zip(sc["scans"], sc["times"])]
try:
    for k, sl in enumerate(lists):
        trk.addMeasurementList(sl)
    trk.synchronize()
    print("ok")
except Exception as e:
    print("failed at call", k + 1, repr(e)[:200])
    st2 = np.zeros(16, dtype=np.uint64)
    trk._lib.mht_forest_debug_read(trk._ctx.handle, b"status2", st2.ctypes.data_as(C.c_void_p), 128)
    print("status2:", [hex(int(x)) for x in st2[:2]], [hex(int(x)) for x in st2[8:10]])
    dbg = np.zeros(8, dtype=np.uint64)
    trk._lib.mht_forest_debug_read(trk._ctx.handle, b"init_dbg", dbg.ctypes.data_as(C.c_void_p), 64)
    print("init_dbg:", dbg.tolist())
