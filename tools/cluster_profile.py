"""Development aid: phase stamps of the single-workgroup cluster kernel on the headline config."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
sc = make_config('cfg3', seed=5446, n_scans=14)
trk = bench.make_tracker(sc, 0)
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    a = np.zeros(8, dtype=np.int32)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"cluster_dbg", a.ctypes.data_as(C.c_void_p), a.nbytes))
    if k >= 10:
        print('scan %d cluster stage %.1f us | cumulative us: edges-in-LDS %.1f  propagation %.1f (%d iters, E=%d)  clear %.1f  heads %.1f  cl_ptr %.1f  members %.1f' % (
            k, 1e6 * trk.toc['Cluster'], a[0] / 100, a[1] / 100, a[6], a[7], a[2] / 100, a[3] / 100, a[4] / 100, a[5] / 100))
