"""Development aid: per-phase wall-clock of grow_kernel in forest mode on the headline config (MHT_GROW_DEBUG=1)."""
import ctypes as C, os, sys
os.environ["MHT_GROW_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
sc = make_config('cfg3', seed=5446, n_scans=14)
trk = bench.make_tracker(sc, 0)
prev = np.zeros(16, dtype=np.uint64)
names = ['stage z/off', 'phase1 predict', 'phase2 fan-out', 'phase3 prefix', 'offsets', 'phase4 children']
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    a = np.zeros(16, dtype=np.uint64)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"grow_dbg", a.ctypes.data_as(C.c_void_p), a.nbytes))
    d = a[:8].astype(np.int64) - prev[:8].astype(np.int64); prev = a
    if k >= 9:
        n = max(int(d[6]), 1)
        print('scan %2d L=%d tiles=%d  grow stage %.1f us | mean us per tile: %s | sum %.1f' % (k, trk.lastScanStats['L'], n, 1e6 * trk.toc['Process'],
              '  '.join('%s %.2f' % (nm, d[q] / n / 100.0) for q, nm in enumerate(names)), d[:6].sum() / n / 100.0))
print('max us per phase over the run:', '  '.join('%s %.1f' % (nm, a[8 + q] / 100.0) for q, nm in enumerate(names)))
