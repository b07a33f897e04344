"""Development aid: per-tile phase stamps of grow_kernel in forest mode on the headline config.  Needs a library built with
MHT_EXTRA_HIPCC_FLAGS=-DMHT_GROW_STAMPS (python -m pymht_amd.build --force); rebuild without it afterwards."""
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
names = ['stage', 'phase1', 'phase2', 'prefix', 'offsets', 'phase4']
raw = len(sys.argv) > 1 and sys.argv[1] == 'raw'      # raw: step through the C ABI without reports -> deferred commits (replay mode)
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    if raw and k >= 10:
        zz = np.ascontiguousarray(z, dtype=np.float32)
        _lib.check(trk._lib.mht_forest_step_host(trk._ctx.handle, zz.ctypes.data_as(C.c_void_p), len(zz)))
        trk.lastScanStats = dict(L=13000)
        trk.toc = dict(Process=0.0)
    else:
        trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 11: continue
    a = np.zeros(32 + 8 * 4000, dtype=np.uint64)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"grow_dbg", a.ctypes.data_as(C.c_void_p), a.nbytes))
    nt = (trk.lastScanStats["L"] + 31) // 32
    ts = a[32:32 + 8 * nt].reshape(nt, 8).astype(np.int64)
    ts = ts[np.abs(ts[:, 0] - np.median(ts[:, 0])) < 5000]      # raw mode: L is not known, drop stale rows of earlier scans
    nt = len(ts)
    t0 = ts[:, 0].min()
    rel = (ts[:, :7] - t0) / 100.0
    d = np.diff(rel, axis=1)
    print('scan %d tiles=%d grow stage %.1f us; kernel span (first tile start -> last tile end) %.1f us' % (k, nt, 1e6 * trk.toc['Process'], rel[:, 6].max()))
    print('   tile start: mean %.1f max %.1f | per-phase mean/max us: %s' % (rel[:, 0].mean(), rel[:, 0].max(),
          '  '.join('%s %.1f/%.1f' % (n, d[:, q].mean(), d[:, q].max()) for q, n in enumerate(names))))
    order = np.argsort(rel[:, 6])
    print('   last tiles to finish:', [(int(i), round(float(rel[i, 0]), 1), round(float(rel[i, 6]), 1)) for i in order[-4:]])
