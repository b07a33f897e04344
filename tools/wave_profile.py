"""Development aid: per-target phase stamps of the wavefront-per-target grow kernel (one sector, MHT_FG_WAVE_SOLO=1) on the headline
config.  Needs a library built with -DMHT_GROW_STAMPS (MHT_LIB_VARIANT=.stamps)."""
import ctypes as C, os, sys
os.environ["MHT_GROW_DEBUG"] = "1"
os.environ["MHT_FG_WAVE_SOLO"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
sc = make_config('cfg3', seed=5446, n_scans=14)
trk = bench.make_tracker(sc, 0, maxTargets=640, deviceTiming=True)
names = ['rt1', 'phase1', 'cands', 'pairs', 'counts+alloc+edges', 'emit', 'chain']
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 11: continue
    a = np.zeros(32 + 16 * 4000, dtype=np.uint64)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"grow_dbg", a.ctypes.data_as(C.c_void_p), a.nbytes))
    ts = a[32:].reshape(4000, 16).astype(np.int64)[:, :8]
    m = ts[(ts[:, 7] > 0) & (ts[:, 0] > 0)]
    m = m[np.abs(m[:, 0] - np.median(m[:, 0])) < 20000]
    rel = (m - m[:, 0].min()) / 100.0
    d = np.diff(rel, axis=1)
    print('scan %d: %d targets, span %.1f us; start mean %.1f max %.1f; per-target total mean %.1f max %.1f' % (k, len(m), rel[:, 7].max(), rel[:, 0].mean(), rel[:, 0].max(), (rel[:, 7] - rel[:, 0]).mean(), (rel[:, 7] - rel[:, 0]).max()))
    print('   per-phase mean/max us: ' + '  '.join('%s %.1f/%.1f' % (n, d[:, q].mean(), d[:, q].max()) for q, n in enumerate(names)))
    print('   stage times', trk.toc.get('Process'), trk.toc.get('Cluster'), trk.toc.get('Optim'))
