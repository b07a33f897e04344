"""Development aid: per-workgroup phase stamps of fgrow_kernel on the headline config.  Needs a library built with
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
trk = bench.make_tracker(sc, 0, maxTargets=int(os.environ.get("MHT_PROF_MAXT", "640")), deviceTiming=True)
names = ['rt1', 'phase1', 'cands', 'pairs', 'counts+alloc', 'edges(w1)', 'emit']
COLS = [0, 1, 2, 3, 4, 5, 6, 7]
raw = len(sys.argv) > 1 and sys.argv[1] == 'raw'      # raw: step through the C ABI without reports -> deferred commits (replay mode)
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    if raw and k >= 10:
        zz = np.ascontiguousarray(z, dtype=np.float32)
        _lib.check(trk._lib.mht_forest_step_host(trk._ctx.handle, zz.ctypes.data_as(C.c_void_p), len(zz)))
    else:
        trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 11: continue
    a = np.zeros(32 + 16 * 4000, dtype=np.uint64)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"grow_dbg", a.ctypes.data_as(C.c_void_p), a.nbytes))
    ts = a[32:].reshape(4000, 16).astype(np.int64)
    w0 = ts[:, :8]
    main = w0[(w0[:, 7] > 0) & (w0[:, 0] > 0) & (w0[:, 5] > 0)]
    main = main[np.abs(main[:, 0] - np.median(main[:, 0])) < 5000]
    chain = ts[(ts[:, 1] > 0) & (ts[:, 7] == 0) & (ts[:, 0] > 0)][:, :2]
    chain = chain[np.abs(chain[:, 0] - np.median(main[:, 0])) < 5000]
    t0 = min(main[:, 0].min(), chain[:, 0].min() if len(chain) else 1 << 62)
    rel = (main - t0) / 100.0
    d = np.diff(rel[:, COLS], axis=1)
    print('scan %d: %d target workgroups, span (first start -> last end) %.1f us; start mean %.1f max %.1f' % (k, len(main), rel[:, 7].max(), rel[:, 0].mean(), rel[:, 0].max()))
    print('   per-phase mean/max us: ' + '  '.join('%s %.1f/%.1f' % (n, d[:, q].mean(), d[:, q].max()) for q, n in enumerate(names)))
    wx = ts[:, 8:16]
    wx = wx[(wx[:, 6] > 0) & (wx[:, 1] > 0) & (np.abs(wx[:, 1] - np.median(main[:, 0])) < 5000)]
    if len(wx):
        dx = np.diff(wx[:, 1:7], axis=1) / 100.0
        print('   emit detail (thread 0): ' + '  '.join('%s %.2f/%.2f' % (nm, dx[:, q].mean(), dx[:, q].max()) for q, nm in enumerate(['ldsbatch+decode', 'path recs', 'score', 'x stores', 'rest'])))
    if len(chain):
        rc = (chain - t0) / 100.0
        print('   %d chain workgroups: start mean %.1f max %.1f, end mean %.1f max %.1f, duration mean %.1f max %.1f' % (
            len(rc), rc[:, 0].mean(), rc[:, 0].max(), rc[:, 1].mean(), rc[:, 1].max(), (rc[:, 1] - rc[:, 0]).mean(), (rc[:, 1] - rc[:, 0]).max()))
    # stale rows of earlier scans would pass for this one: clear
    z0 = np.zeros_like(a)
