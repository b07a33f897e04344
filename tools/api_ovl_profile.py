"""Development aid: the STREAMED drop-in path's grow launch (fgrow_adm_kernel), workgroup by workgroup, relative to the ILP launch it overlaps
-- the streamed counterpart of tools/grow_ovl_profile.py (which drives the replay).  Needs a library built with -DMHT_GROW_STAMPS:
  MHT_LIB_VARIANT=.stamps MHT_EXTRA_HIPCC_FLAGS=-DMHT_GROW_STAMPS python -c "from pymht_amd.build import build_library; build_library(force=True)"
  MHT_LIB_VARIANT=.stamps python tools/api_ovl_profile.py [replay]"""
import ctypes as C, os, sys
os.environ["MHT_GROW_DEBUG"] = "1"; os.environ["MHT_OVL_STAMPS"] = "1"; os.environ["MHT_OVL_FORCE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymht_amd import _lib
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config
n = 440
REPLAY = "replay" in sys.argv      # the same analysis of the replay's launches (bench.Replay), for comparison
sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
if REPLAY:
    os.environ["MHT_OVL_FORCE"] = "1"
    import bench
    births, stats, final, trk0, _ = bench.prepass(sc, 0)
    rp = bench.Replay(sc, births, 0)
    h, lib = rp.h, rp.lib
    step = rp.step
    sync = lambda: _lib.check(lib.mht_synchronize(h))
else:
    lists = [MeasurementList(float(t), z) for t, z in zip(sc["times"], sc["scans"])]
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    h, lib = trk._ctx.handle, trk._lib
    step = lambda: trk.addMeasurementList(lists[k])
    sync = trk.synchronize
out = []
k = 0
for r in range(30):
    for _ in range(40 if r == 0 else 11):
        step(); k += 1
    sync()
    a = np.zeros(16, dtype=np.uint64)
    _lib.check(lib.mht_forest_debug_read(h, b"status2", a.ctypes.data_as(C.c_void_p), a.nbytes))
    w = a.reshape(2, 8)[:, 2:].astype(np.int64)
    new, old = w[k & 1], w[(k - 1) & 1]
    g = np.zeros(32 + 16 * 4000, dtype=np.uint64)
    _lib.check(lib.mht_forest_debug_read(h, b"grow_dbg", g.ctypes.data_as(C.c_void_p), g.nbytes))
    ts_all = g[32:].reshape(4000, 16).astype(np.int64)
    ts = ts_all[:, :8]
    t_ilp0, t_ilp1 = old[1], old[4]
    sel = (ts[:, 0] > t_ilp0) & (ts[:, 7] > ts[:, 0]) & (ts[:, 7] < t_ilp0 + 20000)
    m = ts[sel]
    if len(m) < 100:
        continue
    idx = np.nonzero(sel)[0]
    st = (m[:, 0] - t_ilp1) / 100.0; rec = (m[:, 1] - t_ilp1) / 100.0; en = (m[:, 7] - t_ilp1) / 100.0
    work = (m[:, 7] - m[:, 1]) / 100.0
    ph = np.diff(m, axis=1) / 100.0      # phase durations 0->1 ... 6->7
    last = np.argsort(en)[-5:]
    out.append([(t_ilp1 - t_ilp0) / 100.0, (new[0] - t_ilp1) / 100.0, np.percentile(st, 5), np.median(st), np.percentile(st, 95), st.max(),
                np.median(rec), np.percentile(rec, 95), rec.max(), np.median(work), np.percentile(work, 95), work.max(), np.median(en), np.percentile(en, 95), en.max(),
                (new[5] - t_ilp1) / 100.0, (new[1] - t_ilp1) / 100.0] + list(np.median(ph, axis=0)))
    if r < 6:
        print('round %d: %d target workgroups; the five that end last: ' % (r, len(m)) +
              '; '.join('wg %d start %.1f rec %.1f work %.1f end %.1f' % (idx[i], st[i], rec[i], work[i], en[i]) for i in last))
        o = np.argsort(idx); q = np.array_split(o, 10)
        print('   start by grid decile: ' + ' '.join('%.1f' % np.median(st[x]) for x in q) + '   end: ' + ' '.join('%.1f' % np.max(en[x]) for x in q))
a = np.median(np.array(out), axis=0)
print(('REPLAY' if REPLAY else 'STREAMED') + ', us relative to the END of the overlapped ILP launch (its length %.1f): grow launch first stamp %.1f | target workgroups start p5 %.1f p50 %.1f p95 %.1f max %.1f | '
      'have their record p50 %.1f p95 %.1f max %.1f | work behind the record p50 %.1f p95 %.1f max %.1f | end p50 %.1f p95 %.1f max %.1f | t5 %.1f | next ILP launch starts %.1f' % tuple(a[:17]))
print('   median phase durations of the target workgroups (stamps 0..7): ' + ' '.join('%.1f' % v for v in a[17:]))
