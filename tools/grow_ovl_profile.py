"""Development aid: when the workgroups of an OVERLAPPING grow launch start and end, relative to the ILP launch they overlap (replay of the
headline stream).  Needs a library built with -DMHT_GROW_STAMPS, e.g.
  MHT_LIB_VARIANT=.stamps MHT_EXTRA_HIPCC_FLAGS=-DMHT_GROW_STAMPS python -c "from pymht_amd.build import build_library; build_library(force=True)"
  MHT_LIB_VARIANT=.stamps python tools/grow_ovl_profile.py [rounds]"""
import ctypes as C, os, sys
os.environ["MHT_GROW_DEBUG"] = "1"; os.environ["MHT_OVL_FORCE"] = "1"; os.environ["MHT_OVL_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config
sc = make_config('cfg3', seed=5446, n_scans=400, confine=True)
births, stats, final, trk0, _ = bench.prepass(sc, 0)
rp = bench.Replay(sc, births, 0)
out = []
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    for _ in range(9 if r else 40):
        rp.step()
    rp._lib_mod.check(rp.lib.mht_synchronize(rp.h))
    a = np.zeros(2 * 8, dtype=np.uint64)
    rp._lib_mod.check(rp.lib.mht_forest_debug_read(rp.h, b"status2", a.ctypes.data_as(C.c_void_p), a.nbytes))
    w = a.reshape(2, 8)[:, 2:].astype(np.int64)
    k = rp.k
    new, old = w[k & 1], w[(k - 1) & 1]
    g = np.zeros(32 + 16 * 4000, dtype=np.uint64)
    rp._lib_mod.check(rp.lib.mht_forest_debug_read(rp.h, b"grow_dbg", g.ctypes.data_as(C.c_void_p), g.nbytes))
    ts = g[32:].reshape(4000, 16).astype(np.int64)[:, :8]
    t_ilp0, t_ilp1 = old[1], old[4]
    m = ts[(ts[:, 0] > t_ilp0) & (ts[:, 7] > ts[:, 0]) & (ts[:, 7] < t_ilp0 + 20000)]
    if len(m) < 100: continue
    idx = np.nonzero((ts[:, 0] > t_ilp0) & (ts[:, 7] > ts[:, 0]) & (ts[:, 7] < t_ilp0 + 20000))[0]
    st = (m[:, 0] - t_ilp1) / 100.0; rec = (m[:, 1] - t_ilp1) / 100.0; en = (m[:, 7] - t_ilp1) / 100.0
    dur = en - st; work = (m[:, 7] - m[:, 1]) / 100.0
    last = np.argsort(en)[-5:]
    out.append([(t_ilp1 - t_ilp0) / 100.0, (new[0] - t_ilp1) / 100.0, np.percentile(st, 5), np.median(st), np.percentile(st, 95), st.max(),
                np.median(rec), np.percentile(rec, 95), rec.max(), np.median(work), np.percentile(work, 95), work.max(), np.median(en), np.percentile(en, 95), en.max(),
                (new[5] - t_ilp1) / 100.0, (new[1] - t_ilp1) / 100.0])
    ts_all = g[32:].reshape(4000, 16).astype(np.int64)
    ch = ts_all[(ts_all[:, 1] > t_ilp0) & (ts_all[:, 7] == 0) & (ts_all[:, 0] > t_ilp0) & (ts_all[:, 1] < t_ilp0 + 20000)][:, :2]
    if len(ch) and r < 6:
        cs, ce = (ch[:, 0] - t_ilp1) / 100.0, (ch[:, 1] - t_ilp1) / 100.0
        print('   %d chain workgroups: start p50 %.1f max %.1f, end p50 %.1f max %.1f, duration p50 %.1f max %.1f' % (len(ch), np.median(cs), cs.max(), np.median(ce), ce.max(), np.median(ce - cs), (ce - cs).max()))
    if r < 6:
        print('round %d: %d target workgroups; the five that end last: ' % (r, len(m)) +
              '; '.join('wg %d start %.1f rec %.1f work %.1f end %.1f' % (idx[i], st[i], rec[i], work[i], en[i]) for i in last))
        # start time by position in the grid (deciles)
        o = np.argsort(idx); q = np.array_split(o, 10)
        print('   start by grid decile: ' + ' '.join('%.1f' % np.median(st[x]) for x in q) + '   end: ' + ' '.join('%.1f' % np.max(en[x]) for x in q))
a = np.median(np.array(out), axis=0)
print('us relative to the END of the overlapped ILP launch (its length %.1f): grow launch first stamp %.1f | target workgroups start p5 %.1f p50 %.1f p95 %.1f max %.1f | '
      'have their record p50 %.1f p95 %.1f max %.1f | work behind the record p50 %.1f p95 %.1f max %.1f | end p50 %.1f p95 %.1f max %.1f | t5 %.1f | next ILP launch starts %.1f' % tuple(a))
