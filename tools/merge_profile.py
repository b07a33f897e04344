"""Development aid: the one-launch-per-scan kernel (blp_grow_kernel), workgroup by workgroup -- when a workgroup's ILP share is done, which
grow roles it takes and how long each lasts.  Needs a library built with -DMHT_GROW_STAMPS:
  MHT_LIB_VARIANT=.stamps MHT_EXTRA_HIPCC_FLAGS=-DMHT_GROW_STAMPS python -c "from pymht_amd.build import build_library; build_library(force=True)"
  MHT_LIB_VARIANT=.stamps python tools/merge_profile.py"""
import ctypes as C, os, sys
os.environ["MHT_GROW_DEBUG"] = "1"; os.environ["MHT_OVL_FORCE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
sc = make_config("cfg3", seed=5446, n_scans=440, confine=True)
births, stats, final, trk0, _ = bench.prepass(sc, 0)
rp = bench.Replay(sc, births, 0)
h, lib = rp.h, rp.lib
rows = []
for r in range(24):
    for _ in range(40 if r == 0 else 9):
        rp.step()
    _lib.check(lib.mht_synchronize(h))
    g = np.zeros(32 + 16 * 4000, dtype=np.uint64)
    _lib.check(lib.mht_forest_debug_read(h, b"grow_dbg", g.ctypes.data_as(C.c_void_p), g.nbytes))
    ts = g[32:].reshape(4000, 16)[1000:1512].astype(np.int64)
    ts = ts[ts[:, 0] > 0]
    if len(ts) < 200:
        continue
    t0 = ts[:, 0].min()
    ent = (ts[:, 0] - t0) / 100.0; ilp = (ts[:, 1] - t0) / 100.0; ex = (ts[:, 15] - t0) / 100.0
    nro = (ts[:, 14] >> 60) & 15
    durs_t, durs_c, gaps = [], [], []
    n_main = None
    for w in range(len(ts)):
        prev_end = ts[w, 1]
        for i in range(min(int(nro[w]), 6)):
            role = (ts[w, 14] >> (10 * i)) & 1023
            st, en = ts[w, 2 + 2 * i], ts[w, 3 + 2 * i]
            gaps.append((st - prev_end) / 100.0)
            prev_end = en
            if role == 0:
                continue
            (durs_t if role <= 500 else durs_c).append((en - st) / 100.0)
    last_role_end = np.array([ts[w, 3 + 2 * (min(int(nro[w]), 6) - 1)] if nro[w] > 0 else ts[w, 1] for w in range(len(ts))])
    rows.append([ent.max(), np.percentile(ilp, 25), np.median(ilp), np.percentile(ilp, 95), ilp.max(), np.median(nro), nro.max(),
                 np.median(durs_t), np.percentile(durs_t, 95), np.median(durs_c) if durs_c else 0.0, np.median(gaps), np.percentile(gaps, 95),
                 ((last_role_end - t0) / 100.0).max(), np.median(ex), ex.max()])
    if r < 3:
        allts = g[32:].reshape(4000, 16).astype(np.int64)
        o = np.argsort(ex)[-6:]
        for w in o:
            row = ts[w]
            rl = [(int((row[14] >> (10 * i)) & 1023), (row[2 + 2 * i] - t0) / 100.0, (row[3 + 2 * i] - t0) / 100.0) for i in range(min(int(nro[w]), 6))]
            # the workgroup's own block row (FG_STAMP of its LAST target role, wavefront 0): phases relative to t0
            bid = int(np.nonzero((g[32:].reshape(4000, 16)[1000:1512, 0].astype(np.int64) == row[0]))[0][0])
            ph = [(v - t0) / 100.0 if v > 0 else -1 for v in allts[bid, :8]]
            print("   wg %d: ilp %.1f roles %s exit %.1f | FG_STAMP of its last target role %s" % (bid, ilp[w], ["%d: %.1f-%.1f" % x for x in rl], ex[w], ["%.1f" % v for v in ph]))
    if r < 4:
        o = np.argsort(ex)[-4:]
        print("round %d: %d workgroups; last to exit: " % (r, len(ts)) + "; ".join(
            "wg %d ilp %.1f roles %s exit %.1f" % (w, ilp[w], [int((ts[w, 14] >> (10 * i)) & 1023) for i in range(min(int(nro[w]), 6))], ex[w]) for w in o))
a = np.median(np.array(rows), axis=0)
print("blp_grow_kernel, us from the first workgroup's entry: last entry %.1f | ILP share done p25 %.1f p50 %.1f p95 %.1f max %.1f | roles per workgroup p50 %d max %d | "
      "target role p50 %.1f p95 %.1f, chain role p50 %.1f | ticket gap p50 %.1f p95 %.1f | last role ends %.1f | exit p50 %.1f max %.1f" % tuple(a))
