"""Development aid: what the ILP launch's clusters look like on a stream -- sizes (targets, columns), dual rounds, device time per
cluster (BlpArgs::cl_time) -- and which cluster each scan's launch ends with.  Reads the forest's tables after every scan of an
untimed replay (every read synchronises).
    python tools/ilp_dist.py [cfg3|cfg2|cfg5] [n_scans]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sc = make_config(name, seed=5446, n_scans=n, confine=True)
births = [[] for _ in sc["scans"]]
if name != "cfg5":
    births, _, _, _, _ = bench.prepass(sc, 0)
rp = bench.Replay(sc, births, 0)


def rd(nm, k, dt=np.int32):
    a = np.zeros(max(int(k), 1), dtype=dt)
    rp._lib_mod.check(rp.lib.mht_forest_debug_read(rp.h, nm.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a


rows, worst = [], []
for k in range(n):
    rp.step()
    if k < 20:
        continue
    rp._lib_mod.check(rp.lib.mht_synchronize(rp.h))
    cnt = rd("cl_counts", 8)
    nC, nM, nS = int(cnt[0]), int(cnt[1]), int(cnt[2])
    if nM == 0:
        continue
    ptr, ml, it, st = rd("cl_ptr", nC + 1), rd("multi_list", nM), rd("cl_iters", nC), rd("cl_status", nC)
    tm = rd("cl_time", 8 * nC).reshape(-1, 8)
    nT0 = int(ptr[nC])
    tch, tce, mem = rd("tchild", nT0 + 1), rd("tcend", nT0 + 1), rd("cl_members", nT0)
    per = []
    for c in ml:
        members = mem[ptr[c]:ptr[c + 1]]
        cols = int(sum(int(tce[m]) - int(tch[m]) for m in members))
        per.append((len(members), cols, int(it[c]), int(st[c]), tm[c, 1] / 100.0, tm[c, 0] / 100.0, tm[c, 5] / 100.0, 1.0 if tm[c, 6] == 0 else 0.0))
    per = np.array(per)
    rows.append(per)
    w = per[np.argmax(per[:, 4])]
    worst.append(w)
    print("scan %3d: %3d multi %3d single | max K %3d max cols %5d | slowest: K %2d cols %4d rounds %d status %d  %.1f us (setup %.1f, solved at %.1f)"
          % (k, nM, nS, per[:, 0].max(), per[:, 1].max(), w[0], w[1], w[2], w[3], w[4], w[5], w[6]))
a = np.concatenate(rows)
print("\n%d clusters in %d scans" % (len(a), len(rows)))
for nm, col in (("targets", 0), ("columns", 1), ("rounds", 2), ("us", 4)):
    print("%-8s percentiles 50/75/90/95/99/100: %s" % (nm, " ".join("%.1f" % v for v in np.percentile(a[:, col], [50, 75, 90, 95, 99, 100]))))
for cap_h, cap_k in ((256, 8), (512, 8), (512, 16), (768, 16), (1024, 16), (1024, 32), (1536, 32), (2048, 64)):
    fit = (a[:, 1] <= cap_h) & (a[:, 0] <= cap_k)
    per_scan = np.mean([np.all((r[:, 1] <= cap_h) & (r[:, 0] <= cap_k)) for r in rows])
    print("tier %4d columns / %2d targets: %.2f %% of the clusters fit, %.0f %% of the scans have nothing beyond it" % (cap_h, cap_k, 100 * fit.mean(), 100 * per_scan))
print("rounds histogram:", np.bincount(a[:, 2].astype(int)))
print("status histogram:", np.bincount(a[:, 3].astype(int)))
w = np.array(worst)
print("slowest cluster per scan: mean %.1f us; K mean %.1f, columns mean %.0f, rounds mean %.2f" % (w[:, 4].mean(), w[:, 0].mean(), w[:, 1].mean(), w[:, 2].mean()))
for r in range(int(a[:, 2].max()) + 1):
    m = a[:, 2] == r
    if m.any():
        print("rounds == %d: %5d clusters, us mean %.1f p95 %.1f max %.1f; columns mean %.0f" % (r, m.sum(), a[m, 4].mean(), np.percentile(a[m, 4], 95), a[m, 4].max(), a[m, 1].mean()))
