"""Development aid: the slowest ILPs of a long headline stream (per-cluster times from the forest's debug arrays)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5446
sc = make_config('cfg3', seed=seed, n_scans=n, confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=True)
dump_dir = sys.argv[3] if len(sys.argv) > 3 else None
if dump_dir: os.makedirs(dump_dir, exist_ok=True)
def rd(name, k, dt=np.int32):
    a = np.zeros(k, dtype=dt)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
rows = []
stage = []
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 20: continue
    stage.append(1e6 * trk.toc['Optim'])
    cnt = rd('cl_counts', 8); nC, nM = cnt[0], cnt[1]
    ptr = rd('cl_ptr', nC + 1); ml = rd('multi_list', nM); it = rd('cl_iters', nC); tm = rd('cl_time', 8 * nC).reshape(-1, 8); st = rd('cl_status', nC); nd = rd('cl_nodes', nC)
    nT0 = int(ptr[nC]); tch = rd('tchild', nT0 + 1); tce = rd('tcend', nT0 + 1); mem = rd('cl_members', nT0)
    for c in ml:
        K = ptr[c + 1] - ptr[c]
        nH = int(sum(tce[m] - tch[m] for m in mem[ptr[c]:ptr[c + 1]]))
        rows.append((tm[c, 1] / 100.0, k, int(K), nH, int(it[c]), int(st[c]), int(nd[c]), tm[c, 0] / 100.0, tm[c, 7] / 100.0, tuple(int(v) // 100 for v in tm[c, 2:5])))
        if dump_dir and tm[c, 1] / 100.0 > 60.0:      # the instance itself, for offline experiments (tools/blp_small.py)
            sizes, costs, cols = [], [], []
            for m in mem[ptr[c]:ptr[c + 1]]:
                b, e = int(tch[m]), int(tce[m])
                sizes.append(e - b)
                costs.append(rd('cost@%d' % (8 * b), e - b, np.float64))
                cols.append(rd('path@%d' % (32 * b), 8 * (e - b)).reshape(-1, 8))
            np.savez(os.path.join(dump_dir, 'ilp_s%d_k%d_c%d.npz' % (seed, k, c)), sizes=np.array(sizes), cost=np.concatenate(costs), cols=np.concatenate(cols),
                     us=tm[c, 1] / 100.0, iters=int(it[c]), status=int(st[c]), nodes=int(nd[c]))
rows.sort(reverse=True)
t = np.array([r[0] for r in rows])
print('%d ILPs over %d scans: mean %.1f us, p50 %.1f, p90 %.1f, p99 %.1f, max %.1f; Optim stage mean %.1f p90 %.1f max %.1f' % (
    len(rows), n - 20, t.mean(), np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max(), np.mean(stage), np.percentile(stage, 90), np.max(stage)))
print('slowest: (us, scan, K, nH, iters, status[1 cert 2 bb 3 limit], nodes, setup us, search us)')
for r in rows[:12]: print('  ', tuple(round(float(x), 1) if isinstance(x, (float, np.floating)) else x for x in r))
per_scan = {}
for r in rows: per_scan[r[1]] = max(per_scan.get(r[1], 0), r[0])
m = np.array(list(per_scan.values()))
print('slowest ILP of a scan: mean %.1f p50 %.1f p90 %.1f max %.1f' % (m.mean(), np.percentile(m, 50), np.percentile(m, 90), m.max()))
# who is the slowest ILP of a scan?
worst = {}
for r in rows:
    if r[1] not in worst or r[0] > worst[r[1]][0]: worst[r[1]] = r
import collections
by_it = collections.Counter(min(w[4], 9) for w in worst.values())
by_k = collections.Counter(min(w[2], 12) for w in worst.values())
print('slowest ILP of a scan by dual rounds:', sorted(by_it.items()), ' by K:', sorted(by_k.items()))
for it in sorted(by_it):
    sel = [w for w in worst.values() if min(w[4], 9) == it]
    print('  rounds %d: %3d scans, mean %.1f us, setup %.1f us, columns %.0f, K %.1f' % (it, len(sel), np.mean([w[0] for w in sel]), np.mean([w[7] for w in sel]), np.mean([w[3] for w in sel]), np.mean([w[2] for w in sel])))
allr = np.array([(r[0], r[2], r[3], r[4], r[7]) for r in rows])
for it in range(0, 6):
    m = allr[:, 3] == it
    if m.any(): print('  all ILPs with %d rounds: %5d, mean %.1f us (setup %.1f), columns %.0f' % (it, m.sum(), allr[m, 0].mean(), allr[m, 4].mean(), allr[m, 2].mean()))
