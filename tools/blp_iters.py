"""Development aid: distribution of dual-ascent iterations / per-cluster time of the forest ILPs on the headline config."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sc = make_config('cfg3', seed=5446, n_scans=n)
trk = bench.make_tracker(sc, 0)
def rd(name, n, dt=np.int32):
    a = np.zeros(n, dtype=dt)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
worst_t, worst_it, stage, all_it, all_t = [], [], [], [], []
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 12: continue
    cnt = rd('cl_counts', 8); nC, nM = cnt[0], cnt[1]
    ml = rd('multi_list', nM); it = rd('cl_iters', nC); tm = rd('cl_time', 8 * nC).reshape(-1, 8); st = rd('cl_status', nC)
    tt = tm[ml, 1] / 100.0
    worst_t.append(tt.max()); worst_it.append(it[ml][np.argmax(tt)]); stage.append(1e6 * trk.toc['Optim'])
    all_it.extend(it[ml].tolist()); all_t.extend(tt.tolist())
all_it, all_t = np.array(all_it), np.array(all_t)
print('ILPs: %d over %d scans; iterations: median %d  p90 %d  p99 %d  max %d' % (len(all_it), len(stage), np.median(all_it), np.percentile(all_it, 90), np.percentile(all_it, 99), all_it.max()))
print('per-ILP us: median %.1f p90 %.1f p99 %.1f max %.1f' % (np.median(all_t), np.percentile(all_t, 90), np.percentile(all_t, 99), all_t.max()))
print('per-scan worst ILP us: mean %.1f median %.1f max %.1f ; its iterations: median %d mean %.1f ; Optim stage mean %.1f us' % (
    np.mean(worst_t), np.median(worst_t), np.max(worst_t), np.median(worst_it), np.mean(worst_it), np.mean(stage)))
h = np.bincount(np.minimum(all_it, 40))
print('iteration histogram (0..40+):', h.tolist())
# detail of the slowest ILP of the last scans
ptr = rd('cl_ptr', nC + 1); mem = rd('cl_members', ptr[nC]); tch = rd('tchild', len(trk._tbl["id"]) + 64)
order = np.argsort(-tt)[:6]
for i in order:
    c = ml[i]
    K = ptr[c + 1] - ptr[c]
    nH = sum(tch[m + 1] - tch[m] for m in mem[ptr[c]:ptr[c + 1]])
    print('  last scan: %.1f us  K=%d nH=%d it=%d status=%d  setup %.1f  stamps %s' % (tt[i], K, nH, it[c], st[c], tm[c, 0] / 100.0, (tm[c, 2:6] / 100.0).round(1).tolist()))
print('effective shader clock during the ILPs (s_memtime ticks per us): median %.0f MHz' % np.median(tm[ml, 6] / (tm[ml, 1] / 100.0)))
