"""Development aid: per-cluster ILP timing of the forest on the headline config (uses mht_forest_debug_read)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
sc = make_config('cfg3', seed=5446, n_scans=24, confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=True)
def rd(name, n, dt=np.int32):
    a = np.zeros(n, dtype=dt)
    bench._lib_check = None
    from pymht_amd import _lib
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 18: continue
    cnt = rd('cl_counts', 8); nC, nM = cnt[0], cnt[1]
    ptr = rd('cl_ptr', nC + 1); ml = rd('multi_list', nM); it = rd('cl_iters', nC); tm = rd('cl_time', 8 * nC).reshape(-1, 8); st = rd('cl_status', nC)
    nT0 = int(mem_n) if (mem_n := ptr[nC]) else 0
    tch = rd('tchild', nT0 + 1); tce = rd('tcend', nT0 + 1); mem = rd('cl_members', ptr[nC])
    rows = []
    for c in ml:
        K = ptr[c + 1] - ptr[c]
        nH = sum(tce[m] - tch[m] for m in mem[ptr[c]:ptr[c + 1]])
        rows.append((tm[c, 1] / 100.0, tm[c, 0] / 100.0, K, nH, it[c], st[c], tuple(np.round(tm[c, 2:6] / 100.0, 1))))
    rows.sort(reverse=True)
    tot = np.array([r[0] for r in rows]); su = np.array([r[1] for r in rows])
    mhz = np.median([tm[c, 6] / (tm[c, 1] / 100.0) for c in ml])
    print('   effective shader clock (clock64 ticks per us of wall_clock64): median %.0f MHz' % mhz)
    print('scan %2d: %3d ILPs  total us: max %.1f mean %.1f | setup us: max %.1f mean %.1f | Optim stage %.1f us | worst: %s' % (
        k, nM, tot.max(), tot.mean(), su.max(), su.mean(), 1e6 * trk.toc['Optim'],
        ' '.join('(%.0fus K=%d nH=%d it=%d setup %.1f A %.1f B %.1f C %.1f loop %.1f)' % ((r[0], r[2], r[3], r[4], r[1]) + r[6]) for r in rows[:4])))
    import collections
    byK = collections.defaultdict(list)
    for r in rows: byK[min(r[2], 9)].append(r[0])
    print('      by K (targets): ' + '  '.join('K=%d%s n=%d mean %.1f max %.1f' % (kk, '+' if kk == 9 else '', len(v), np.mean(v), max(v)) for kk, v in sorted(byK.items())))
