"""Development aid: phases of blp_uf_kernel's workgroups (union-find prologue, setup, solve, epilogue) on the headline stream.
MHT_GROW_DEBUG=1 MHT_BLP_STAMPS=1 python tools/blp_uf_profile.py [n_scans]"""
import ctypes as C, os, sys
os.environ["MHT_GROW_DEBUG"] = "1"; os.environ["MHT_BLP_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd import _lib
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sc = make_config('cfg3', seed=5446, n_scans=n, confine=True)
trk = bench.make_tracker(sc, 0, deviceTiming=True, useInitiator=False)
def rd(name, k, dt=np.int32):
    a = np.zeros(k, dtype=dt)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
uf = os.environ.get('MHT_NO_UF') != '1'
names = ['loads', 'chase', 'scan', 'tables', 'members', '->body', '->call', '->solve', 'setup', 'solve', 'epilogue', '->end'] if uf else ['->body', '->call', '->solve', 'setup', 'solve', 'epilogue', '->end']
acc, spans, starts, ends, drains, agree = [], [], [], [], [], []
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k < 25: continue
    a = rd('grow_dbg', 2 * (32 + 16 * 4000)).view(np.uint64)
    ts = a[32:].reshape(4000, 16).astype(np.int64)
    cnt = rd('cl_counts', 8); nM = int(cnt[1])
    w = ts[:nM]
    w = w[(w[:, 0] > 0) & (w[:, 11] > 0)]
    if not len(w): continue
    cols = [0, 1, 2, 3, 4, 7, 12, 13, 8, 9, 10, 11, 15] if uf else [0, 12, 13, 8, 9, 10, 11, 15]
    d = np.diff(w[:, cols], axis=1) / 100.0
    acc.append(d.mean(axis=0))
    allw = ts[(ts[:, 0] > 0) & (ts[:, 15] > 0)]
    allw = allw[np.abs(allw[:, 0] - np.median(w[:, 0])) < 20000]      # (rows of workgroups beyond this scan's grid are stale)
    t0 = allw[:, 0].min()
    xcc = (allw[:, 14] & 7); phys = allw[:, 14] >> 8
    drains.append([((allw[xcc == x, 15].max() - t0) / 100.0 if (xcc == x).any() else np.nan) for x in range(8)])
    agree.append(float(np.mean((phys & 7) == xcc)))
    ends.append(np.percentile((allw[:, 15] - t0) / 100.0, [5, 25, 50, 75, 95, 100]))
    spans.append(((allw[:, 15].max() - t0) / 100.0, (w[:, 11].max() - t0) / 100.0, (allw[:, 0].max() - t0) / 100.0, (w[:, 8] - w[:, 0]).mean() / 100.0))
m = np.mean(acc, axis=0)
print('workgroups with an ILP, mean us per phase: ' + '  '.join('%s %.2f' % (nm, v) for nm, v in zip(names, m)))
s = np.mean(spans, axis=0)
print('launch: first start -> last end %.1f us, -> last ILP end %.1f, last workgroup start %.1f; entry -> first solve %.1f' % tuple(s))
print('workgroup end times (us from the first start), percentiles 5/25/50/75/95/100: ' + ' '.join('%.1f' % v for v in np.mean(ends, axis=0)))
print('per-XCD drain time (us): ' + ' '.join('%.1f' % v for v in np.nanmean(drains, axis=0)) + ';  workgroups with XCC_ID == blockIdx %% 8: %.0f %%' % (100 * np.mean(agree)))
