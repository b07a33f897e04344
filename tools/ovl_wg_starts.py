"""Development aid: when do the workgroups of an overlapping grow launch start, relative to the start of the ILP launch in front of it?
Needs a library built with -DMHT_GROW_STAMPS:  MHT_LIB_VARIANT=.st MHT_EXTRA_HIPCC_FLAGS=-DMHT_GROW_STAMPS python -m pymht_amd.build
then  MHT_LIB_VARIANT=.st python tools/ovl_wg_starts.py"""
import ctypes as C, os, sys
os.environ["MHT_GROW_DEBUG"] = "1"; os.environ["MHT_OVL_FORCE"] = "1"; os.environ["MHT_OVL_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config
sc = make_config('cfg3', seed=5446, n_scans=400, confine=True)
births, stats, final, trk0, _ = bench.prepass(sc, 0)
rp = bench.Replay(sc, births, 0)
def rd(name, k, dt):
    a = np.zeros(k, dtype=dt)
    rp._lib_mod.check(rp.lib.mht_forest_debug_read(rp.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
for r in range(8):
    for _ in range(9 if r else 40):
        rp.step()
    rp._lib_mod.check(rp.lib.mht_synchronize(rp.h))
    w = rd("status2", 16, np.uint64).reshape(2, 8)[:, 2:].astype(np.int64)
    k = rp.k
    new, old = w[k & 1], w[(k - 1) & 1]
    ts = rd("grow_dbg", 32 + 16 * 4000, np.uint64)[32:].reshape(4000, 16).astype(np.int64)
    st = ts[:, 0]; en = ts[:, 7]
    ok = (st > old[1] - 5000) & (st < old[1] + 20000)
    s0 = (st[ok] - old[1]) / 100.0; e0 = (en[ok & (en > 0)] - old[1]) / 100.0
    print('scan %d: ILP k-1 runs 0 .. %.1f; grow k stamp-wg start %.1f; %d stamped workgroups start at p0/p10/p50/p90/p100 = %s; target workgroups end p50/p100 = %s' % (
        k, (old[4] - old[1]) / 100.0, (new[0] - old[1]) / 100.0, ok.sum(), ' '.join('%.1f' % v for v in np.percentile(s0, [0, 10, 50, 90, 100])),
        ' '.join('%.1f' % v for v in np.percentile(e0, [50, 100])) if len(e0) else '-'))
