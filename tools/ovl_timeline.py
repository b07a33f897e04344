"""Development aid: device wall-clock stamps of consecutive scans of the replay (grow start / end, ILP start / end) -- how far the
grow launch of scan k+1 overlaps the ILP launch of scan k.   MHT_OVL_STAMPS=1 python tools/ovl_timeline.py [n_rounds]"""
import ctypes as C, os, sys
os.environ["MHT_OVL_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymht_amd.utils.scenario import make_config
sc = make_config('cfg3', seed=5446, n_scans=400, confine=True)
births, stats, final, trk0, _ = bench.prepass(sc, 0)
rp = bench.Replay(sc, births, 0)
rows = []
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    for _ in range(9 if r else 40):
        rp.step()
    rp._lib_mod.check(rp.lib.mht_synchronize(rp.h))
    a = np.zeros(2 * 8, dtype=np.uint64)      # DevStatus: 4 ints + t[6]
    rp._lib_mod.check(rp.lib.mht_forest_debug_read(rp.h, b"status2", a.ctypes.data_as(C.c_void_p), a.nbytes))
    w = a.reshape(2, 8)[:, 2:].astype(np.int64)      # t[0..5] by scan parity
    k = rp.k          # scans stepped so far: the newest is scan k (1-based), parity k & 1
    new, old = w[k & 1], w[(k - 1) & 1]
    # old = scan k-1: grow start t0, ILP start t1 (prologue) / t2, ILP end t4, grow end t5; new = scan k
    rows.append([(old[4] - old[1]) / 100.0, (new[0] - old[1]) / 100.0, (new[5] - old[1]) / 100.0, (new[1] - old[1]) / 100.0, (new[1] - new[0]) / 100.0, (old[1] - old[0]) / 100.0])
a = np.array(rows)
print('us from the start of ILP launch k-1:  its end %.1f | grow k starts %.1f, its last target workgroup ends %.1f | ILP launch k starts %.1f   (grow k start -> ILP k start %.1f; grow k-1 start -> ILP k-1 start %.1f)' % tuple(np.median(a, axis=0)))
for r in a[:10]: print('   ', ' '.join('%7.1f' % v for v in r))
