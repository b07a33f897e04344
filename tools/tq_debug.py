"""Development: a short replay with MHT_TWO_QUEUES=1 (the overlapping grow launch on a second hardware queue), step by step."""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pymht_amd.utils.scenario import make_config
sc = make_config("cfg3", seed=5446, n_scans=40, confine=True)
births = [[] for _ in sc["scans"]]
rp = bench.Replay(sc, births, 0)
for k in range(30):
    rp.step()
    if k < 6 or k % 8 == 0:
        torch.cuda.synchronize()
        print("step", k, "ok", flush=True)
torch.cuda.synchronize()
rep, recs = rp.report()
print("error", rep.error, "targets", rep.n_targets, flush=True)
import ctypes as C
v = np.zeros(1, dtype=np.int32)
rp.lib.mht_forest_debug_read(rp.h, b"tq_launches", v.ctypes.data_as(C.c_void_p), 4)
print("tq launches", int(v[0]))
rp.close()
