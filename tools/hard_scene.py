"""Development: one fuzz scenario (tests/fuzz_util.scenario_of) through the drop-in Tracker, per-scan ILP stage time from the device stamps.
usage: hard_scene.py SEED"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fuzz_util import scenario_of
from test_tracker_gpu import make_tracker
from pymht_amd.utils.classDefinitions import MeasurementList

seed = int(sys.argv[1])
sc, N, eta2, desc = scenario_of(seed)
print(desc)
trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"], blpNodeLimit=1 << 22)
worst = 0.0
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    st = trk.lastScanStats
    ms = trk.toc['Optim'] * 1e3
    worst = max(worst, ms)
    sizes = sorted((len(c) for c in trk.__clusterList__), reverse=True)[:3]
    print("scan %2d  L=%5d  ilp=%3d branched=%d limit=%d  largest clusters %s  optim %.2f ms" % (k, st["L"], st["ilp"], st["branched"], st.get("limit", 0), sizes, ms), flush=True)
    if ms > 20:
        import ctypes as C
        T = trk.nTargets + 64
        def rd(name, n):
            a = np.zeros(n, dtype=np.int32)
            trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), C.c_int64(n * 4))
            return a
        cnts = rd("cl_counts", 8); ml = rd("multi_list", T)[:cnts[1]]
        stt, it, nd, tm = rd("cl_status", T), rd("cl_iters", T), rd("cl_nodes", T), rd("cl_time", 8 * T).reshape(-1, 8)
        tch, tce, ptr, mem = rd("tchild", T + 1), rd("tcend", T + 1), rd("cl_ptr", T + 1), rd("cl_members", T)
        for c in ml:
            cols = sum(int(tce[t] - tch[t]) for t in mem[ptr[c]:ptr[c + 1]])
            print("     cluster %d: K=%d columns=%d status=%d iters=%d nodes=%d  time(10ns ticks): setup %d total %d stamps %s" % (
                c, ptr[c + 1] - ptr[c], cols, stt[c], it[c], nd[c], tm[c][0], tm[c][1], tm[c][2:6].tolist()))
print("worst optim stage %.2f ms" % worst)
trk.close()
