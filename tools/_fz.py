import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fuzz_util import scenario_of
from test_tracker_gpu import make_tracker, tracker_selected
from trace_util import make_oracle
from pymht_amd.utils.classDefinitions import MeasurementList
seed = int(sys.argv[1])
sc, N, eta2, desc = scenario_of(seed)
print(desc)
g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, x0=sc["x0"], t0=sc["t0"], accepted=None)
trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"])
g["accepted"] = acc
o = make_oracle(g)
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    info = o.add_scan(float(t), z)
    trk.addMeasurementList(MeasurementList(float(t), z))
    st = trk.lastScanStats
    os_, ts = o.selected(), tracker_selected(trk)
    same = np.array_equal(os_["meas"], ts["meas"])
    print('scan %d: L=%d ilps=%d branched=%d same selection=%s  sum cnllr ours %.9f oracle %.9f' % (k, st["L"], st["ilp"], st["branched"], same, ts["cnllr"].sum(), os_["cnllr"].sum()))
    if not same:
        d = np.where(os_["meas"] != ts["meas"])[0]
        print('   differing targets', d[:20], 'ours', ts["meas"][d][:20], 'oracle', os_["meas"][d][:20])
        hits = ts["meas"][ts["meas"] > 0]
        print('   our selection uses a measurement twice:', len(np.unique(hits)) != len(hits))
        break
import ctypes as C
from pymht_amd import _lib
def rd(name, k, dt=np.int32):
    a = np.zeros(k, dtype=dt)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a
cnt = rd('cl_counts', 8); nC, nM = cnt[0], cnt[1]
ptr = rd('cl_ptr', nC + 1); ml = rd('multi_list', nM); nT0 = int(ptr[nC])
tch = rd('tchild', nT0 + 1); tce = rd('tcend', nT0 + 1); mem = rd('cl_members', nT0)
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'ilp2'), exist_ok=True)
for c in ml:
    K = ptr[c + 1] - ptr[c]
    sizes, costs, cols = [], [], []
    for m in mem[ptr[c]:ptr[c + 1]]:
        b, e = int(tch[m]), int(tce[m])
        sizes.append(e - b)
        costs.append(rd('cost@%d' % (8 * b), e - b, np.float64))
        cols.append(rd('path@%d' % (32 * b), 8 * (e - b)).reshape(-1, 8))
    if sum(sizes) > 1500:
        np.savez(os.path.join(ROOT, 'gpurun_out', 'ilp2', 'ilp_s%d_k%d_c%d.npz' % (seed, k, c)), sizes=np.array(sizes), cost=np.concatenate(costs), cols=np.concatenate(cols), us=0.0, iters=0, status=0, nodes=0)
        print('dumped cluster', c, 'K', K, 'nH', sum(sizes))
