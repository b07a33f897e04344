"""Development: hunt for a rare divergence of the batched group replay (bench.py::run_multi's path): S sectors stepped as two groups on two
streams, every sector also stepped as a single forest; compared every CHECK scans.  usage: group_stress.py S N_SCANS CHECK [REPEATS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from pymht_amd import parallel
from pymht_amd.sectors import SectorGroup
from pymht_amd.utils.scenario import make_config

S, N, CHECK = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
REP = int(sys.argv[4]) if len(sys.argv) > 4 else 1
local = 0
scs, brs = [], []
for q in range(S):
    sq = make_config("cfg3", seed=parallel.sector_seed(5446, 0) + 17 * q, n_scans=N, centre=(0.0, 20000.0 * q), confine=True)
    bq, stq, fq, _, _ = bench.prepass(sq, local)
    scs.append(sq); brs.append(bq)
print("prepass done", flush=True)


def digest(r):
    rep, recs = r.report()
    live = recs[recs["status"] == 0]
    return rep.error, [(int(x["id"]), int(x["sel_meas"]), int(x["n_leaves"])) for x in live]


for rep_i in range(REP):
    NG = int(os.environ.get("STRESS_GROUPS", "2"))
    streams = [torch.cuda.Stream(device=local, priority=(-1 if (q % 2) else 0)) for q in range(NG)]
    rps = []
    for q in range(S):
        with torch.cuda.stream(streams[q % NG]):
            rps.append(bench.Replay(scs[q], brs[q], local))
    grps = [SectorGroup([r.trk for r in rps[gi::NG]]) for gi in range(NG)]
    solo = [bench.Replay(scs[q], brs[q], local) for q in range(S)]
    bad = None
    for k in range(N):
        for gi in range(NG):
            mem = rps[gi::NG]
            grps[gi].step_dev([r.z.data_ptr() + int(r.zoff[k]) * 8 for r in mem], [r.M[k] for r in mem])
        for r in rps:
            r.births_after_step()
        for r in solo:
            r.step()
        if (k + 1) % CHECK == 0 or k == N - 1:
            torch.cuda.synchronize()
            for q in range(S):
                a, b = digest(rps[q]), digest(solo[q])
                if a != b:
                    bad = (k, q, a[0], b[0], len(a[1]), len(b[1]), [x for x, y in zip(a[1], b[1]) if x != y][:5], [y for x, y in zip(a[1], b[1]) if x != y][:5])
                    break
            if bad:
                break
    print("repeat %d: %s" % (rep_i, "ok" if bad is None else "MISMATCH scan %d sector %d err %d/%d live %d/%d group %s solo %s" % bad), flush=True)
    for g in grps:
        g.close()
    for r in rps + solo:
        r.close()
