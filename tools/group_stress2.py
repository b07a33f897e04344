"""Development: rare divergence hunt, part 2 -- replays compared with their prepass finals only (nothing fetched in between).
usage: group_stress2.py MODE S N_SCANS REPEATS    MODE: g1 / g2 (one / two groups), solo1 (single forests, one stream), solo2 (single forests on two streams)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from pymht_amd import parallel
from pymht_amd.sectors import SectorGroup
from pymht_amd.utils.scenario import make_config

MODE, S, N, REP = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
local = 0
scs, brs, fins = [], [], []
for q in range(S):
    sq = make_config("cfg3", seed=parallel.sector_seed(5446, 0) + 17 * q, n_scans=N, centre=(0.0, 20000.0 * q), confine=True)
    bq, stq, fq, _, _ = bench.prepass(sq, local)
    scs.append(sq); brs.append(bq); fins.append(fq)
nbad = 0
for rep_i in range(REP):
    NS = 2 if MODE in ("g2", "solo2") else 1
    streams = [torch.cuda.Stream(device=local, priority=(-1 if (q % 2) else 0)) for q in range(NS)]
    rps = []
    for q in range(S):
        with torch.cuda.stream(streams[q % NS]):
            rps.append(bench.Replay(scs[q], brs[q], local))
    grps = [SectorGroup([r.trk for r in rps[gi::NS]]) for gi in range(NS)] if MODE.startswith("g") else []
    for k in range(N):
        if grps:
            for gi in range(NS):
                mem = rps[gi::NS]
                grps[gi].step_dev([r.z.data_ptr() + int(r.zoff[k]) * 8 for r in mem], [r.M[k] for r in mem])
            for r in rps:
                r.births_after_step()
        else:
            for r in rps:
                r.step()
    torch.cuda.synchronize()
    bad = []
    for q, r in enumerate(rps):
        rep, recs = r.report()
        got = [(int(x["id"]), int(x["sel_meas"])) for x in recs if int(x["status"]) == 0]
        if got != fins[q] or rep.error:
            bad.append((q, rep.error, len(got), len(fins[q])))
    if bad:
        nbad += 1
        print("repeat %d: MISMATCH %s" % (rep_i, bad), flush=True)
    for g in grps:
        g.close()
    for r in rps:
        r.close()
print("mode %s S=%d N=%d: %d of %d repeats diverged" % (MODE, S, N, nbad, REP))
