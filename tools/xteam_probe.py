import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch, time
from test_sharded_gpu import _rd
from pymht_amd.parallel import ClusterShardedTracker
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_scenario
sc = make_scenario(T=66, radius=201.0, lambda_phi=1.5e-4, n_scans=7, P_d=0.73, period=2.5, seed=5494)
def mk():
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=9.21, useInitiator=False, maxTargets=512, maxNodes=1 << 18)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    return trk
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 2
solo = mk()
parts = [ClusterShardedTracker(mk(), shards, i, exchange=lambda t: None) for i in range(shards)]
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    sl = MeasurementList(float(t), z)
    solo.addMeasurementList(sl); solo._ctx.synchronize()
    for p in parts: p.begin(sl)
    for p in parts: p.trk._ctx.synchronize()
    n_team = int(_rd(parts[0].trk, "cl_counts", 8)[5])
    lst = _rd(parts[0].trk, "team_list", 8)
    for q in range(n_team):
        c = int(lst[q])
        print("scan", k, "team cluster", c, "solo status/nodes/time_us", _rd(solo, "cl_status", 512)[c], _rd(solo, "cl_nodes", 512)[c], _rd(solo, "cl_time", 4096).reshape(-1, 8)[c, 1] / 100.0,
              "| shards:", [(int(_rd(p.trk, "cl_status", 512)[c]), int(_rd(p.trk, "cl_nodes", 512)[c]), _rd(p.trk, "cl_time", 4096).reshape(-1, 8)[c, 1] / 100.0) for p in parts])
    merged = torch.stack([p.sel_rel for p in parts]).max(dim=0).values
    for p in parts:
        p.sel_rel.copy_(merged); p.end()
