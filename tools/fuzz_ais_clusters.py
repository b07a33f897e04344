"""Development: the cluster tables of the tracker and of the oracle on the scans of an AIS fuzz seed where they differ (tests/fuzz_util.py::run_case_ais).
usage: fuzz_ais_clusters.py SEED"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from fuzz_util import scenario_of
from trace_util import make_oracle_ais
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.ais import AisMessage, AisMessageList
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_ais
import mht_oracle as orc
seed = int(sys.argv[1])
sc, N, eta2, desc = scenario_of(seed)
N = min(N, 7)
prng = np.random.default_rng(seed + 1234)
equipped, p_report = float(prng.choice([0.3, 0.6, 1.0])), float(prng.choice([0.4, 0.8]))
ais = make_ais(sc, seed=seed + 5, equipped=equipped, p_report=p_report)
rr = 1.5 * sc["radius"]
ais_init = bool(prng.uniform() < 0.6)
if ais_init and prng.uniform() < 0.5:
    sc["x0"] = sc["x0"][::2].copy()
g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, eta2_ais=9.45, x0=sc["x0"], t0=sc["t0"],
         radar_range=rr, position=np.asarray(sc["centre"], dtype=np.float64), with_initiator=True, accepted=None)
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=eta2, radarRange=rr, position=g["position"], aisAided=True,
              maxTargets=512, maxNodes=1 << 19, maxMeasurements=512)
acc = []
for x in sc["x0"]:
    n0 = trk.nTargets
    trk.initiateTarget(Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized"))
    acc.append(trk.nTargets > n0)
g["accepted"] = acc
o = make_oracle_ais(g)
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    on = bool(prng.uniform() < 0.3)
    msgs = ais[k] if prng.uniform() < 0.85 else []
    info = o.add_scan(float(t), z, prune_similar=on, ais=[orc.AisMessage(m[0], m[1].copy(), m[2], m[3]) for m in msgs], ais_initialization=ais_init)
    trk.addMeasurementList(MeasurementList(float(t), z), AisMessageList([AisMessage(*m) for m in msgs]), aisInitialization=ais_init, pruneSimilar=on)
    oc = [list(map(int, c)) for c in o.clusters]
    tc = [list(map(int, np.asarray(c))) for c in trk.__clusterList__]
    same = oc == tc
    print('scan %d: %d targets, oracle %d clusters (%d ILPs), tracker %d clusters (%d ILPs): %s' % (k, len(o.targets), len(oc), o.n_ilp, len(tc), trk.nOptimSolved, 'same' if same else 'DIFFERENT'))
    if not same:
        so, st = set(map(tuple, oc)), set(map(tuple, tc))
        print('   only oracle :', sorted(so - st))
        print('   only tracker:', sorted(st - so))
        # association sets of the targets involved, from the oracle's leaves
        inv = sorted({t for c in (so ^ st) for t in c})
        lb = o.leaf_batch()
        for ti in inv:
            tid = o.targets[ti].ID if ti < len(o.targets) else None
            m = lb["ID"] == tid
            print('   target index %d (ID %s): %d leaves, leaf measurements %s, leaf mmsi %s' % (ti, tid, int(m.sum()), sorted(set(map(int, lb["meas"][m]))), sorted(set(map(int, lb["mmsi"][m])))))
        if k >= 2: break
trk.close()
