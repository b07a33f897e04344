"""Development: why an AIS fuzz seed fails (tests/fuzz_util.py::run_case_ais): which part of the failing check, and by how much.  usage: fuzz_ais_why.py SEED..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import fuzz_util
from fuzz_util import scenario_of
from trace_util import make_oracle_ais
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.ais import AisMessage, AisMessageList
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_ais
import mht_oracle as orc

for seed in [int(a) for a in sys.argv[1:]]:
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
    print("seed", seed, desc, "init", ais_init)
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        on = bool(prng.uniform() < 0.3)
        msgs = ais[k] if prng.uniform() < 0.85 else []
        info = o.add_scan(float(t), z, prune_similar=on, ais=[orc.AisMessage(m[0], m[1].copy(), m[2], m[3]) for m in msgs], ais_initialization=ais_init)
        trk.addMeasurementList(MeasurementList(float(t), z), AisMessageList([AisMessage(*m) for m in msgs]), aisInitialization=ais_init, pruneSimilar=on)
        lb, tb = o.leaf_batch(), trk.leafBatch()
        same_n = len(lb["ID"]) == len(tb["ID"])
        sets = same_n and np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]) and np.array_equal(lb["mmsi"], tb["mmsi"])
        if sets:
            dx = np.abs(np.asarray(lb["x"], dtype=np.float64).reshape(-1, 4) - np.asarray(tb["x"], dtype=np.float64).reshape(-1, 4))
            dc = np.abs(lb["cnllr"] - tb["cnllr"])
            print("  scan %d: similar %d msgs %d leaves %d: sets equal; max |dx| %.3e (pos) %.3e (vel), max |dcnllr| %.3e  (tolerances %.0e rel + %.0e, score %.0e)" % (
                k, on, len(msgs), len(lb["ID"]), dx[:, :2].max() if len(dx) else 0, dx[:, 2:].max() if len(dx) else 0, dc.max() if len(dc) else 0,
                fuzz_util.AIS_X_REL, fuzz_util.AIS_X_ATOL, fuzz_util.AIS_SCORE_ATOL))
        else:
            print("  scan %d: LEAF SETS DIFFER: oracle %d leaves, device %d" % (k, len(lb["ID"]), len(tb["ID"])))
            if same_n:
                bad = np.where((lb["ID"] != tb["ID"]) | (lb["meas"] != tb["meas"]) | (lb["mmsi"] != tb["mmsi"]))[0]
                for r in bad[:8]:
                    print("     row %d: oracle ID %d meas %d mmsi %d cnllr %.9f | device ID %d meas %d mmsi %d cnllr %.9f" % (r, lb["ID"][r], lb["meas"][r], lb["mmsi"][r], lb["cnllr"][r], tb["ID"][r], tb["meas"][r], tb["mmsi"][r], tb["cnllr"][r]))
            break
    trk.close()
