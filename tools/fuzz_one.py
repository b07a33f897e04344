"""Development aid: one fuzz scenario with similar-state pruning, scan by scan, printing where the leaf sets differ.
python tools/fuzz_one.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fuzz_util import scenario_of
from test_tracker_gpu import make_tracker
from trace_util import make_oracle
from pymht_amd.utils.classDefinitions import MeasurementList
seed = int(sys.argv[1])
sc, N, eta2, desc = scenario_of(seed)
print(desc)
g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, x0=sc["x0"], t0=sc["t0"], accepted=None)
prng = np.random.default_rng(seed + 77)
thr = float(prng.choice([4.0, 6.0, 12.0]))
trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"], pruneThreshold=thr)
g["accepted"] = acc
o = make_oracle(g)
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    on = bool(prng.uniform() < 0.75)
    info = o.add_scan(float(t), z, prune_similar=on, prune_threshold=thr)
    trk.addMeasurementList(MeasurementList(float(t), z), pruneSimilar=on)
    lb, tb = o.leaf_batch(), trk.leafBatch()
    same_n = len(lb["ID"]) == len(tb["ID"])
    print("scan", k, "similar", on, "leaves", len(lb["ID"]), len(tb["ID"]))
    if same_n:
        bad = np.where((lb["ID"] != tb["ID"]) | (lb["meas"] != tb["meas"]) | (np.abs(lb["x"] - tb["x"]).max(axis=1) > 1e-6 * np.maximum(np.abs(lb["x"]).max(axis=1), 1.0)))[0]
        for i in bad[:6]:
            print("   leaf", i, "ID", lb["ID"][i], tb["ID"][i], "meas", lb["meas"][i], tb["meas"][i], "x", lb["x"][i], tb["x"][i], "cnllr", lb["cnllr"][i], tb["cnllr"][i], "flags", tb["flags"][i])
        if len(bad): break
    else:
        break
trk.close()
