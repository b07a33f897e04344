import os, sys
ROOT='/root/repo'
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
trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"])
g["accepted"] = acc
o = make_oracle(g)
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    info = o.add_scan(float(t), z)
    trk.addMeasurementList(MeasurementList(float(t), z))
    oi = [r.ID for r in o.targets]; ti = [r.ID for r in trk.__targetList__]
    print("scan", k, "M", len(z), "targets oracle", len(oi), "tracker", len(ti), "same", oi == ti, "| L", info["L"], trk.lastScanStats["L"])
    if oi != ti:
        ox = {r.ID: np.asarray(r.x if hasattr(r,'x') else r.x_0)[:4] for r in o.targets}
        tx = {r.ID: np.asarray(r.x_0)[:4] for r in trk.__targetList__}
        print("  oracle ids", oi[-12:]); print("  tracker ids", ti[-12:])
        new_o = [r for r in o.targets if r.ID in info["new_ids"]]
        so = {tuple(np.round(np.asarray(r.x, float)[:2], 2)): r for r in new_o}
        st = {tuple(np.round(np.asarray(r.x_0, float)[:2], 2)) for r in trk.__targetList__}
        print("  new in oracle", len(new_o), "dead", info["dead"])
        leaves = [(root.ID, l) for root in o.targets for l in root.leaves() if root.ID not in info["new_ids"]]
        for key, r in so.items():
            if key not in st:
                d = sorted((float(np.linalg.norm(np.asarray(l.x[:2], float) - np.asarray(r.x[:2], float))), rid) for rid, l in leaves)[:2]
                db = sorted((float(np.linalg.norm(np.asarray(q.x[:2], float) - np.asarray(r.x[:2], float))), q.ID) for q in new_o if q.ID != r.ID)[:2]
                print("   missing in tracker: id", r.ID, key, "meas", r.meas, "nearest old leaf", d, "nearest births", db)
        print("  merge threshold", o.merge_threshold)
        break
trk.close()
