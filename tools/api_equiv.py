"""Development: the drop-in API streamed (commit + admission inside the next grow launch, reports folded two scans late) against the same
scans looked at one by one (post_scan_kernel behind every scan), headline scene.   usage: api_equiv.py N_SCANS SEED [SEED ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config

n = int(sys.argv[1])
bad = 0
for seed in [int(a) for a in sys.argv[2:]]:
    sc = make_config("cfg3", seed=seed, n_scans=n, confine=True)
    runs = []
    for look in (True, False):
        trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, logScanStats=True)
        trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
        for z, t in zip(sc["scans"], sc["times"]):
            trk.addMeasurementList(MeasurementList(float(t), z))
            if look:
                _ = trk.lastScanStats
        nodes = list(trk.getTrackNodes())
        fin = [(r.ID, nd.measurementNumber, tuple(np.asarray(nd.x_0).tolist())) for r, nd in zip(trk.__targetList__, nodes)]
        log = [{k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in s.items()} for s in trk.scanStatsLog]
        runs.append((log, fin, sorted(v.ID for v in trk.__terminatedTargets__)))
        trk.close()
    ok = runs[0] == runs[1]
    births = sum(max(b["nTargets"] - a["nTargets"], 0) for a, b in zip(runs[0][0][:-1], runs[0][0][1:]))
    print("seed %d: %d scans, %d targets at the end, >= %d births, %d terminated: %s" % (seed, n, len(runs[0][1]), births, len(runs[0][2]), "identical" if ok else "DIFFERENT"), flush=True)
    if not ok:
        bad += 1
        for k, (a, b) in enumerate(zip(runs[0][0], runs[1][0])):
            if a != b:
                print("  first differing scan %d: %s | %s" % (k + 1, {q: a[q] for q in a if q != "unused"}, {q: b[q] for q in b if q != "unused"}))
                break
print("%d of %d seeds differ" % (bad, len(sys.argv) - 2))
