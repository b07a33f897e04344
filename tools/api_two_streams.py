"""Development: two drop-in trackers on two non-blocking torch streams, scans interleaved and streamed (nothing looked at until the end),
against the same trackers run alone on the default stream.   usage: api_two_streams.py N_SCANS REPEATS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config

n, reps = int(sys.argv[1]), int(sys.argv[2])
scs = [make_config("cfg3", seed=s, n_scans=n, confine=True) for s in (5446, 77)]


def mk(sc):
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, logScanStats=True)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    return trk


def final(trk):
    nodes = list(trk.getTrackNodes())
    fin = [(r.ID, nd.measurementNumber, tuple(np.asarray(nd.x_0).tolist())) for r, nd in zip(trk.__targetList__, nodes)]
    log = [{k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in s.items()} for s in trk.scanStatsLog]
    return log, fin


ref = []
for sc in scs:
    t = mk(sc)
    for z, tm in zip(sc["scans"], sc["times"]):
        t.addMeasurementList(MeasurementList(float(tm), z))
    ref.append(final(t))
    t.close()
bad = 0
for r in range(reps):
    streams = [torch.cuda.Stream(device=0, priority=-(q % 2)) for q in range(2)]
    trks = []
    for q in range(2):
        with torch.cuda.stream(streams[q]):
            trks.append(mk(scs[q]))
    for k in range(n):
        for q in range(2):
            trks[q].addMeasurementList(MeasurementList(float(scs[q]["times"][k]), scs[q]["scans"][k]))
    got = [final(t) for t in trks]
    ok = got == ref
    bad += 0 if ok else 1
    print("repeat %d: %s" % (r, "identical" if ok else "DIFFERENT"), flush=True)
    for t in trks:
        t.close()
print("%d of %d repeats differ" % (bad, reps))
