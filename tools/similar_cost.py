"""Development aid: what similar-state pruning (addMeasurementList(pruneSimilar=True)) costs / saves on the headline stream through the
drop-in API.   python tools/similar_cost.py [n_scans]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
lists = [MeasurementList(float(t), z) for t, z in zip(sc["times"], sc["scans"])]
for on in (False, True):
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    for sl in lists[:16]:
        trk.addMeasurementList(sl, pruneSimilar=on)
    trk.synchronize()
    t0 = time.perf_counter()
    for sl in lists[16:]:
        trk.addMeasurementList(sl, pruneSimilar=on)
    trk.synchronize()
    dt = time.perf_counter() - t0
    st = trk.lastScanStats
    print("pruneSimilar=%s: %.0f scans/s through the API, last scan L=%d G=%d ilp=%d targets=%d" % (
        on, (n - 16) / dt, st["L"], st["G"], st["ilp"], trk.nTargets))
    trk.close()
