"""Development: what AIS traffic costs through the drop-in API -- the headline scene (500 targets, ~500 measurements per scan, N = 5) with
a share of the ships reporting; scans per second with and without messages, device stage times.  usage: ais_cost.py [equipped] [n_scans]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.ais import AisMessage, AisMessageList
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config, make_ais

equipped = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sc = make_config("cfg3", seed=5446, n_scans=n_scans, confine=True)
ais = make_ais(sc, seed=11, equipped=equipped, p_report=0.7)
for mode in ("radar only (aisAided forest)", "with AIS messages", "with AIS, aisInitialization=False"):
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=5, eta2=5.99, radarRange=float(sc["radius"]) * 1.2, position=np.asarray(sc["centre"], dtype=float),
                  aisAided=True, maxTargets=2048, maxNodes=1 << 19, maxMeasurements=1024)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    t0 = None
    proc = []
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        if k == 12:
            trk.synchronize(); t0 = time.perf_counter()
        msgs = AisMessageList([AisMessage(*m) for m in ais[k]]) if mode != "radar only (aisAided forest)" else AisMessageList()
        trk.addMeasurementList(MeasurementList(float(t), z), msgs, aisInitialization=(mode == "with AIS messages"))
        if k >= 12 and k % 8 == 0:
            proc.append((trk.toc['Process'] * 1e6, trk.toc['Optim'] * 1e6, trk.lastScanStats["L"], len(msgs)))
    trk.synchronize()
    dt = time.perf_counter() - t0
    print("%-36s %7.0f scans/s   (grow stage incl. fusion / ILP stage us, leaves, messages of sampled scans: %s)" % (mode, (n_scans - 12) / dt, [tuple(int(v) for v in p) for p in proc[:4]]))
    trk.close()
