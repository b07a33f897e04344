"""Development probe (container only): does the reference's AIS path run, and what does it produce?"""
import sys, os, logging
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import refimport
logging.disable(logging.CRITICAL)
mods = refimport.load()
from pymht_amd.utils.scenario import make_config
T, pv, Target = mods["tracker"], mods["pv"], mods["pyTarget"].Target
cd = mods["classDefinitions"]
sc = make_config("cfg1", seed=172362)
print(sc.keys(), len(sc["scans"]), sc["period"], sc["N"])
trk = T.Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, radarRange=float(sc["radius"]), position=np.asarray(sc["centre"], dtype=float))
for x in sc["x0"]:
    trk.initiateTarget(Target(sc["t0"], None, x.copy(), pv.P0.copy(), status="preinitialized"))
rng = np.random.default_rng(5)
truth = sc.get("truth")
print("truth" in sc, [k for k in sc.keys()])
x = np.array(sc["x0"], dtype=np.float64)
tprev = sc["t0"]
for k in range(len(sc["scans"])):
    z, t = sc["scans"][k], float(sc["times"][k])
    msgs = []
    for i in range(len(x)):
        if rng.uniform() < 0.6:
            tm = tprev + float(rng.integers(1, 4)) * (t - tprev) / 4.0
            st = x[i].copy(); st[:2] += st[2:] * (tm - sc["t0"])
            st = st + rng.normal(0, 1.0, 4) * np.array([1, 1, 0.1, 0.1])
            msgs.append(cd.AIS_message(time=tm, state=st, mmsi=257000000 + i, highAccuracy=bool(rng.uniform() > 0.5)))
    ais = cd.AisMessageList(msgs)
    trk.addMeasurementList(cd.MeasurementList(t, z), ais, checkIntegrity=True)
    nl = [len(r.getLeafNodes()) for r in trk.__targetList__]
    sel = trk.__trackNodes__
    print(k, "M", len(z), "ais", len(ais), "T", len(trk.__targetList__), "leaves", sum(nl), "mmsi sel", [n.mmsi for n in sel][:12])
    tprev = t
