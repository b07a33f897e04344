"""Development: one AIS fuzz seed, verbose on the first mismatching scan.  usage: fuzz_ais_debug.py SEED"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import fuzz_util
from trace_util import make_oracle_ais
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.ais import AisMessage, AisMessageList
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_ais
import mht_oracle as orc

seed = int(sys.argv[1])
sc, N, eta2, desc = fuzz_util.scenario_of(seed)
N = min(N, 7)
prng = np.random.default_rng(seed + 1234)
equipped, p_report = float(prng.choice([0.3, 0.6, 1.0])), float(prng.choice([0.4, 0.8]))
ais = make_ais(sc, seed=seed + 5, equipped=equipped, p_report=p_report)
rr = 1.5 * sc["radius"]
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
print(desc)
for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
    on = bool(prng.uniform() < 0.3)
    msgs = ais[k] if prng.uniform() < 0.85 else []
    info = o.add_scan(float(t), z, prune_similar=on, ais=[orc.AisMessage(m[0], m[1].copy(), m[2], m[3]) for m in msgs])
    trk.addMeasurementList(MeasurementList(float(t), z), AisMessageList([AisMessage(*m) for m in msgs]), aisInitialization=False, pruneSimilar=on)
    nodes = list(trk.getTrackNodes())
    os_ = o.selected()
    t_meas = np.array([-1 if n.measurementNumber is None else n.measurementNumber for n in nodes], dtype=np.int64)
    t_mmsi = np.array([0 if n.mmsi is None else n.mmsi for n in nodes], dtype=np.int64)
    t_cn = np.array([float(n.cumulativeNLLR) for n in nodes])
    lb, tb = o.leaf_batch(), trk.leafBatch()
    print("scan", k, "similar", on, "msgs", len(msgs), "L", info["L"], "fused", info["n_fused"], "ilp", o.n_ilp, trk.nOptimSolved)
    bad = np.where((os_["meas"] != t_meas) | (os_["mmsi"] != t_mmsi))[0] if len(t_meas) == len(os_["meas"]) else None
    if bad is None or len(bad):
        print("  SELECTION differs at targets", bad)
        for i in (bad if bad is not None else []):
            print("   target", i, "ID", os_["ID"][i], "oracle meas/mmsi/cnllr", os_["meas"][i], os_["mmsi"][i], repr(os_["cnllr"][i]), " device", t_meas[i], t_mmsi[i], repr(t_cn[i]))
            rows = np.where(lb["target"] == i)[0]
            cl = [c for c in o.clusters if i in c][0]
            print("    cluster", cl, " leaves of the target:", len(rows))
            order = np.argsort(lb["cnllr"][rows])[:6]
            for r in rows[order]:
                print("     oracle leaf meas %3d mmsi %10d cnllr %.12f | device meas %3d mmsi %10d cnllr %.12f" % (lb["meas"][r], lb["mmsi"][r], lb["cnllr"][r], tb["meas"][r] if r < len(tb["meas"]) else -9, tb["mmsi"][r] if r < len(tb["mmsi"]) else -9, tb["cnllr"][r] if r < len(tb["cnllr"]) else 0))
        break
    if len(lb["ID"]) != len(tb["ID"]) or not (np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]) and np.array_equal(lb["mmsi"], tb["mmsi"])):
        print("  LEAVES differ: oracle", len(lb["ID"]), "device", len(tb["ID"]))
        n = min(len(lb["ID"]), len(tb["ID"]))
        d = np.where((lb["ID"][:n] != tb["ID"][:n]) | (lb["meas"][:n] != tb["meas"][:n]) | (lb["mmsi"][:n] != tb["mmsi"][:n]))[0]
        print("  first differing rows", d[:10])
        for r in d[:6]:
            print("   row", r, "oracle ID/meas/mmsi", lb["ID"][r], lb["meas"][r], lb["mmsi"][r], lb["cnllr"][r], " device", tb["ID"][r], tb["meas"][r], tb["mmsi"][r], tb["cnllr"][r])
        if len(d):
            r0 = max(int(d[0]) - 3, 0)
            for r in range(r0, min(r0 + 12, n)):
                print("    ", r, "o", lb["target"][r], lb["ID"][r], lb["meas"][r], lb["mmsi"][r], "%.6f" % lb["cnllr"][r], "| d", tb["target"][r], tb["ID"][r], tb["meas"][r], tb["mmsi"][r], "%.6f" % tb["cnllr"][r])
        break
    if len(lb["x"]):
        scale = np.maximum(np.abs(lb["x"]).max(axis=1, keepdims=True), 1.0)
        rel = np.abs(lb["x"] - tb["x"]) / scale
        r = int(np.argmax(rel.max(axis=1)))
        print("   worst leaf state: rel %.3e  row %d target %d meas %d mmsi %d  oracle %s device %s  cnllr diff %.2e" % (rel.max(), r, lb["target"][r], lb["meas"][r], lb["mmsi"][r],
              np.array2string(lb["x"][r], precision=6), np.array2string(tb["x"][r], precision=6), np.abs(lb["cnllr"] - tb["cnllr"]).max()))
        Pd = np.abs(lb["P"] - tb["P"]).reshape(len(lb["P"]), -1).max(axis=1) / np.abs(lb["P"]).reshape(len(lb["P"]), -1).max(axis=1)
        print("   worst covariance rel %.3e" % Pd.max())
trk.close()
