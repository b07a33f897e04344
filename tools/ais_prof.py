import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.ais import AisMessage, AisMessageList
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config, make_ais
sc = make_config("cfg3", seed=5446, n_scans=40, confine=True)
ais = make_ais(sc, seed=11, equipped=0.5, p_report=0.7)
trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=5, eta2=5.99, radarRange=float(sc["radius"]) * 1.2, position=np.asarray(sc["centre"], dtype=float),
              aisAided=True, maxTargets=2048, maxNodes=1 << 19, maxMeasurements=1024)
trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
def run(k0, k1):
    for k in range(k0, k1):
        trk.addMeasurementList(MeasurementList(float(sc["times"][k]), sc["scans"][k]), AisMessageList([AisMessage(*m) for m in ais[k]]))
    trk.synchronize()
run(0, 12)
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); run(12, 40); dt = time.perf_counter() - t0; pr.disable()
print("%.2f ms per scan" % (1e3 * dt / 28), {k: round(v * 1e6) for k, v in trk.toc.items() if isinstance(v, float)})
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
