"""Development aid: hunt for dense scenarios whose ILPs branch for long (teams off); prints the slowest."""
import os, sys, time
os.environ["MHT_BLP_NO_TEAMS"] = os.environ.get("MHT_BLP_NO_TEAMS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.scenario import make_scenario
from pymht_amd.utils.classDefinitions import MeasurementList
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 240.0
rows = []
t_start = time.time()
for s in range(seed0, seed0 + n):
    if time.time() - t_start > budget: break
    rng = np.random.default_rng(s)
    T = int(rng.integers(45, 71)); radius = float(rng.uniform(150, 300)); N = int(rng.integers(4, 8)); lam = float(rng.choice([1e-5, 5e-5, 1.5e-4]))
    if os.environ.get("HUNT_SMALL"):      # clusters that fit LDS (<= 2048 columns): short windows
        N = int(rng.integers(2, 5)); T = int(rng.integers(55, 71)); radius = float(rng.uniform(180, 260))
    P_d = float(rng.uniform(0.6, 0.95)); eta2 = float(rng.choice([5.99, 9.21])); period = float(rng.choice([1.0, 2.5]))
    sc = make_scenario(T=T, radius=radius, lambda_phi=lam, n_scans=9, P_d=P_d, period=period, seed=s)
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=eta2, useInitiator=False, maxTargets=512, maxNodes=1 << 19, blpTimeLimit=3.0)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    worst, br, L = 0.0, 0, 0
    try:
        for z, t in zip(sc["scans"], sc["times"]):
            trk.addMeasurementList(MeasurementList(float(t), z))
            st = trk.lastScanStats
            worst = max(worst, 1e3 * trk.toc["Optim"]); br += int(st["branched"]); L = max(L, int(st["L"]))
            if L > 60000: break
    except Exception as e:
        print("seed", s, "error", repr(e)[:100])
    trk.close()
    rows.append((worst, s, T, round(radius), N, lam, round(P_d, 2), eta2, period, br, L))
    print("seed %d T=%d r=%.0f N=%d lam=%.1e Pd=%.2f eta2=%.2f dt=%.1f: worst optim %.2f ms branched %d Lmax %d" % (s, T, radius, N, lam, P_d, eta2, period, worst, br, L), flush=True)
rows = [r for r in rows if not os.environ.get('HUNT_SMALL') or r[10] < 2600]
rows.sort(reverse=True)
print("TOP:")
for r in rows[:8]: print(r)
