"""Development aid: randomised parity runs -- the drop-in Tracker (device forest, host initiator) against the oracle on random
small scenarios (targets, area, clutter, window, detection probability, gate, radar period), scan by scan.
python tools/fuzz_parity.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_tracker_gpu import make_tracker, tracker_selected, states_close, SCORE_ATOL
from trace_util import make_oracle
from pymht_amd.utils.scenario import make_scenario
from pymht_amd.utils.classDefinitions import MeasurementList

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    T = int(rng.integers(1, 70)); radius = float(rng.uniform(80, 900)); lam = float(rng.choice([0.0, 1e-6, 1e-5, 5e-5, 1.5e-4]))
    N = int(rng.integers(1, 8)); P_d = float(rng.uniform(0.5, 0.99)); eta2 = float(rng.choice([4.61, 5.99, 9.21])); period = float(rng.choice([1.0, 2.5, 4.0]))
    ns = int(rng.integers(4, 12))
    sc = make_scenario(T=T, radius=radius, lambda_phi=lam, n_scans=ns, P_d=P_d, period=period, seed=seed0 + case)
    desc = 'case %d: T=%d r=%.0f lam=%.1e N=%d Pd=%.2f eta2=%.2f dt=%.1f scans=%d' % (case, T, radius, lam, N, P_d, eta2, period, ns)
    g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, x0=sc["x0"], t0=sc["t0"], accepted=None)
    t0 = time.time()
    try:
        trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"])
        g["accepted"] = acc
        o = make_oracle(g)
        ok = True
        for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
            if time.time() - t0 > 15 or (k > 0 and st["L"] > 2000): print(desc, 'stopped after scan', k, '(the oracle gets slow beyond this size)'); break
            info = o.add_scan(float(t), z)
            trk.addMeasurementList(MeasurementList(float(t), z))
            st = trk.lastScanStats
            if trk.toc['Total'] > 0.2: print(desc, 'SLOW scan', k, '%.2f s' % trk.toc['Total'], 'L', st['L'], 'ilp', st['ilp'], 'branched', st['branched'], 'iters', st['blp_iters_max'], flush=True)
            os_, ts = o.selected(), tracker_selected(trk)
            lb, tb = o.leaf_batch(), trk.leafBatch()
            checks = [(st["L"], st["G"]) == (info["L"], info["G"]), np.array_equal(st["unused"], info["unused"]),
                      [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__],
                      np.array_equal(os_["ID"], ts["ID"]) and np.array_equal(os_["meas"], ts["meas"]),
                      states_close(os_["x"], ts["x"]) and np.allclose(os_["cnllr"], ts["cnllr"], rtol=0, atol=SCORE_ATOL),
                      len(o.clusters) == len(trk.__clusterList__), np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]),
                      o.n_ilp == trk.nOptimSolved]
            if not all(checks):
                print(desc, 'MISMATCH at scan', k, checks); ok = False; bad += 1; break
        trk.close()
        if ok: print(desc, 'ok  L=%d ilp=%d %.1fs' % (st["L"], o.n_ilp, time.time() - t0), flush=True)
    except Exception as e:
        print(desc, 'ERROR', repr(e)[:300], flush=True); bad += 1
print('%d cases, %d bad' % (n_cases, bad))
