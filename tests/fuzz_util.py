"""Randomised parity scenarios (shared by tests/test_fuzz_gpu.py and tools/fuzz_parity.py): the drop-in Tracker (device forest, device
initiator) against the oracle (NumPy restatement of the reference, its own initiator restatement) on a random small scenario --
targets, area, clutter, N-scan window, detection probability, gate, radar period -- scan by scan."""
import os
import time

import numpy as np

# all-leaves state tolerance against the LIVE oracle (this host's BLAS): the north star's 1e-6; MHT_FUZZ_REL=0 demands bit equality
# (holds where the host's OpenBLAS picks the kernels the device arithmetic follows: Haswell / SkylakeX / Zen)
FUZZ_REL = float(os.environ.get("MHT_FUZZ_REL", "1e-6"))


def scenario_of(seed):
    from pymht_amd.utils.scenario import make_scenario
    rng = np.random.default_rng(seed)
    T = int(rng.integers(1, 70)); radius = float(rng.uniform(80, 900)); lam = float(rng.choice([0.0, 1e-6, 1e-5, 5e-5, 1.5e-4]))
    N = int(rng.integers(1, 8)); P_d = float(rng.uniform(0.5, 0.99)); eta2 = float(rng.choice([4.61, 5.99, 9.21])); period = float(rng.choice([1.0, 2.5, 4.0]))
    ns = int(rng.integers(4, 12))
    sc = make_scenario(T=T, radius=radius, lambda_phi=lam, n_scans=ns, P_d=P_d, period=period, seed=seed)
    desc = 'seed %d: T=%d r=%.0f lam=%.1e N=%d Pd=%.2f eta2=%.2f dt=%.1f scans=%d' % (seed, T, radius, lam, N, P_d, eta2, period, ns)
    return sc, N, eta2, desc


def run_case(seed, max_leaves=2000, budget_s=15.0, similar=False):
    """Returns (ok, description, message).  similar=True: similar-state pruning (addMeasurementList(pruneSimilar=True)) switched on
    and off at random from scan to scan, with a random pruneThreshold."""
    from test_tracker_gpu import make_tracker, tracker_selected, states_close, SCORE_ATOL
    from trace_util import make_oracle
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc, N, eta2, desc = scenario_of(seed)
    g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, x0=sc["x0"], t0=sc["t0"], accepted=None)
    t0 = time.time()
    prng = np.random.default_rng(seed + 77)
    thr = float(prng.choice([4.0, 6.0, 12.0])) if similar else 4
    if similar:
        desc += ' similar thr=%.0f' % thr
    trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"], pruneThreshold=thr)
    try:
        g["accepted"] = acc
        o = make_oracle(g)
        st, msg = {"L": 0}, ''
        for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
            if time.time() - t0 > budget_s or (k > 0 and st["L"] > max_leaves):
                msg = 'stopped after scan %d (the oracle gets slow beyond this size)' % k
                break
            on = bool(similar and prng.uniform() < 0.75)
            info = o.add_scan(float(t), z, prune_similar=on, prune_threshold=thr)
            trk.addMeasurementList(MeasurementList(float(t), z), pruneSimilar=on)
            st = trk.lastScanStats
            os_, ts = o.selected(), tracker_selected(trk)
            lb, tb = o.leaf_batch(), trk.leafBatch()
            checks = [(st["L"], st["G"]) == (info["L"], info["G"]), np.array_equal(st["unused"], info["unused"]),
                      [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__],
                      np.array_equal(os_["ID"], ts["ID"]) and np.array_equal(os_["meas"], ts["meas"]),
                      states_close(os_["x"], ts["x"]) and np.allclose(os_["cnllr"], ts["cnllr"], rtol=0, atol=SCORE_ATOL),
                      len(o.clusters) == len(trk.__clusterList__),
                      # (ALL leaves, not only the selected ones, at the north star's 1e-6 -- against a live oracle on this host's BLAS)
                      np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]) and states_close(lb["x"], tb["x"], rel=FUZZ_REL),
                      o.n_ilp == trk.nOptimSolved]
            if not all(checks):
                return False, desc, 'MISMATCH at scan %d: gating %s unused %s targets %s selection %s states %s clusters %s leaves %s ilps %s' % ((k,) + tuple(checks))
        return True, desc, msg + ' L=%d ilp=%d %.1fs' % (st["L"], o.n_ilp, time.time() - t0)
    finally:
        trk.close()
