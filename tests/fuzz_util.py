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


def run_case_streamed(seed, max_leaves=2000, budget_s=15.0):
    """The same scenarios with a host that streams the scans in and looks at the end: commit and admission of the initiator's births ride
    in the next scan's grow launch (fgrow_adm_kernel), reports are folded two scans late.  The oracle runs first (it decides where the
    scenario stops); per-scan statistics (from the tracker's log) and the final state are compared."""
    from test_tracker_gpu import make_tracker, tracker_selected, states_close, SCORE_ATOL
    from trace_util import make_oracle
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc, N, eta2, desc = scenario_of(seed)
    g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, x0=sc["x0"], t0=sc["t0"], accepted=None)
    t0 = time.time()
    trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"], logScanStats=True)
    try:
        g["accepted"] = acc
        o = make_oracle(g)
        infos = []
        for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
            if time.time() - t0 > budget_s or (k > 0 and infos[-1]["L"] > max_leaves):
                break
            infos.append(dict(o.add_scan(float(t), z), n_ilp=o.n_ilp, n_targets=len(o.targets)))
        for z, t in zip(sc["scans"][:len(infos)], sc["times"][:len(infos)]):
            trk.addMeasurementList(MeasurementList(float(t), z))          # (nothing is looked at in between)
        os_, ts = o.selected(), tracker_selected(trk)
        lb, tb = o.leaf_batch(), trk.leafBatch()
        log = trk.scanStatsLog
        checks = [len(log) == len(infos) and all((s["L"], s["G"], s["ilp"], s["nTargets"]) == (i["L"], i["G"], i["n_ilp"], i["n_targets"]) and np.array_equal(s["unused"], i["unused"])
                                                   for s, i in zip(log, infos)),
                  [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__],
                  np.array_equal(os_["ID"], ts["ID"]) and np.array_equal(os_["meas"], ts["meas"]),
                  states_close(os_["x"], ts["x"]) and np.allclose(os_["cnllr"], ts["cnllr"], rtol=0, atol=SCORE_ATOL),
                  np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]) and states_close(lb["x"], tb["x"], rel=FUZZ_REL)]
        if not all(checks):
            return False, desc, 'MISMATCH (streamed): per-scan statistics %s targets %s selection %s states %s leaves %s' % tuple(checks)
        return True, desc, ' streamed %d scans, %d targets, %.1fs' % (len(infos), len(o.targets), time.time() - t0)
    finally:
        trk.close()


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


# AIS-aided scenarios: the reference carries the covariances of a target that has met an AIS message in float64 (models/ais.py:4), and so
# does the forest (csrc/mht_vtab.h: float64 values; csrc/mht_la64.h: dgemm chains and dgesv in OpenBLAS' order): states and covariances of
# ALL leaves are compared bit for bit, in the reference's dtypes; only the cumulative scores have a tolerance (the NLLR constant's log).
AIS_SCORE_ATOL = 2e-5


def run_case_ais(seed, max_leaves=2500, budget_s=20.0):
    """AIS-aided variant (Tracker(aisAided=True), addMeasurementList(scan, aisList, aisInitialization=False); tracker.py:417-552): random
    scenario with N <= 7 and a finite radar range, random AIS traffic (share of equipped targets, report probability), similar-state
    pruning on some scans.  Decisions exact; states and covariances of the selected nodes and of all leaves bit for bit (np.array_equal,
    float64 where the reference carries float64); cumulative scores to AIS_SCORE_ATOL."""
    from trace_util import make_oracle_ais, oracle_rows
    from util import live_numpy_f64_is_pinned
    exact = live_numpy_f64_is_pinned()      # (float64 LAPACK order of THIS host's numpy = the one csrc/mht_la64.h restates)
    same = np.array_equal if exact else (lambda a, b: np.shape(a) == np.shape(b) and np.allclose(a, b, rtol=1e-12, atol=1e-12))
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.ais import AisMessage, AisMessageList
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_ais
    import mht_oracle as orc
    sc, N, eta2, desc = scenario_of(seed)
    N = min(N, 7)
    prng = np.random.default_rng(seed + 1234)
    equipped, p_report = float(prng.choice([0.3, 0.6, 1.0])), float(prng.choice([0.4, 0.8]))
    ais = make_ais(sc, seed=seed + 5, equipped=equipped, p_report=p_report)
    rr = 1.5 * sc["radius"]
    ais_init = bool(prng.uniform() < 0.6)          # the reference's default: messages no track took start tracks
    if ais_init and prng.uniform() < 0.5:
        sc["x0"] = sc["x0"][::2].copy()            # (half of the ships have no track at the start)
    desc += ' AIS equipped=%.1f p=%.1f N=%d init=%d' % (equipped, p_report, N, ais_init)
    g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, eta2_ais=9.45, x0=sc["x0"], t0=sc["t0"],
             radar_range=rr, position=np.asarray(sc["centre"], dtype=np.float64), with_initiator=True, accepted=None)
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=eta2, radarRange=rr, position=g["position"], aisAided=True,
                  maxTargets=512, maxNodes=1 << 19, maxMeasurements=512)
    t0 = time.time()
    try:
        acc = []
        for x in sc["x0"]:
            n0 = trk.nTargets
            trk.initiateTarget(Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized"))
            acc.append(trk.nTargets > n0)
        g["accepted"] = acc
        o = make_oracle_ais(g)
        st, msg, nf = {"L": 0}, '', 0
        for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
            if time.time() - t0 > budget_s or (k > 0 and st["L"] > max_leaves):
                msg = 'stopped after scan %d' % k
                break
            on = bool(prng.uniform() < 0.3)
            msgs = ais[k] if prng.uniform() < 0.85 else []
            info = o.add_scan(float(t), z, prune_similar=on, ais=[orc.AisMessage(m[0], m[1].copy(), m[2], m[3]) for m in msgs], ais_initialization=ais_init)
            trk.addMeasurementList(MeasurementList(float(t), z), AisMessageList([AisMessage(*m) for m in msgs]), aisInitialization=ais_init, pruneSimilar=on)
            st = trk.lastScanStats
            nodes = list(trk.getTrackNodes())
            tb = trk.leafBatch()
            lb = oracle_rows([l for r in o.targets for l in r.leaves()])
            os_ = oracle_rows(o.track_nodes)
            t_mmsi = np.array([0 if n.mmsi is None else n.mmsi for n in nodes], dtype=np.int64)
            # (None without an identity: a merged new target, m_of_n.py:150 -- 0 in the oracle's table; None with one: no radar measurement)
            t_meas = np.array([(-1 if n.mmsi is not None else 0) if n.measurementNumber is None else n.measurementNumber for n in nodes], dtype=np.int64)
            o_meas = np.where((os_["meas"] == -1) & (os_["mmsi"] == 0), 0, os_["meas"])
            nf += info["n_fused"]
            t_x = np.array([np.asarray(n.x_0, dtype=np.float64) for n in nodes]).reshape(-1, 4)
            checks = [st["L"] == info["L"], np.array_equal(st["unused"], info["unused"]),
                      [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__],
                      np.array_equal(os_["ID"], [n.ID for n in nodes]) and np.array_equal(o_meas, t_meas) and np.array_equal(os_["mmsi"], t_mmsi),
                      same(os_["x"], t_x) and np.allclose(os_["cnllr"], [float(n.cumulativeNLLR) for n in nodes], rtol=0, atol=AIS_SCORE_ATOL),
                      len(o.clusters) == len(trk.__clusterList__) and all(np.array_equal(a, np.asarray(b)) for a, b in zip(o.clusters, trk.__clusterList__)),
                      np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]) and np.array_equal(lb["mmsi"], tb["mmsi"])
                      and same(lb["x"], tb["x"]) and np.array_equal(lb["Pf64"], tb["Pf64"]) and same(lb["P"], tb["P"])
                      and np.allclose(lb["cnllr"], tb["cnllr"], rtol=0, atol=AIS_SCORE_ATOL),
                      o.n_ilp == trk.nOptimSolved]
            if not all(checks):
                return False, desc, 'MISMATCH at scan %d: L %s unused %s targets %s selection %s states %s clusters %s leaves %s ilps %s' % ((k,) + tuple(checks))
        return True, desc, msg + ' L=%d fused=%d ilp=%d %.1fs' % (st["L"], nf, o.n_ilp, time.time() - t0)
    finally:
        trk.close()


def run_case_ct(seed, max_leaves=2500, budget_s=15.0, similar=False):
    """Constant-turn forests (pymht_amd/models/ct.py, six states; MHT_FOREST_CT): every hypothesis its own Phi(T, w), formed on the device
    from its f64 sin / cos.  Random scenario as above, every root with a RANDOM turn rate w (from exactly 0 and |w| below the model's
    straight-line threshold through gentle turns to 0.6 rad/s) and turn-rate derivative a, against the live oracle whose per-leaf arithmetic
    is the reference's kalman.predict_single + kalman.precalc restated (oracle.process_leaves_ct; kalman.py:67-70, :82-101).
    Decisions -- gating counts, unused measurements, target lists, clusters, selections, leaf sets, number of ILPs -- exact; states and
    covariances of all leaves 1e-6.  similar=True: similar-state pruning (tracker.py:230-231, :1233-1239; pyTarget.py:358-412) switched on and off at
    random from scan to scan with a random merge radius -- the merged node's mean covariance lives under the missed-detection child's key."""
    from test_tracker_gpu import tracker_selected, states_close, SCORE_ATOL
    from trace_util import make_oracle
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import ct
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc, N, eta2, desc = scenario_of(seed)
    prng = np.random.default_rng(seed + 4242)
    T = len(sc["x0"])
    kind = prng.integers(0, 5, size=T)
    w = np.where(kind == 0, 0.0, np.where(kind == 1, prng.uniform(-1e-9, 1e-9, size=T), np.where(kind == 2, prng.normal(0.0, 0.02, size=T),
                 np.where(kind == 3, prng.uniform(-0.6, 0.6, size=T), prng.normal(0.0, 0.15, size=T)))))
    a = np.where(prng.uniform(size=T) < 0.5, 0.0, prng.normal(0.0, 1e-3, size=T))
    x0 = np.concatenate([sc["x0"], w[:, None], a[:, None]], axis=1)
    thr = float(prng.choice([4.0, 6.0, 12.0])) if similar else 4
    desc += ' CT |w|max=%.3f' % float(np.abs(w).max()) + (' similar thr=%.0f' % thr if similar else '')
    trk = Tracker(ct, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=eta2, useInitiator=False, maxTargets=256, maxNodes=1 << 18, maxMeasurements=512,
                  pruneThreshold=thr)
    t0 = time.time()
    try:
        acc = []
        for x in x0:
            n0 = trk.nTargets
            trk.initiateTarget(Target(sc["t0"], None, x.copy(), ct.P0, status="preinitialized"))
            acc.append(trk.nTargets > n0)
        g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2, x0=x0, t0=sc["t0"], accepted=acc)
        o = make_oracle(g, with_initiator=False, model=ct)
        st, msg = {"L": 0}, ''
        for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
            if time.time() - t0 > budget_s or (k > 0 and st["L"] > max_leaves):
                msg = 'stopped after scan %d' % k
                break
            on = bool(similar and prng.uniform() < 0.75)
            info = o.add_scan(float(t), z, prune_similar=on, prune_threshold=thr)
            trk.addMeasurementList(MeasurementList(float(t), z), pruneSimilar=on)
            st = trk.lastScanStats
            os_, ts = o.selected(), tracker_selected(trk, 6)
            lb, tb = o.leaf_batch(), trk.leafBatch()
            checks = [(st["L"], st["G"]) == (info["L"], info["G"]), np.array_equal(st["unused"], info["unused"]),
                      [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__],
                      np.array_equal(os_["ID"], ts["ID"]) and np.array_equal(os_["meas"], ts["meas"]),
                      states_close(os_["x"], ts["x"]) and np.allclose(os_["cnllr"], ts["cnllr"], rtol=0, atol=SCORE_ATOL),
                      len(o.clusters) == len(trk.__clusterList__) and all(np.array_equal(a_, np.asarray(b_)) for a_, b_ in zip(o.clusters, trk.__clusterList__)),
                      np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]) and states_close(lb["x"], tb["x"], rel=1e-6)
                      and states_close(lb["P"], np.asarray(tb["P"], dtype=np.float64), rel=1e-6),
                      o.n_ilp == trk.nOptimSolved]
            if not all(checks):
                return False, desc, 'MISMATCH at scan %d: gating %s unused %s targets %s selection %s states %s clusters %s leaves %s ilps %s' % ((k,) + tuple(checks))
        return True, desc, msg + ' L=%d ilp=%d %.1fs' % (st["L"], o.n_ilp, time.time() - t0)
    finally:
        trk.close()
