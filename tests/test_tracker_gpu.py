"""GPU: the whole scan path (`Tracker.addMeasurementList`) on the device forest against the golden scan traces
recorded from the reference, and against the oracle on a fresh seeded scenario."""
import os
import numpy as np
import pytest

import mht_oracle as orc
from trace_util import check_scan_against_fixture, make_oracle

pytestmark = pytest.mark.gpu

# cumulative score slack: each scan adds an NLLR whose per-leaf float32 constant may differ by 1 ulp(f32) (util.NLLR_ATOL)
SCORE_ATOL = 2e-5


def states_close(a, b, rel=1e-6):
    """The north star's state tolerance against a LIVE oracle (another host's BLAS kernels may differ in the last bit, one
    float32 ulp for initiator-born float32 chains): 1e-6 relative to the largest component of each state vector -- a
    velocity near zero carries the rounding of the ~1e2..1e3 m positions it was differenced from."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    if a.size == 0:
        return True
    scale = np.maximum(np.abs(a).max(axis=1, keepdims=True), 1.0)
    return bool(np.all(np.abs(a - b) <= rel * scale))


def make_tracker(period, lambda_phi, lambda_nu, P_d, N, eta2, x0, t0, P0s=None, x0_f32=None, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    trk = Tracker(pv, period, lambda_phi, lambda_nu, P_d=P_d, N=N, eta2=eta2, **kw)
    acc = []
    for i, x in enumerate(x0):
        n0 = len(trk.__targetList__)
        xr = x.astype(np.float32) if (x0_f32 is not None and x0_f32[i]) else x.copy()      # (g16: float32 roots, own covariances)
        trk.initiateTarget(Target(t0, None, xr, pv.P0 if P0s is None else np.array(P0s[i], dtype=np.float32), status="preinitialized"))
        acc.append(len(trk.__targetList__) > n0)
    return trk, acc


def tracker_selected(trk, nx=4):
    nodes = list(trk.getTrackNodes())
    return dict(ID=np.array([n.ID for n in nodes], dtype=np.int64),
                x=np.array([np.asarray(n.x_0, dtype=np.float64) for n in nodes]).reshape(-1, nx),
                cnllr=np.array([float(n.cumulativeNLLR) for n in nodes]),
                meas=np.array([0 if n.measurementNumber is None else n.measurementNumber for n in nodes], dtype=np.int64))      # (None: a merged new target, m_of_n.py:150)


@pytest.mark.parametrize("name", ["g2_trace_cfg1", "g3_trace_dense", "g3b_trace_cfg2", "g6_trace_cfg3", "g6b_trace_cfg3_long",
                                  "g13_trace_similar", "g13b_trace_similar_cfg2", "g13c_trace_similar_cfg3", "g16_fgrow_kat",
                                  # the same trace in a forest made for 8 192 targets: clustering tables in HBM (cluster_big_kernel), the
                                  # device initiator behind the scan instead of next to the clustering
                                  "g3_trace_dense:big", "g13_trace_similar:big"])
def test_tracker_replays_reference_trace(name, gold_dir):
    """Every scan of a trace recorded from the real reference: gating counts, unused measurements, selections, clusters, target lists,
    terminations -- and the states and covariances of ALL leaves bit for bit (np.array_equal, or sha-256 over all leaves for the hashed
    traces), both dtype chains, births of the device initiator included.  g16 is the known-answer trace of fgrow_kernel itself: roots
    with their own covariances and float32 states (oracle/gen_golden.py::gen_g16)."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    big = name.endswith(":big")
    g = np.load(os.path.join(gold_dir, name.split(":")[0] + ".npz"))
    trk, acc = make_tracker(float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), float(g["P_d"]),
                            int(g["N"]), float(g["eta2"]), g["x0"], float(g["t0"]),
                            P0s=g["P0s"] if "P0s" in g.files else None, x0_f32=g["x0_f32"] if "x0_f32" in g.files else None,
                            **(dict(maxTargets=8192, maxNodes=1 << 18) if big else {}))
    assert acc == [bool(a) for a in g["accepted"]]
    for k in range(int(g["n_scans"])):
        p = "s%02d_" % k
        ids_before = [r.ID for r in trk.__targetList__]
        # (g13*: recorded with addMeasurementList(pruneSimilar=True), tracker.py:230-231)
        trk.addMeasurementList(MeasurementList(float(g["times"][k]), g[p + "z"]), pruneSimilar=bool(g["prune_similar"]) if "prune_similar" in g else False)
        st = trk.lastScanStats
        assert [st["L"], st["G"], st["M"]] == g[p + "LGM"].tolist(), "scan %d L/G/M" % k      # gating: exact counts
        assert np.array_equal(st["unused"], g[p + "unused"]), "scan %d unused measurements" % k
        # selection is reported for the targets alive before termination; the fixture holds survivors only
        sel = tracker_selected(trk)
        leaf = trk.leafBatch()
        leaf_cmp = dict(ID=leaf["ID"].astype(np.int64), meas=leaf["meas"].astype(np.int64), x=leaf["x"], cnllr=leaf["cnllr"], P=leaf["P"])
        ids_after = np.array([r.ID for r in trk.__targetList__])
        # clusters refer to the target list before termination
        check_scan_against_fixture(g, k, ids_after, sel, trk.__clusterList__, len(leaf["ID"]), leaf_cmp, score_atol=SCORE_ATOL)
        dead = sorted(i for i in ids_before if i not in ids_after.tolist())
        assert dead == g[p + "dead"].tolist()
        assert st["branched"] == 0 or st["branched"] <= st["ilp"]
    trk.close()


@pytest.mark.parametrize("N,P_d,period,eta2,lam,seed", [
    (4, 0.85, 2.5, 5.99, 3e-5, 99),
    (1, 0.95, 2.5, 5.99, 3e-5, 7),        # window of one scan: the root advances every scan
    (2, 0.60, 1.0, 9.21, 2e-5, 8),        # low detection probability, wide gate, fast radar
    (6, 0.90, 4.0, 4.61, 4e-5, 9),        # slow radar, narrow gate
    (8, 0.80, 2.5, 5.99, 1.5e-5, 10),     # paths of 9 rows: every cluster takes the ILP kernel's HBM policy
])
def test_tracker_vs_oracle_fresh_scenario(N, P_d, period, eta2, lam, seed):
    """Scenarios that are in no fixture (other windows, detection probabilities, radar periods, gate sizes): oracle and device
    forest side by side, scan by scan."""
    from pymht_amd.utils.scenario import make_scenario
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = make_scenario(T=40, radius=500.0, lambda_phi=lam, n_scans=14, P_d=P_d, period=period, seed=seed)
    g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=eta2,
             x0=sc["x0"], t0=sc["t0"], accepted=None)
    trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, eta2, sc["x0"], sc["t0"])
    g["accepted"] = acc
    o = make_oracle(g)
    assert [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__]
    n_ilp = 0
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        info = o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        assert (st["L"], st["G"]) == (info["L"], info["G"]), k
        assert np.array_equal(st["unused"], info["unused"]), k
        assert [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__], k
        os_, ts = o.selected(), tracker_selected(trk)
        assert np.array_equal(os_["ID"], ts["ID"]) and np.array_equal(os_["meas"], ts["meas"]), k
        assert states_close(os_["x"], ts["x"]) and np.allclose(os_["cnllr"], ts["cnllr"], rtol=0, atol=SCORE_ATOL), k
        assert len(o.clusters) == len(trk.__clusterList__)
        lb, tb = o.leaf_batch(), trk.leafBatch()
        assert np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]), k
        assert states_close(lb["x"], tb["x"]), k
        n_ilp += o.n_ilp
        assert o.n_ilp == trk.nOptimSolved
    assert n_ilp > 0 or N == 1
    # the selected leaf's ancestor chain (lazy `parent` through the device ring) matches the oracle's history
    for n_o, n_t in zip(o.track_nodes, trk.getTrackNodes()):
        h = n_o.history_meas()
        chain = [m.measurementNumber for m in n_t.backtrackNodes()]
        assert h[-len(chain):] == [0 if c is None else int(c) for c in chain][-len(h):] or h[-3:] == chain[-3:]
    trk.close()


@pytest.mark.parametrize("N,P_d,thr,lam,seed,T,radius", [
    (5, 0.9, 4, 3e-5, 21, 40, 500.0),
    (3, 0.8, 7.5, 6e-5, 22, 40, 400.0),       # wider merge radius, more clutter: several hits fused into one node
    (2, 0.95, 15.0, 1e-3, 23, 25, 250.0),     # dense clutter and a radius as wide as the gate: means over up to 6 hits (new covariance values)
])
def test_tracker_vs_oracle_similar_state_pruning(N, P_d, thr, lam, seed, T, radius):
    """Similar-state pruning (tracker.py:230-231, pyTarget.py:358-412) switched on and off from scan to scan, oracle and device
    forest side by side: leaves (fused ones gone, merged ones in the missed-detection slot), selections, clusters, births."""
    from pymht_amd.utils.scenario import make_scenario
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = make_scenario(T=T, radius=radius, lambda_phi=lam, n_scans=16, P_d=P_d, seed=seed)
    g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=N, eta2=5.99,
             x0=sc["x0"], t0=sc["t0"], accepted=None)
    trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], N, 5.99, sc["x0"], sc["t0"], pruneThreshold=thr)
    g["accepted"] = acc
    o = make_oracle(g)
    n_merged = n_multi = 0
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        on = (k % 5) != 3
        before = sum(len(r.leaves()) for r in o.targets)
        info = o.add_scan(float(t), z, prune_similar=on, prune_threshold=thr)
        trk.addMeasurementList(MeasurementList(float(t), z), pruneSimilar=on)
        st = trk.lastScanStats
        assert (st["L"], st["G"]) == (info["L"], info["G"]), k
        assert before == info["L"]
        assert np.array_equal(st["unused"], info["unused"]), k
        assert [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__], k
        os_, ts = o.selected(), tracker_selected(trk)
        assert np.array_equal(os_["ID"], ts["ID"]) and np.array_equal(os_["meas"], ts["meas"]), k
        assert states_close(os_["x"], ts["x"]) and np.allclose(os_["cnllr"], ts["cnllr"], rtol=0, atol=SCORE_ATOL), k
        assert len(o.clusters) == len(trk.__clusterList__)
        lb, tb = o.leaf_batch(), trk.leafBatch()
        assert np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]), k
        assert states_close(lb["x"], tb["x"]), k
        assert np.allclose(lb["cnllr"], tb["cnllr"], rtol=0, atol=SCORE_ATOL), k
        assert np.allclose(lb["P"], tb["P"], rtol=2e-6, atol=1e-6), k
        assert not np.any(tb["flags"] & 8)
        # merged nodes: measurement-less leaves that do not sit at their parent's prediction
        n_merged += sum(1 for r in o.targets for l in r.leaves()
                        if l.meas == 0 and l.parent is not None and l.parent.kids[0] is l
                        and not np.allclose(np.asarray(l.x, dtype=np.float64), o.A @ np.asarray(l.parent.x, dtype=np.float64), rtol=0, atol=1e-4))
        n_multi += sum(1 for c in o.clusters if len(c) > 1)
    assert n_merged > 0 and n_multi > 0
    trk.close()


def test_terminated_tracks_keep_their_history():
    """A terminated track's view walks back through its whole life (the reference keeps it via _pruneEverythingExceptHistory,
    tracker.py:353-381): window nodes captured at termination + the committed roots -- compared with the oracle's terminated nodes,
    also when it is looked at long after the device ring has moved on."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_scenario
    sc = make_scenario(T=30, radius=400.0, lambda_phi=3e-5, n_scans=16, P_d=0.55, seed=3)      # low P_d: tracks die
    trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], 3, 5.99, sc["x0"], sc["t0"], useInitiator=False)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99)
    for x, a in zip(sc["x0"], acc):
        assert o.initiate_target(sc["t0"], x.copy(), orc.model_P0()) == a
    for z, t in zip(sc["scans"], sc["times"]):
        o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
    dead = trk.__terminatedTargets__
    assert len(dead) == len(o.terminated) >= 3
    want = {n.ID: n.history_meas() for n in o.terminated}
    for v in dead:
        chain = [0 if m.measurementNumber is None else int(m.measurementNumber) for m in v.backtrackNodes()]
        assert chain == want[v.ID], v.ID
    trk.close()


def test_terminated_tracks_keep_their_history_streamed_with_the_device_initiator():
    """The same with the default drop-in configuration -- device initiator, reports folded TWO scans late (tracker.py here:
    _queue_report) -- and scans streamed in without a look in between: the window chain of a track that died in scan k is fetched when
    scan k + 2 has been queued, so the ring must still hold the layer of its root (N + 4 layers; with N + 3 the root and the committed
    history of tracks that die mid-stream were lost).  At least four more scans follow the deaths."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_scenario
    from trace_util import make_oracle
    sc = make_scenario(T=30, radius=400.0, lambda_phi=3e-5, n_scans=20, P_d=0.55, seed=3)      # low P_d: tracks die
    trk, acc = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], 3, 5.99, sc["x0"], sc["t0"])      # (useInitiator defaults to True)
    o = make_oracle(dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=3, eta2=5.99, x0=sc["x0"], t0=sc["t0"], accepted=acc))
    died_at = {}
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        o.add_scan(float(t), z)
        for n in o.terminated:
            died_at.setdefault(n.ID, k)
        trk.addMeasurementList(MeasurementList(float(t), z))          # (streamed: nothing is looked at in between)
    dead = trk.__terminatedTargets__
    assert len(dead) == len(o.terminated) >= 3
    assert sum(1 for k in died_at.values() if k <= len(sc["scans"]) - 5) >= 2, "the scenario must have deaths with >= 4 scans behind them"
    want = {n.ID: n.history_meas() for n in o.terminated}
    for v in dead:
        nodes = v.backtrackNodes()
        chain = [0 if m.measurementNumber is None else int(m.measurementNumber) for m in nodes]
        assert chain == want[v.ID], v.ID
    trk.close()


def test_toc_is_the_scans_own_cost_not_the_hosts_idle_time(caplog):
    """addMeasurementList is pipelined (the report of scan k is folded by the call for scan k+1), so the wall time between a call and
    its fold is the host's idle time.  toc['Total'] must be what the scan cost -- host call + device stages + fold -- and the per-stage
    keys the reference always logs (tracker.py:87-98, :291-294) must be there on every scan without deviceTiming: they come from
    wall-clock stamps the kernels take themselves."""
    import logging
    import time as _time
    from pymht_amd.utils.scenario import make_scenario
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = make_scenario(T=60, radius=600.0, lambda_phi=3e-5, n_scans=6, P_d=0.9, period=0.05, seed=5)
    trk, _ = make_tracker(sc["period"], sc["lambda_phi"], 1e-4, sc["P_d"], 4, 5.99, sc["x0"], sc["t0"])
    with caplog.at_level(logging.WARNING, logger="pymht_amd.tracker"):
        for z, t in zip(sc["scans"], sc["times"]):
            trk.addMeasurementList(MeasurementList(float(t), z))
            _time.sleep(0.08)                      # a host that feeds one scan per (longer than the) radar period and reads nothing in between
        log = trk.runtimeLog
    assert not [r for r in caplog.records if "real time demand" in r.getMessage()], "idle time between scans was billed to the scans"
    n = len(sc["scans"])
    for key in ("Total", "Process", "Cluster", "Optim", "ILP-Prune", "DynN", "N-Prune", "Terminate", "Init"):
        assert len(log[key]) == n, key
    tot, proc, clu, opt = (np.array(log[k]) for k in ("Total", "Process", "Cluster", "Optim"))
    assert np.all(tot < 0.02) and np.all(tot > 0)
    assert np.all(proc > 1e-6) and np.all(proc < 1e-3) and np.all(clu > 1e-6) and np.all(clu < 1e-3) and np.all(opt > 1e-6) and np.all(opt < 5e-3)
    assert np.all(tot >= proc + clu + opt - 1e-9)
    trk.close()


def test_batch_admission_on_a_non_blocking_stream():
    """A tracker made while a non-blocking stream is current (torch's side streams are) admits the same batch of initial targets every
    time: nothing the library does on the way (growing a staging buffer, clearing it) may go out on the NULL stream, which such a stream
    does not wait for -- zeros landing behind the staged candidates put them all at the origin, where all but one are refused as
    neighbours (Tracker.initiateTarget, tracker.py:147-160)."""
    import torch
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.scenario import make_scenario
    sc = make_scenario(T=500, radius=9000.0, lambda_phi=3e-6, n_scans=1, P_d=0.9, seed=11)
    want = None
    for rep in range(200):
        side = torch.cuda.Stream(device=0, priority=-1 if rep % 2 else 0)
        with torch.cuda.stream(side):
            trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99, useInitiator=False,
                          maxTargets=1024, maxNodes=1 << 16, maxMeasurements=1024)
            out = trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
            ids = [t.ID for t in out]
            trk.close()
        if want is None:
            want = ids
            assert len(want) > 400
        assert ids == want, (rep, len(ids), len(want))


@pytest.mark.parametrize("T,seed", [(60, 5), (24, 6)])
def test_streamed_scans_equal_scans_looked_at_one_by_one(T, seed):
    """The drop-in API with the device initiator, two ways: a host that looks at the tracker after every scan (the scan's commit and the
    admission of what its initiator gave birth to then run as a launch of their own, post_scan_kernel, and every report is folded at
    once) and a host that streams the scans in and looks at the end (commit and admission ride in workgroup 0 of the NEXT scan's grow
    launch, fgrow_adm_kernel; newborn targets are grown by that launch's extra workgroups; reports are folded two scans late).  No
    initial targets: every track is started by the initiator, most of them in the same scan (more than the 16 workgroups that grow
    newborn targets: they loop), later ones one or two at a time, some tracks die.  Scan statistics, target lists, selected
    hypotheses and track histories must agree exactly (Tracker.initiateTarget / addMeasurementList, tracker.py:147-160, :264-278)."""
    from pymht_amd.tracker import Tracker
    from pymht_amd.models import pv
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_scenario
    sc = make_scenario(T=T, radius=2500.0, lambda_phi=4e-6, n_scans=24, P_d=0.8, seed=seed)
    runs = []
    for look in (True, False):
        trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=4, eta2=5.99, logScanStats=True,
                      maxTargets=512, maxNodes=1 << 17, maxMeasurements=512)
        for z, t in zip(sc["scans"], sc["times"]):
            trk.addMeasurementList(MeasurementList(float(t), z))
            if look:
                _ = trk.lastScanStats
        sel = tracker_selected(trk)
        log = [{k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in s.items()} for s in trk.scanStatsLog]
        chains = [[m.measurementNumber for m in n.backtrackNodes()] for n in trk.getTrackNodes()]
        dead = sorted(v.ID for v in trk.__terminatedTargets__)
        runs.append((log, sel, chains, dead, [r.ID for r in trk.__targetList__]))
        trk.close()
    (la, sa, ca, da, ia), (lb, sb, cb, db, ib) = runs
    assert ia == ib and da == db and len(ia) >= T // 2
    births = [b["nTargets"] - a["nTargets"] for a, b in zip(la[:-1], la[1:])]
    assert max(births) > 16 or T < 32, births              # (one scan gives birth to more targets than there are workgroups for them)
    assert la == lb
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert ca == cb


def test_streamed_headline_scans_with_many_streams_alive():
    """A process that has made more streams than the device has hardware queues (PyTorch keeps a pool of 32 per priority once one is
    asked for): the forest's side stream then SHARES a hardware queue with other streams.  The streamed path must not depend on which
    -- with the initiator's launch queued in front of the next scan's staging kernel it deadlocked until the device-side spin timeouts
    (2 s per scan) once every CU was held by target workgroups waiting for the staged scan (the headline size fills the machine).
    Several trackers one after the other, each with streams of its own: every scan well inside a millisecond, same results each time."""
    import time
    import torch
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    keep = [torch.cuda.Stream(device=0, priority=p) for p in (0, -1) for _ in range(4)]      # (materialises both pools)
    sc = make_config("cfg3", seed=5446, n_scans=48, confine=True)
    lists = [MeasurementList(float(t), z) for z, t in zip(sc["scans"], sc["times"])]
    finals = []
    for rep in range(4):
        trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, maxTargets=2048, maxNodes=1 << 19,
                      maxMeasurements=1024)
        trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
        trk.addMeasurementList(lists[0])
        trk.synchronize()
        t0 = time.perf_counter()
        for sl in lists[1:]:
            trk.addMeasurementList(sl)
        trk.synchronize()
        dt = time.perf_counter() - t0
        assert dt < 0.5, "47 streamed scans took %.3f s: a device-side wait timed out" % dt
        finals.append(tracker_selected(trk))
        trk.close()
    for f in finals[1:]:
        for k in finals[0]:
            assert np.array_equal(f[k], finals[0][k]), k
    del keep
