"""GPU: edge cases of the device forest through the Tracker API (empty and ragged inputs,
terminations, capacity errors) and the exact branch-and-bound path inside the pipeline."""
import numpy as np
import pytest

import mht_oracle as orc
from trace_util import make_oracle

pytestmark = pytest.mark.gpu


def _mk(sc, N=3, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=5.99, **kw)
    for x in sc["x0"]:
        trk.initiateTarget(Target(sc["t0"], None, x.copy(), pv.P0))
    return trk


def _scenario(**kw):
    from pymht_amd.utils.scenario import make_scenario
    args = dict(T=30, radius=400.0, lambda_phi=3e-5, n_scans=8, P_d=0.85, seed=7)
    args.update(kw)
    return make_scenario(**args)


def test_empty_scans_and_no_targets():
    from pymht_amd.tracker import Tracker
    from pymht_amd.models import pv
    from pymht_amd.utils.classDefinitions import MeasurementList
    trk = Tracker(pv, 2.5, 1e-5, 1e-4, P_d=0.9, N=3, useInitiator=False)
    # no targets, no measurements
    trk.addMeasurementList(MeasurementList(1002.5, np.zeros((0, 2), dtype=np.float32)))
    assert trk.lastScanStats["L"] == 0 and len(trk.getTrackNodes()) == 0
    # no targets, some measurements: nothing is gated, everything is unused
    trk.addMeasurementList(MeasurementList(1005.0, np.random.default_rng(0).uniform(-50, 50, (7, 2)).astype(np.float32)))
    assert trk.lastScanStats["unused"].all() and trk.lastScanStats["L"] == 0
    trk.close()
    # targets, but scans without measurements: every hypothesis gets exactly its missed-detection child
    sc = _scenario(n_scans=3)
    trk = _mk(sc, useInitiator=False)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99)
    for x in sc["x0"]:
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0())
    assert [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__]
    for k in range(3):
        z = np.zeros((0, 2), dtype=np.float32)
        n_before = len(trk.__targetList__)
        trk.addMeasurementList(MeasurementList(float(sc["times"][k]), z))
        o.add_scan(float(sc["times"][k]), z)
        st = trk.lastScanStats
        assert st["G"] == 0 and st["L"] == n_before          # all-miss trees: one leaf per target alive before the scan
        want = o.selected()
        got = trk.getTrackNodes()
        assert [int(g.ID) for g in got] == want["ID"].tolist()
        # live oracle on this host's BLAS: last-bit differences are possible (the golden fixtures pin exact values)
        assert np.allclose(np.array([g.x_0 for g in got]).reshape(-1, 4), want["x"], rtol=1e-6, atol=1e-9)
        assert np.allclose(np.array([g.cumulativeNLLR for g in got]), want["cnllr"], rtol=0, atol=2e-5)
    trk.close()


def test_range_termination_matches_oracle():
    """Tracks leaving radarRange are terminated (tracker.py:895) -- the target list shrinks exactly like the oracle's."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(T=25, radius=300.0, n_scans=8, seed=21, sigma_v=20.0)
    rng_ = 260.0
    trk = _mk(sc, useInitiator=False, radarRange=rng_)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99, radarRange=rng_)
    for x in sc["x0"]:
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0())
    killed = 0
    for z, t in zip(sc["scans"], sc["times"]):
        info = o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
        assert [r.ID for r in o.targets] == [r.ID for r in trk.__targetList__]
        killed += len(info["dead"])
    assert killed > 0 and any(n.status == "OutOfRange" for n in trk.__terminatedTargets__)
    trk.close()


def test_forced_branch_and_bound_inside_the_forest():
    """blpMaxIter=0 switches the dual ascent off: every ILP with a conflict among the per-target minimisers has to be
    proven optimal by the GPU branch and bound.  Selections must still equal the oracle's exact optimum."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    sc = make_config("cfg2", seed=11, n_scans=10)          # 50 targets in 700 m, ~200 measurements/scan: many contested ILPs
    trk = _mk(sc, N=4, useInitiator=False, blpMaxIter=0)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=4, eta2=5.99)
    for x in sc["x0"]:
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0())
    branched = 0
    for z, t in zip(sc["scans"], sc["times"]):
        o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
        want, got = o.selected(), trk.getTrackNodes()
        assert [int(g.ID) for g in got] == want["ID"].tolist()
        assert [int(g.measurementNumber) for g in got] == want["meas"].tolist()
        branched += trk.lastScanStats["branched"]
    assert branched > 0
    trk.close()


def test_capacity_overflow_is_reported_not_fatal():
    from pymht_amd import _lib
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(T=40, radius=300.0, lambda_phi=8e-5, n_scans=8, seed=5)
    trk = _mk(sc, N=5, useInitiator=False, maxTargets=64, maxNodes=1024)
    with pytest.raises(_lib.MhtError) as ei:
        for z, t in zip(sc["scans"], sc["times"]):
            trk.addMeasurementList(MeasurementList(float(t), z))
    assert ei.value.code in (_lib.MHT_E_CAPACITY, _lib.MHT_E_STATE)
    with pytest.raises(RuntimeError):       # the tracker refuses further scans (its device forest is dead)
        trk.addMeasurementList(MeasurementList(float(sc["times"][-1]) + 2.5, sc["scans"][-1]))
    trk.close()


def test_api_rejections_and_views():
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(n_scans=6)
    trk = _mk(sc, useInitiator=False, deviceTiming=True)      # (per-stage device times in runtimeLog)
    with pytest.raises(NotImplementedError):
        trk.addMeasurementList(MeasurementList(float(sc["times"][0]), sc["scans"][0]), aisList=[object()])
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99)
    for x in sc["x0"]:
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0())
    for z, t in zip(sc["scans"], sc["times"]):
        o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
    assert set(trk.getRuntimeAverage()) >= {"Total", "Process", "Cluster", "Optim", "N-Prune"}
    # lazy parent chains (device ring + committed root history) reproduce the oracle's measurement history
    for n_o, n_t in zip(o.track_nodes, trk.getTrackNodes()):
        chain = [0 if m.measurementNumber is None else int(m.measurementNumber) for m in n_t.backtrackNodes()]
        assert chain == n_o.history_meas()
        assert abs(float(n_t.getScore()) - float(n_o.score())) < 2e-5
        assert n_t.P_0.shape == (4, 4) and np.allclose(n_t.P_0, np.asarray(n_o.P, dtype=np.float32), rtol=1e-6)
    # leaves of every root, DFS order
    for r_o, r_t in zip(o.targets, trk.__targetList__):
        lo, lt = r_o.leaves(), r_t.getLeafNodes()
        assert [l.meas for l in lo] == [l.measurementNumber for l in lt]
    trk.close()


def test_many_targets_multi_chunk_paths():
    """1500 targets / ~1600 measurements per scan: more targets than one pass of the single-workgroup kernels covers (cluster
    scans 1024 per pass, commit 512), several hundred clusters -- against the oracle, scan by scan."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(T=1500, radius=9000.0, lambda_phi=1e-6, n_scans=4, seed=21)
    trk = _mk(sc, N=3, useInitiator=False, maxTargets=2048, maxMeasurements=2048)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99)
    for x0 in sc["x0"]:
        o.initiate_target(sc["t0"], x0.copy(), orc.model_P0())
    assert trk.nTargets == len(o.targets) > 1400
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        info = o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        assert (st["L"], st["G"]) == (info["L"], info["G"]), k
        assert np.array_equal(st["unused"], info["unused"]), k
        want = o.selected()
        sel = trk._sel[0]
        assert sel["id"].tolist() == want["ID"].tolist(), k
        assert sel["sel_meas"].tolist() == want["meas"].tolist(), k
        assert len(o.clusters) == st["clusters"], k
        assert len(trk.leafBatch()["ID"]) == len(o.leaf_batch()["ID"]), k
    trk.close()


def test_four_thousand_measurements_per_scan():
    """max_meas = 4096 (the bound of the r3 build was 2048): 300 targets in ~3 600 measurements per scan, window 3 -- 64 association
    words per leaf, 28 672 measurement nodes in the clustering graph -- against the oracle, scan by scan."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(T=300, radius=9000.0, lambda_phi=1.4e-5, n_scans=4, seed=33)
    assert 3000 < max(len(z) for z in sc["scans"]) <= 4096
    trk = _mk(sc, N=3, useInitiator=False, maxTargets=512, maxMeasurements=4096)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99)
    for x0 in sc["x0"]:
        o.initiate_target(sc["t0"], x0.copy(), orc.model_P0())
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        info = o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        assert (st["L"], st["G"]) == (info["L"], info["G"]), k
        assert np.array_equal(st["unused"], info["unused"]), k
        want = o.selected()
        sel = trk._sel[0]
        assert sel["id"].tolist() == want["ID"].tolist(), k
        assert sel["sel_meas"].tolist() == want["meas"].tolist(), k
        assert len(o.clusters) == st["clusters"], k
        b, ob = trk.leafBatch(), o.leaf_batch()
        assert np.array_equal(b["ID"], ob["ID"]) and np.array_equal(b["x"], ob["x"]), k
    trk.close()


def test_report_after_births_raw_abi():
    """C-ABI call order step -> add_targets -> report (what a streaming host does): the report of the scan has the rows of the
    targets that took part in it, and the targets added in between show up, with their ids, in the next report."""
    import ctypes as C
    from pymht_amd import _lib
    from pymht_amd.tracker import _REPORT_DTYPE
    sc = _scenario(T=30, n_scans=3, seed=11)
    trk = _mk(sc, N=3, useInitiator=False)
    lib, h = trk._lib, trk._ctx.handle
    n0 = trk.nTargets

    def report():
        rep = _lib.MhtScanReport()
        _lib.check(lib.mht_forest_report(h, C.byref(rep)))
        recs = np.ctypeslib.as_array(C.cast(rep.targets, C.POINTER(C.c_uint8)), shape=(rep.n_targets * _REPORT_DTYPE.itemsize,)) \
            .view(_REPORT_DTYPE).copy()
        return rep, recs

    p = lambda a: a.ctypes.data_as(C.c_void_p)
    z = [np.ascontiguousarray(s_, dtype=np.float32) for s_ in sc["scans"]]
    _lib.check(lib.mht_forest_step_host(h, p(z[0]), len(z[0])))
    x0 = np.array([[5000.0, 5000.0, 1.0, -2.0], [-5000.0, 4000.0, 0.0, 3.0]])          # far from everything: both are accepted
    P0 = np.tile(np.asarray(orc.model_P0(), dtype=np.float32).reshape(1, 16), (2, 1))
    fl, pd, me = np.zeros(2, np.uint8), np.full(2, 0.85), np.zeros(2, np.int32)
    acc, ids = np.zeros(2, np.uint8), np.zeros(2, np.int32)
    _lib.check(lib.mht_forest_add_targets(h, 2, p(x0), p(P0), p(fl), p(pd), p(me), 1, p(acc), p(ids)))
    assert acc.tolist() == [1, 1] and ids.tolist() == [n0, n0 + 1]
    rep, recs = report()
    assert rep.scan == 1 and rep.n_targets == n0 and recs["id"].tolist() == list(range(n0))
    _lib.check(lib.mht_forest_step_host(h, p(z[1]), len(z[1])))
    rep, recs = report()
    assert rep.scan == 2 and rep.n_targets == rep.n_alive + int((recs["status"] != 0).sum())
    assert recs["id"][-2:].tolist() == [n0, n0 + 1]
    assert np.allclose(recs["sel_x"][-2:, 0:2], x0[:, 0:2] + 2.5 * x0[:, 2:4])          # nothing gated out there: the miss hypothesis
    trk.close()


def test_deep_window_ticket_numbered_tiles():
    """200 targets with a 7-scan window: > 20 k leaves = more grow tiles than workgroups are co-resident, so the forest's
    grow kernel numbers its tiles by the ticket counter; 8-row ILP columns.  Against the oracle, scan by scan."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(T=200, radius=3000.0, lambda_phi=1.5e-6, n_scans=10, P_d=0.9, seed=31)
    trk = _mk(sc, N=7, useInitiator=False)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=7, eta2=5.99)
    for x0 in sc["x0"]:
        o.initiate_target(sc["t0"], x0.copy(), orc.model_P0())
    Lmax = 0
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        info = o.add_scan(float(t), z)
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        Lmax = max(Lmax, st["L"])
        assert (st["L"], st["G"]) == (info["L"], info["G"]), k
        assert np.array_equal(st["unused"], info["unused"]), k
        want = o.selected()
        sel = trk._sel[0]
        assert sel["id"].tolist() == want["ID"].tolist() and sel["sel_meas"].tolist() == want["meas"].tolist(), k
    assert Lmax > 512 * 32, Lmax          # the regime this test is about
    trk.close()


def test_two_trackers_interleaved_share_nothing():
    """Two trackers in one process on the same GPU, scans interleaved: each must behave exactly as it does alone (contexts
    share no device state: scratch, counters, LDS attributes, error strings)."""
    from pymht_amd.utils.classDefinitions import MeasurementList

    def digest(trk):
        sel = trk._sel[0]
        st = trk.lastScanStats
        return (st["L"], st["G"], st["ilp"], sel["id"].tolist(), sel["sel_meas"].tolist(), float(sel["sel_cnllr"].sum()))

    scA = _scenario(T=40, radius=500.0, lambda_phi=3e-5, n_scans=10, seed=41)
    scB = _scenario(T=25, radius=300.0, lambda_phi=6e-5, n_scans=10, P_d=0.7, seed=42)
    alone = []
    for sc, N in ((scA, 3), (scB, 5)):
        trk = _mk(sc, N=N)
        out = []
        for z, t in zip(sc["scans"], sc["times"]):
            trk.addMeasurementList(MeasurementList(float(t), z))
            out.append(digest(trk))
        trk.close()
        alone.append(out)
    ta, tb = _mk(scA, N=3), _mk(scB, N=5)
    for k in range(10):
        ta.addMeasurementList(MeasurementList(float(scA["times"][k]), scA["scans"][k]))
        tb.addMeasurementList(MeasurementList(float(scB["times"][k]), scB["scans"][k]))
        assert digest(ta) == alone[0][k], k
        assert digest(tb) == alone[1][k], k
    ta.close()
    tb.close()


def test_preinitialize_batch_equals_one_by_one():
    """Tracker.preInitialize(simList) (tracker.py:139-145) admits the ground-truth objects of simList[0] in one device batch;
    the result must equal initiateTarget called one by one, including the rejection of objects closer than the merge threshold to
    an earlier one."""
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv

    class SimTarget:                                  # what the reference's simulator hands over: .time, .cartesianState()
        def __init__(self, t, x):
            self.time, self._x = t, x

        def cartesianState(self):
            return self._x

    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-300, 300, size=(60, 2)), rng.normal(0, 5, size=(60, 2))], axis=1)
    xs[7, 0:2] = xs[3, 0:2] + [1.0, -2.0]              # too close to an earlier one: rejected
    xs[41, 0:2] = xs[40, 0:2] + [0.5, 0.5]
    a = Tracker(pv, 2.5, 1e-5, 1e-4, P_d=0.9, N=3)
    a.preInitialize([[SimTarget(1000.0, x.copy()) for x in xs]])
    b = Tracker(pv, 2.5, 1e-5, 1e-4, P_d=0.9, N=3)
    for x in xs:
        b.initiateTarget(Target(1000.0, None, x.copy(), pv.P0, status="preinitialized"))
    assert a.nTargets == b.nTargets and 40 < a.nTargets <= 58             # (a few random placements are too close as well)
    la, lb = a.leafBatch(), b.leafBatch()
    assert np.array_equal(la["ID"], lb["ID"]) and np.array_equal(la["x"], lb["x"]) and np.array_equal(la["P"], lb["P"])
    assert [r.ID for r in a.__targetList__] == list(range(a.nTargets))
    a.close()
    b.close()


@pytest.mark.parametrize("T,N,radius,lam,n_scans,kw", [
    (60, 3, 500.0, 3e-5, 12, {}),                                                          # dense clusters
    (200, 7, 3000.0, 1.5e-6, 10, {}),                                                      # N = 7: targets with > 64 leaves (chunked, two passes)
    (700, 3, 6000.0, 1e-6, 6, dict(maxTargets=1024, maxMeasurements=1024)),                # > 512 targets: chunked compaction
])
def test_deferred_commit_equals_immediate_commit(T, N, radius, lam, n_scans, kw):
    """A scan's target-side commit runs inside the NEXT scan's grow_kernel unless the host asks for the committed state first.
    Both orders must leave the same forest: tracker A reports after every scan (commit_kernel on its own every time), tracker B
    steps through the C ABI without looking (deferred commits), with births in the middle (they force one commit), and is
    compared at the end: report rows, used-measurement mask and the complete leaf list."""
    import ctypes as C
    from pymht_amd import _lib
    from pymht_amd.tracker import _REPORT_DTYPE
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(T=T, radius=radius, lambda_phi=lam, n_scans=n_scans, seed=5)
    A = _mk(sc, N=N, useInitiator=False, **kw)
    B = _mk(sc, N=N, useInitiator=False, **kw)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    x0 = np.array([[radius * 3, radius * 3, 1.0, -2.0], [-radius * 3, radius * 2, 0.0, 3.0]])          # far from everything
    P0 = np.tile(np.asarray(orc.model_P0(), dtype=np.float32).reshape(1, 16), (2, 1))
    fl, pd, me = np.zeros(2, np.uint8), np.full(2, 0.85), np.zeros(2, np.int32)

    def births(trk):
        acc, ids = np.zeros(2, np.uint8), np.zeros(2, np.int32)
        _lib.check(trk._lib.mht_forest_add_targets(trk._ctx.handle, 2, p(x0), p(P0), p(fl), p(pd), p(me), 1, p(acc), p(ids)))
        assert acc.tolist() == [1, 1]
        return ids.tolist()

    def report(trk):
        rep = _lib.MhtScanReport()
        _lib.check(trk._lib.mht_forest_report(trk._ctx.handle, C.byref(rep)))
        recs = np.ctypeslib.as_array(C.cast(rep.targets, C.POINTER(C.c_uint8)), shape=(rep.n_targets * _REPORT_DTYPE.itemsize,)) \
            .view(_REPORT_DTYPE).copy()
        used = np.ctypeslib.as_array(C.cast(rep.used, C.POINTER(C.c_uint64)), shape=(rep.used_words,)).copy()
        hdr = (rep.scan, rep.n_targets, rep.n_alive, rep.n_leaves_in, rep.n_children, rep.n_leaves_out, rep.n_clusters, rep.n_ilp)
        return hdr, recs, used

    z = [np.ascontiguousarray(s_, dtype=np.float32) for s_ in sc["scans"]]
    mid = n_scans // 2
    for k in range(n_scans):
        _lib.check(A._lib.mht_forest_step_host(A._ctx.handle, p(z[k]), len(z[k])))
        ra = report(A)
        _lib.check(B._lib.mht_forest_step_host(B._ctx.handle, p(z[k]), len(z[k])))
        if k == mid:
            assert births(A) == births(B)
    hb, rb, ub = report(B)
    assert ra[0] == hb, (ra[0], hb)
    # node indices (sel_node, root_node) are handles: a target's children take their block of the node index space with an
    # atomic, so the numbering differs from run to run; everything else must be identical
    for name in _REPORT_DTYPE.names:
        if name not in ("sel_node", "root_node"):
            assert np.array_equal(ra[1][name], rb[name]), name
    assert np.array_equal(ra[2], ub)
    la, lb = A.leafBatch(), B.leafBatch()
    assert len(la["ID"]) == hb[5] > 0
    for key in la:
        if key != "node":
            assert np.array_equal(la[key], lb[key]), key
    A.close()
    B.close()


def test_deferred_commit_edge_cases_raw_abi():
    """Deferred commits (no report between scans) with (a) no targets at all, (b) targets that all die while nobody looks,
    (c) a pool overflow in the middle: the overflow voids every later scan and surfaces in the first report asked for."""
    import ctypes as C
    from pymht_amd import _lib
    from pymht_amd.tracker import Tracker
    from pymht_amd.models import pv
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def step(trk, z):
        z = np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 2)
        _lib.check(trk._lib.mht_forest_step_host(trk._ctx.handle, p(z), len(z)))

    def report(trk):
        rep = _lib.MhtScanReport()
        rc = trk._lib.mht_forest_report(trk._ctx.handle, C.byref(rep))
        return rc, rep

    # (a) an empty forest stepped three times
    trk = Tracker(pv, 2.5, 1e-5, 1e-4, P_d=0.9, N=3, useInitiator=False)
    rng = np.random.default_rng(1)
    for k in range(3):
        step(trk, rng.uniform(-50, 50, (5, 2)))
    rc, rep = report(trk)
    assert rc == 0 and (rep.scan, rep.n_targets, rep.n_alive, rep.n_leaves_in, rep.n_children) == (3, 0, 0, 0, 0)
    trk.close()
    # (b) two targets whose score limit kills them after a few empty scans, nobody looking in between
    sc = _scenario(T=2, n_scans=12, seed=3)
    ref = _mk(sc, N=3, useInitiator=False)
    trk = _mk(sc, N=3, useInitiator=False)
    empty = np.zeros((0, 2), dtype=np.float32)
    alive = []
    for k in range(12):
        step(ref, empty)
        rc, rep = report(ref)
        assert rc == 0
        alive.append(rep.n_alive)
        step(trk, empty)
    assert alive[0] == 2 and alive[-1] == 0, alive          # they do die on the way
    rc, rep = report(trk)
    assert rc == 0 and (rep.scan, rep.n_targets, rep.n_alive, rep.n_leaves_out) == (12, 0, 0, 0)
    ref.close()
    trk.close()
    # (c) overflow while nobody looks
    sc = _scenario(T=40, radius=300.0, lambda_phi=8e-5, n_scans=8, seed=5)
    trk = _mk(sc, N=5, useInitiator=False, maxTargets=64, maxNodes=1024)
    for z in sc["scans"]:
        step(trk, z)
    rc, rep = report(trk)
    assert rc in (_lib.MHT_E_CAPACITY, _lib.MHT_E_STATE), rc
    with pytest.raises(_lib.MhtError):       # the forest refuses further scans
        step(trk, sc["scans"][-1])
    trk.close()


def test_streaming_equals_reading_after_every_scan():
    """The drop-in API is pipelined: `addMeasurementList` only queues the scan, its report rides to the host with the NEXT scan's grow
    launch (or in a launch of its own as soon as somebody looks), the initiator runs next to the scan's clustering.  Three hosts on
    the same stream with births and terminations (config 2): A looks at the results after every scan, B streams everything and looks
    once at the end, C mixes (leaf snapshots, reports, injected targets in between).  Same forest, same births, same reports."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    sc = make_config("cfg2", seed=5446, n_scans=24)
    A, B, Cc = _mk(sc, N=sc["N"], logScanStats=True), _mk(sc, N=sc["N"], logScanStats=True), _mk(sc, N=sc["N"], logScanStats=True)
    extra = Target(sc["t0"], None, np.array([5000.0, 5000.0, 1.0, 1.0]), pv.P0)
    ids_a = []
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        for trk in (A, B, Cc):
            trk.addMeasurementList(MeasurementList(float(t), z))
        ids_a.append([r.ID for r in A.__targetList__])                  # (waits for scan k's report: a launch of its own)
        if k % 3 == 1:
            assert Cc.lastScanStats["M"] == len(z)
        if k % 5 == 2:
            Cc.leafBatch()
        if k == 9:
            for trk in (A, B, Cc):
                trk.initiateTarget(Target(sc["t0"], None, extra.x_0.copy(), pv.P0))      # (host-side injection between two scans)
    assert [r.ID for r in B.__targetList__] == ids_a[-1] == [r.ID for r in Cc.__targetList__]
    assert len(set(map(tuple, ids_a))) > 3          # births / terminations happened
    for name in ("L", "G", "M", "ilp", "clusters", "leaves_out"):
        a, b, c = ([s[name] for s in t.scanStatsLog] for t in (A, B, Cc))
        assert a == b == c, name
    la, lb, lc = A.leafBatch(), B.leafBatch(), Cc.leafBatch()
    for key in la:
        if key != "node":
            assert np.array_equal(la[key], lb[key]) and np.array_equal(la[key], lc[key]), key
    sa, sb = A._sel[0], B._sel[0]
    for name in ("id", "status", "sel_meas", "sel_x", "sel_cnllr", "score", "root_scan", "root_meas", "root_x"):
        assert np.array_equal(sa[name], sb[name]), name
    for trk in (A, B, Cc):
        trk.close()


def test_console_log_of_the_reference(capsys):
    """printTime / printCluster / printInfo of addMeasurementList, getTimeLogHeader / getTimeLogString (tracker.py:1402-1467): the
    reference's columns, fed from the device's scan report when the scan is folded."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(n_scans=3)
    trk = _mk(sc, N=3, deviceTiming=True)
    for z, t in zip(sc["scans"], sc["times"]):
        trk.addMeasurementList(MeasurementList(float(t), z), printTime=True, printCluster=True, printInfo=True)
    trk.synchronize()
    out = capsys.readouterr().out
    assert out.count("Clusters:") == 3 and out.count("Added scan number:") == 3 and out.count("nTrack") == 3
    line = trk.getTimeLogString()
    for word in ("nTrack", "Total", "Process(", "Cluster(", "Optim(", "DynN", "N-Prune", "Kill", "Init"):
        assert word in line
    assert line.startswith("3 ") and "Process(%4.0f+0  /" % len(sc["scans"][-1]) in line
    assert "Num Targets" in trk.getTimeLogHeader()
    trk.printTimeLogHeader(); trk.printTargetList()
    assert "TargetList:" in capsys.readouterr().out
    trk.close()


def test_report_two_scans_back_raw_abi():
    """C-ABI order of a streaming host with the device initiator: mht_forest_scan(k) queues scan k, the report of scan k - 1 leaves
    the device inside scan k's grow launch -- a host that folds the report of scan k - 2 behind its call for scan k
    (mht_forest_report_get(which = 2)) never waits for that launch.  Every report is that of its scan; which = 2 is refused
    (MHT_E_STATE) once the last scan's report has been flushed to the host block of the same parity."""
    import ctypes as C
    from pymht_amd import _lib
    sc = _scenario(T=30, n_scans=7, seed=12)
    trk = _mk(sc, N=3)
    lib, h, ih = trk._lib, trk._ctx.handle, trk.initiator.handle
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    z = [np.ascontiguousarray(s_, dtype=np.float32) for s_ in sc["scans"]]

    from pymht_amd.tracker import _REPORT_DTYPE

    def get(which):
        rep = _lib.MhtScanReport()
        rc = lib.mht_forest_report_get(h, which, C.byref(rep))
        if rc == 0 and rep.n_targets:      # the rows: compacted index and leaves kept are consistent with the statuses (whoever pushed the rows)
            recs = np.ctypeslib.as_array(C.cast(rep.targets, C.POINTER(C.c_uint8)), shape=(rep.n_targets * _REPORT_DTYPE.itemsize,)).view(_REPORT_DTYPE)
            alive = recs["status"] == 0
            assert int(alive.sum()) == rep.n_alive
            assert np.array_equal(recs["new_index"], np.where(alive, np.cumsum(alive) - 1, -1))
            assert np.all(recs["n_leaves"][alive] >= 1) and np.all(recs["n_leaves"][~alive] == 0)
            assert int(recs["n_leaves"].sum()) == rep.n_leaves_out
        return rc, rep.scan, rep.n_targets, rep.n_alive

    seen = {}
    for k in range(1, 7):
        _lib.check(lib.mht_forest_scan(h, ih, p(z[k - 1]), len(z[k - 1]), float(sc["times"][k - 1])))
        if k >= 3:
            rc, s_, nT, na = get(2)
            assert rc == 0 and s_ == k - 2, (k, rc, s_)
            seen[s_] = (nT, na)
        else:
            assert get(2)[0] == _lib.MHT_E_STATE          # (no scan two back yet)
    rc, s_, nT, na = get(1)
    assert rc == 0 and s_ == 5
    rc, s_, nT6, na6 = get(0)                              # (flushes scan 6's commit, admission and report)
    assert rc == 0 and s_ == 6 and nT6 >= na6 > 0
    assert get(2)[0] == _lib.MHT_E_STATE                   # scan 4's block now holds scan 6
    assert sorted(seen) == [1, 2, 3, 4] and all(nT >= na > 0 for nT, na in seen.values())
    # the tracker object is still usable: its own bookkeeping was bypassed, so only the library is exercised from here on
    _lib.check(lib.mht_forest_scan(h, ih, p(z[6]), len(z[6]), float(sc["times"][6])))
    rc, s_, _, _ = get(0)
    assert rc == 0 and s_ == 7
    trk._pendq = []
    trk.close()


def test_ilp_grid_hint_survives_a_bulk_admission():
    """The ILP launch is sized by the LAST scan's cluster counts (a host-mapped hint word; mht_forest.hip) -- a workgroup's tables hold 4
    multi-target and 32 single-target clusters, so a stale small hint must not size a launch for many more targets than it was taken
    from (advisor, round 5: after a few one-target scans a bulk `initiateTarget` left 32 workgroups for 1 300 targets and the clusters
    beyond the tables kept stale selections).  One target for three scans, then 1 299 more at once, then three more scans: every scan
    against the oracle (gating counts, selections, states, clusters, leaf sets)."""
    from test_tracker_gpu import tracker_selected, states_close, SCORE_ATOL
    from trace_util import make_oracle
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_scenario
    import mht_oracle as orc
    sc = make_scenario(T=1300, radius=9000.0, lambda_phi=2e-8, n_scans=6, P_d=0.9, period=2.5, seed=4711)
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=5.99, useInitiator=False, maxTargets=2048, maxNodes=1 << 17, maxMeasurements=2048)
    try:
        first = [Target(sc["t0"], None, sc["x0"][0].copy(), pv.P0, status="preinitialized")]
        assert len(trk._add_targets(first)) == 1
        g = dict(period=sc["period"], lambda_phi=sc["lambda_phi"], lambda_nu=1e-4, P_d=sc["P_d"], N=3, eta2=5.99, x0=sc["x0"][:1], t0=sc["t0"], accepted=[True])
        o = make_oracle(g, with_initiator=False)
        for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
            if k == 3:      # the bulk admission: the other targets where they are now
                xs = sc["truth"][2][1:]
                cands = [Target(float(sc["times"][2]), None, x.copy(), pv.P0, status="preinitialized") for x in xs]
                admitted = set(id(c) for c in trk._add_targets(cands))
                acc = [o.initiate_target(float(sc["times"][2]), x.copy(), orc.model_P0(), status="preinitialized") for x in xs]
                assert acc == [id(c) in admitted for c in cands] and sum(acc) > 1250
            info = o.add_scan(float(t), z)
            trk.addMeasurementList(MeasurementList(float(t), z))
            st = trk.lastScanStats
            os_, ts = o.selected(), tracker_selected(trk)
            assert (st["L"], st["G"]) == (info["L"], info["G"]) and np.array_equal(st["unused"], info["unused"]), k
            assert np.array_equal(os_["ID"], ts["ID"]) and np.array_equal(os_["meas"], ts["meas"]), k
            assert states_close(os_["x"], ts["x"]) and np.allclose(os_["cnllr"], ts["cnllr"], rtol=0, atol=SCORE_ATOL), k
            assert len(o.clusters) == len(trk.__clusterList__) and o.n_ilp == trk.nOptimSolved, k
            lb, tb = o.leaf_batch(), trk.leafBatch()
            assert np.array_equal(lb["ID"], tb["ID"]) and np.array_equal(lb["meas"], tb["meas"]), k
        assert trk.nTargets > 1200
    finally:
        trk.close()


def test_more_births_than_the_initiator_holds_is_a_capacity_error(monkeypatch):
    """Round 6's fuzz campaign (seed 1695381: 124 confirmed tracks in one scan against max_born = 128 before merging) showed the
    candidates beyond the initiator's capacity dropped WITHOUT an error when the initiator runs behind the forest's scan: only the
    stand-alone seam (mht_initiator_born) read InitDev::overflow.  Now the forest's sticky flag takes it: MHT_E_CAPACITY, dead forest."""
    from pymht_amd import _lib
    from pymht_amd.initiators import m_of_n
    from pymht_amd.utils.classDefinitions import MeasurementList
    monkeypatch.setattr(m_of_n, "MAX_BORN", 3)
    sc = _scenario(T=2, radius=300.0, lambda_phi=1e-6, n_scans=10, seed=5)
    trk = _mk(sc, N=3)
    p0 = np.array([[x, y] for y in (-150.0, 150.0) for x in (-180.0, -60.0, 60.0, 180.0)], np.float32)
    v = np.array([4.0, 1.0], np.float32)
    with pytest.raises(_lib.MhtError) as ei:
        for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):      # eight steady strangers: preliminary tracks in the second scan, confirmed in the third
            zz = np.concatenate([np.asarray(z, np.float32).reshape(-1, 2), p0 + v * np.float32(t - sc["times"][0])]).astype(np.float32)
            trk.addMeasurementList(MeasurementList(float(t), zz))
        trk.getTrackNodes()      # (the scan behind the overflowing initiator is void; its report is folded two calls later or by the first look)
    assert ei.value.code in (_lib.MHT_E_CAPACITY, _lib.MHT_E_STATE)
    trk.close()


def test_a_hundred_births_in_one_scan_match_the_oracle():
    """The scene that found it: four targets in ~380 clutter points per scan (lambda 1.5e-4, dt 4 s): 41 births in the fourth scan, 123
    admitted of 124 in the fifth -- target lists, states and selections against the oracle on every scan (tests/fuzz_util.py)."""
    from fuzz_util import run_case
    ok, desc, msg = run_case(1695381)
    assert ok, (desc, msg)


def test_chains_gathered_in_one_launch_equal_the_synchronous_walk():
    """mht_forest_chains_begin / _fetch (the window chains of terminated tracks, gathered without stopping the stream) against
    mht_forest_chain node by node: 400 leaves at once (a block that outgrows the 128 KB pinned at creation), float32 and float64 form,
    a ticket that is fetched late (behind two more scans) and one that has expired (eight later tickets)."""
    import ctypes as C
    from pymht_amd import _lib
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = _scenario(T=60, radius=500.0, lambda_phi=4e-5, n_scans=9, P_d=0.8, seed=11)
    trk = _mk(sc, N=4, useInitiator=False)
    lib, h = trk._lib, trk._ctx.handle
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def sync_chain(scan, node, n_max):
        nodes, meas, x, cn, P, n = np.zeros(n_max, np.int32), np.zeros(n_max, np.int32), np.zeros((n_max, 4)), np.zeros(n_max), np.zeros((n_max, 16), np.float32), C.c_int32(0)
        _lib.check(lib.mht_forest_chain(h, scan, node, n_max, p(nodes), p(meas), p(x), p(cn), p(P), C.byref(n)))
        k = n.value
        return nodes[:k].copy(), meas[:k].copy(), x[:k].copy(), cn[:k].copy(), P[:k].copy()

    def fetch(ticket, i, n_max, f64):
        nodes, meas, x, cn, n = np.zeros(n_max, np.int32), np.zeros(n_max, np.int32), np.zeros((n_max, 4)), np.zeros(n_max), C.c_int32(0)
        P, fl = np.zeros((n_max, 16), np.float64 if f64 else np.float32), np.zeros(n_max, np.uint8)
        _lib.check(lib.mht_forest_chains_fetch(h, ticket, i, p(nodes), p(meas), p(x), p(cn), p(P), p(fl), C.byref(n)))
        k = n.value
        return nodes[:k].copy(), meas[:k].copy(), x[:k].copy(), cn[:k].copy(), P[:k].copy()

    for z, t in zip(sc["scans"][:6], sc["times"][:6]):
        trk.addMeasurementList(MeasurementList(float(t), z))
    scan = len(trk.__scanHistory__)
    start = np.ascontiguousarray(trk.leafBatch()["node"][:400], dtype=np.int32)
    assert len(start) >= 300
    n_max = 6
    want = [sync_chain(scan, int(nd), n_max) for nd in start]
    t32, t64 = C.c_int64(-1), C.c_int64(-1)
    _lib.check(lib.mht_forest_chains_begin(h, scan, p(start), len(start), n_max, 0, C.byref(t32)))
    _lib.check(lib.mht_forest_chains_begin(h, scan, p(start), len(start), n_max, 1, C.byref(t64)))
    for z, t in zip(sc["scans"][6:8], sc["times"][6:8]):      # (two more scans are queued and run before anybody looks)
        trk.addMeasurementList(MeasurementList(float(t), z))
    for i in (0, 1, 57, len(start) - 1):
        for got, tk in ((fetch(t32.value, i, n_max, False), "f32"), (fetch(t64.value, i, n_max, True), "f64")):
            assert len(got[0]) == len(want[i][0]) >= 2, (i, tk)
            for a_, b_ in zip(got[:4], want[i][:4]):
                assert np.array_equal(a_, b_), (i, tk)
            assert np.array_equal(got[4].astype(np.float32), want[i][4]), (i, tk)
    # eight later tickets: the first one's block has been reused
    scan = len(trk.__scanHistory__)
    one = np.ascontiguousarray(trk.leafBatch()["node"][:3], dtype=np.int32)
    tk = C.c_int64(-1)
    for _ in range(8):
        _lib.check(lib.mht_forest_chains_begin(h, scan, p(one), 3, n_max, 0, C.byref(tk)))
    with pytest.raises(_lib.MhtError) as ei:
        fetch(t32.value, 0, n_max, False)
    assert ei.value.code == _lib.MHT_E_INVALID and "expired" in str(ei.value)
    assert len(fetch(tk.value, 2, n_max, False)[0]) >= 2
    trk.close()
