"""GPU: the AIS-aided path through the C ABI.  `mht_fuse_ais` (Tracker.__fuseRadarAndAis, tracker.py:417-552) against the
known-answer vectors recorded from the reference (G19): the same children in the same order -- which messages gate, which radar
measurements gate behind them, pure-AIS children, the identity filter -- and their states / covariances BIT FOR BIT (float64 dgemm chains
and LAPACK dgesv restated operation by operation, csrc/mht_la64.h), scores to the NLLR tolerance of tests/ais_util.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fuse_seam_matches_reference_vectors(gpu_ctx, gold_dir):
    from ais_util import g19_case, check_children
    from pymht_amd.device import fuse_radar_and_ais
    from pymht_amd.models import pv
    g = np.load(os.path.join(gold_dir, "g19_ais_fusion.npz"))
    lam = float(g["lambda_phi"]) + float(g["lambda_nu"])
    total = 0
    for ci in range(int(g["n_cases"])):
        c = g19_case(g, ci)
        n = len(c["x"])
        flags = np.where(c["xf32"], 1, 0).astype(np.uint8)          # MHT_F_STATE_F32
        for own_of in (None, int(c["msgs"][0].mmsi)):
            own = np.zeros(n, dtype=np.int32) if own_of is None else np.full(n, own_of, dtype=np.int32)
            r = fuse_radar_and_ais(gpu_ctx, pv, float(g["eta2"]), lam, c["x"], c["P"], c["pd"], flags, own, c["msgs"], float(g[c["p"] + "t_leaf"]),
                                   float(g[c["p"] + "t_scan"]), c["eta2_ais"], c["lambda_ais"], c["z"])
            for l in range(n):
                a, b = int(r["child_ptr"][l]), int(r["child_ptr"][l + 1])
                if own_of is None:
                    check_children(c, l, r["x"][a:b], r["P"][a:b], r["radar"][a:b], r["nllr"][a:b], r["mmsi"][a:b])
                    total += b - a
                else:          # only that ship's messages: the reference's children with the others filtered out
                    ra, rb = int(c["ptr"][l]), int(c["ptr"][l + 1])
                    keep = c["out_mmsi"][ra:rb] == own_of
                    assert np.array_equal(r["radar"][a:b], c["out_radar"][ra:rb][keep]) and np.all(r["mmsi"][a:b] == own_of)
                    assert np.array_equal(r["x"][a:b], c["out_x"][ra:rb][keep])
    assert total == 590


def test_fuse_seam_edge_cases(gpu_ctx):
    from pymht_amd.device import fuse_radar_and_ais
    from pymht_amd.ais import AisMessage
    from pymht_amd.models import pv
    x = np.array([[10.0, 20.0, 1.0, -1.0]])
    P = pv.P0[None].astype(np.float32)
    z = np.zeros((0, 2), dtype=np.float32)
    # no messages: no children; no leaves: an empty CSR
    r = fuse_radar_and_ais(gpu_ctx, pv, 5.99, 1e-4, x, P, [0.9], [0], [0], [], 0.0, 2.5, 9.45, 1e-7, z)
    assert list(r["child_ptr"]) == [0, 0]
    r = fuse_radar_and_ais(gpu_ctx, pv, 5.99, 1e-4, np.zeros((0, 4)), np.zeros((0, 4, 4), dtype=np.float32), [], [], [], [AisMessage(1.0, x[0], 257000001, True)],
                           0.0, 2.5, 9.45, 1e-7, z)
    assert list(r["child_ptr"]) == [0]
    # a message on top of the leaf, no radar measurement at all: one pure-AIS child
    r = fuse_radar_and_ais(gpu_ctx, pv, 5.99, 1e-4, x, P, [0.9], [0], [0], [AisMessage(1.0, [11.0, 19.0, 1.0, -1.0], 257000001, True)], 0.0, 2.5, 9.45, 1e-7, z)
    assert list(r["child_ptr"]) == [0, 1] and list(r["radar"]) == [-1] and list(r["mmsi"]) == [257000001]


# ---- the AIS-aided path through the forest and the drop-in Tracker ---------------------------------------------------------------
# Everything is compared exactly: which children exist, in which order, with which radar measurement and which identity; clusters;
# selections; target lists; births and terminations -- and the states and covariances of ALL leaves BIT FOR BIT, in the dtype the
# reference gives them: an AIS-updated node's covariance is float64 (models/ais.py:4: ais.C is float64), and NumPy promotes the target's
# whole batch from the next scan on (np.array of the leaves' x_0 / P_0, tracker.py:859-870).  The forest carries such covariances as
# float64 values of its table (csrc/mht_vtab.h, MHT_F_COV_F64).  Only the cumulative scores have a tolerance (the NLLR constant's log).
SCORE_ATOL = 2e-5     # cumulative scores (NLLR constant: float32 log, see test_tracker_gpu.py)


@pytest.mark.parametrize("name", ["g18_trace_ais_cfg1", "g18b_trace_ais_dense", "g18c_trace_ais_n5", "g18d_trace_ais_similar",
                                  # the reference's default: messages no track took start tracks (aisInitialization=True, m_of_n.py:262-280)
                                  "g18e_trace_ais_init", "g18f_trace_ais_init_dense"])
def test_tracker_replays_reference_ais_trace(name, gold_dir):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.ais import AisMessage, AisMessageList
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, name + ".npz"))
    trk = Tracker(pv, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]), eta2=float(g["eta2"]),
                  eta2_ais=float(g["eta2_ais"]), radarRange=float(g["radar_range"]), position=g["position"], aisAided=True,
                  useInitiator=bool(g["with_initiator"]), maxTargets=256, maxNodes=1 << 16, maxMeasurements=256)
    try:
        for x, ok in zip(g["x0"], g["accepted"]):
            n0 = trk.nTargets
            trk.initiateTarget(Target(float(g["t0"]), None, x.copy(), pv.P0, status="preinitialized"))
            assert (trk.nTargets > n0) == bool(ok)
        n_fused = n_pure = 0
        for k in range(int(g["n_scans"])):
            p = "s%02d_" % k
            msgs = AisMessageList([AisMessage(float(t), s, int(m), bool(h)) for t, s, m, h in
                                   zip(g[p + "ais_time"], g[p + "ais_state"], g[p + "ais_mmsi"], g[p + "ais_high"])])
            trk.addMeasurementList(MeasurementList(float(g["times"][k]), g[p + "z"]), msgs, aisInitialization=bool(g["ais_init"]) if "ais_init" in g.files else False,
                                   pruneSimilar=bool(g["prune_similar"]))
            nodes = list(trk.getTrackNodes())
            assert np.array_equal([n.ID for n in nodes], g[p + "sel_ID"]), k
            assert np.array_equal([r.ID for r in trk.__targetList__], g[p + "ids"]), k
            meas = np.array([-1 if n.measurementNumber is None else n.measurementNumber for n in nodes], dtype=np.int64)
            mmsi = np.array([0 if n.mmsi is None else n.mmsi for n in nodes], dtype=np.int64)
            assert np.array_equal(meas, g[p + "sel_meas"]) and np.array_equal(mmsi, g[p + "sel_mmsi"]), (k, meas, g[p + "sel_meas"], mmsi, g[p + "sel_mmsi"])
            assert np.array_equal(np.array([np.asarray(n.x_0, dtype=np.float64) for n in nodes]).reshape(-1, 4), g[p + "sel_x"]), (k, "selected states (bit for bit)")
            assert np.array_equal(np.array([np.asarray(n.P_0, dtype=np.float64) for n in nodes]).reshape(-1, 4, 4), g[p + "sel_P"]), (k, "selected covariances (bit for bit)")
            assert np.array_equal([np.asarray(n.P_0).dtype == np.float64 for n in nodes], g[p + "sel_Pf64"]), (k, "dtype of the selected covariances")
            assert np.allclose([float(n.cumulativeNLLR) for n in nodes], g[p + "sel_cnllr"], rtol=0, atol=SCORE_ATOL), k
            st = trk.lastScanStats
            assert st["L"] == int(g[p + "LGM"][0]) and np.array_equal(st["unused"], g[p + "unused"]), k
            ptr, mem = g[p + "cl_ptr"], g[p + "cl_members"]
            cl = trk.__clusterList__
            assert len(cl) == len(ptr) - 1 and all(np.array_equal(np.asarray(c), mem[ptr[i]:ptr[i + 1]]) for i, c in enumerate(cl)), k
            lb = trk.leafBatch()
            assert np.array_equal(lb["ID"], g[p + "leaf_ID"]) and np.array_equal(lb["meas"], g[p + "leaf_meas"]), k
            assert np.array_equal(lb["mmsi"], g[p + "leaf_mmsi"]), k
            assert np.array_equal(lb["x"], g[p + "leaf_x"]), (k, "leaf states (bit for bit): %d rows differ" % int(np.any(lb["x"] != g[p + "leaf_x"], axis=1).sum()))
            assert np.array_equal(lb["Pf64"], g[p + "leaf_Pf64"]), (k, "which leaves carry a float64 covariance")
            assert np.array_equal(lb["P"], g[p + "leaf_P"]), (k, "leaf covariances (bit for bit): %d differ" % int(np.any(lb["P"] != g[p + "leaf_P"], axis=(1, 2)).sum()))
            assert np.allclose(lb["cnllr"], g[p + "leaf_cnllr"], rtol=0, atol=SCORE_ATOL), k
            n_fused += int((lb["mmsi"] != 0).sum())
            n_pure += int((lb["meas"] < 0).sum())
        assert n_fused > 50          # (the trace does exercise fused children)
    finally:
        trk.close()


@pytest.mark.parametrize("name", ["g18b_trace_ais_dense", "g18c_trace_ais_n5"])
def test_identities_on_ancestor_views(name, gold_dir):
    """`Target.mmsi` of EVERY node of a track's history (pyTarget.py:34), not only of the selected nodes and the leaves: the ancestors inside
    the device window and the committed roots behind it carry the identity of the AIS message they were updated with (and a measurement
    number of None where they have no radar measurement, tracker.py:520), as the oracle's trees -- pinned bit for bit to the reference by
    tests/test_oracle_golden.py on the same fixture -- have them."""
    from trace_util import make_oracle_ais, ais_messages
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.ais import AisMessage, AisMessageList
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, name + ".npz"))
    trk = Tracker(pv, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]), eta2=float(g["eta2"]),
                  eta2_ais=float(g["eta2_ais"]), radarRange=float(g["radar_range"]), position=g["position"], aisAided=True,
                  useInitiator=bool(g["with_initiator"]), maxTargets=256, maxNodes=1 << 16, maxMeasurements=256)
    o = make_oracle_ais(g)
    try:
        for x in g["x0"]:
            trk.initiateTarget(Target(float(g["t0"]), None, x.copy(), pv.P0, status="preinitialized"))
        n_hist_ids = 0
        for k in range(int(g["n_scans"])):
            p = "s%02d_" % k
            msgs = ais_messages(g, k)
            o.add_scan(float(g["times"][k]), g[p + "z"], ais=msgs, prune_similar=bool(g["prune_similar"]),
                       ais_initialization=bool(g["ais_init"]) if "ais_init" in g.files else False)
            trk.addMeasurementList(MeasurementList(float(g["times"][k]), g[p + "z"]),
                                   AisMessageList([AisMessage(float(m.time), m.state, int(m.mmsi), bool(m.highAccuracy)) for m in msgs]),
                                   aisInitialization=bool(g["ais_init"]) if "ais_init" in g.files else False, pruneSimilar=bool(g["prune_similar"]))
            if k % 3 != 2 and k != int(g["n_scans"]) - 1:
                continue
            nodes = list(trk.getTrackNodes())
            assert [n.ID for n in nodes] == [n.ID for n in o.track_nodes]
            for tn, on in zip(nodes, o.track_nodes):
                got, want = [], []
                a = tn
                while a is not None:
                    got.append((int(a.scanNumber), a.measurementNumber if a.measurementNumber is None else int(a.measurementNumber), a.mmsi))
                    a = a.parent
                b = on
                while b is not None:
                    want.append((int(b.scan), b.meas if b.meas is None else int(b.meas), b.mmsi))
                    b = b.parent
                assert got == want, (k, tn.ID, got, want)
                n_hist_ids += sum(1 for q in got[1:] if q[2] is not None)
        assert n_hist_ids > 20      # (identities were found on ancestors, inside and behind the window)
    finally:
        trk.close()


@pytest.mark.parametrize("name", ["g18b_trace_ais_dense", "g18f_trace_ais_init_dense", "g18d_trace_ais_similar"])
def test_ais_trace_across_value_table_generations(name, gold_dir, monkeypatch):
    """float64 covariances are values of the forest's table like the float32 ones (two ids each, csrc/mht_vtab.h).  These streams take 1-2 k ids
    per scan (a fused child's covariance and its pseudo parent; the per-leaf values of promoted targets); with a table of 16 384 ids the host
    sees it filling after four scans and the live leaves -- float32 and float64 ones, pseudo-parent keys of fused children and of merged
    hypotheses included -- are re-keyed into the other generation (vt_rebuild_kernel; a switch at most every R + 2 scans, so a smaller table
    would overflow before the second one).  The trace recorded from the reference must still come out bit for bit."""
    import ctypes as C
    from pymht_amd import _lib
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.ais import AisMessage, AisMessageList
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, name + ".npz"))
    monkeypatch.setenv("MHT_VTAB_CAP", "16384")
    trk = Tracker(pv, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]), eta2=float(g["eta2"]),
                  eta2_ais=float(g["eta2_ais"]), radarRange=float(g["radar_range"]), position=g["position"], aisAided=True,
                  useInitiator=bool(g["with_initiator"]), maxTargets=256, maxNodes=1 << 16, maxMeasurements=256)
    monkeypatch.delenv("MHT_VTAB_CAP")
    try:
        for x in g["x0"]:
            trk.initiateTarget(Target(float(g["t0"]), None, x.copy(), pv.P0, status="preinitialized"))
        for k in range(int(g["n_scans"])):
            p = "s%02d_" % k
            msgs = AisMessageList([AisMessage(float(t), s, int(m), bool(h)) for t, s, m, h in
                                   zip(g[p + "ais_time"], g[p + "ais_state"], g[p + "ais_mmsi"], g[p + "ais_high"])])
            trk.addMeasurementList(MeasurementList(float(g["times"][k]), g[p + "z"]), msgs, aisInitialization=bool(g["ais_init"]) if "ais_init" in g.files else False,
                                   pruneSimilar=bool(g["prune_similar"]))
            lb = trk.leafBatch()
            assert np.array_equal(lb["ID"], g[p + "leaf_ID"]) and np.array_equal(lb["meas"], g[p + "leaf_meas"]) and np.array_equal(lb["mmsi"], g[p + "leaf_mmsi"]), k
            assert np.array_equal(lb["x"], g[p + "leaf_x"]) and np.array_equal(lb["Pf64"], g[p + "leaf_Pf64"]) and np.array_equal(lb["P"], g[p + "leaf_P"]), k
            assert np.array_equal([r.ID for r in trk.__targetList__], g[p + "ids"]), k
        r = np.zeros(1, np.int32)
        _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, b"vt_rebuilds", r.ctypes.data_as(C.c_void_p), 4))
        assert r[0] >= 1, "the table was meant to fill: %d generation switches" % int(r[0])
    finally:
        trk.close()


def test_fuse_seam_float64_covariances_match_live_oracle(gpu_ctx, gold_dir):
    """`mht_fuse_ais_f64`: leaves the reference carries in float64 (node.P_0 float64 behind an AIS update, tracker.py:449-450 runs
    kalman.predict_single on it as it is).  The G19 leaves with their covariances perturbed to values float32 cannot hold, against the
    oracle's `fuse_radar_ais` (the reference's NumPy expressions) evaluated HERE: children and order exact; states and covariances bit for bit
    where this host's numpy runs the kernel set csrc/mht_la64.h restates, 1e-12 otherwise."""
    import mht_oracle as orc
    from ais_util import g19_case
    from util import live_numpy_f64_is_pinned
    from pymht_amd.device import fuse_radar_and_ais
    from pymht_amd.models import pv
    g = np.load(os.path.join(gold_dir, "g19_ais_fusion.npz"))
    lam = float(g["lambda_phi"]) + float(g["lambda_nu"])
    exact = live_numpy_f64_is_pinned()
    rng = np.random.default_rng(64)
    total = 0
    for ci in range(min(int(g["n_cases"]), 6)):
        c = g19_case(g, ci)
        n = len(c["x"])
        keep = [l for l in range(n) if not c["xf32"][l]]      # (a float64-covariance node has a float64 state)
        if not keep:
            continue
        x = np.asarray(c["x"], dtype=np.float64)[keep]
        P = np.asarray(c["P"], dtype=np.float64)[keep]
        P = P * (1.0 + 1e-9 * rng.uniform(-1, 1, size=(len(keep), 1, 1))) + np.eye(4)[None] * 1e-7 * rng.uniform(0, 1, size=(len(keep), 1, 1))
        pd = np.asarray(c["pd"], dtype=np.float64)[keep]
        flags = np.full(len(keep), 16, dtype=np.uint8)          # MHT_F_COV_F64
        t_leaf, t_scan = float(g[c["p"] + "t_leaf"]), float(g[c["p"] + "t_scan"])
        r = fuse_radar_and_ais(gpu_ctx, pv, float(g["eta2"]), lam, x, P, pd, flags, np.zeros(len(keep), dtype=np.int32), c["msgs"], t_leaf, t_scan,
                               c["eta2_ais"], c["lambda_ais"], c["z"])

        class Leaf:
            pass
        leaves = []
        for i in range(len(keep)):
            lf = Leaf()
            lf.time, lf.x, lf.P, lf.P_d = t_leaf, x[i], P[i], float(pd[i])
            leaves.append(lf)
        msgs = [orc.AisMessage(float(m.time), np.asarray(m.state, dtype=np.float64), int(m.mmsi), bool(m.highAccuracy)) for m in c["msgs"]]
        want = orc.fuse_radar_ais(leaves, msgs, c["z"], t_scan, orc.model_C(), orc.model_R(), float(g["eta2"]), c["eta2_ais"], lam, c["lambda_ais"])
        for i, kids in enumerate(want):
            a, b = int(r["child_ptr"][i]), int(r["child_ptr"][i + 1])
            assert b - a == len(kids), (ci, i)
            for q, (xk, Pk, rk, nk, mk) in enumerate(kids):
                assert (r["radar"][a + q] == (-1 if rk is None else rk)) and r["mmsi"][a + q] == mk
                if exact:
                    assert np.array_equal(r["x"][a + q], xk) and np.array_equal(r["P"][a + q], Pk), (ci, i, q)
                else:
                    assert np.allclose(r["x"][a + q], xk, rtol=1e-12, atol=1e-12) and np.allclose(r["P"][a + q], Pk, rtol=1e-12, atol=1e-12)
                assert abs(r["nllr"][a + q] - nk) < 1e-11
            total += len(kids)
    assert total > 100
