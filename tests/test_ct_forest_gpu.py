"""GPU: the CONSTANT-TURN forest (BASELINE config 5's model as named: pymht_amd/models/ct.py, six states [x, y, vx, vy, w, a]) behind the
Tracker API: `Tracker(ct, ...)` makes a forest with MHT_FOREST_CT (libmht_amd6.so) in which every hypothesis carries its own transition
Phi(T, w) and its own covariance chain -- the reference's per-hypothesis form kalman.predict_single + kalman.precalc on a batch of one
(kalman.py:67-70, :82-101), which is also what the stateless seam does with mht_model_x.transition = 1 (G21).

g23 was recorded with the oracle tracker whose Kalman steps were the REFERENCE's own kalman functions (oracle/gen_golden.py::gen_g23: the
reference ships no six-state model).  Replayed scan by scan: gating counts, unused measurements, clusters, selections,
terminations, leaf sets exactly; states and covariances to 1e-6 relative -- the device's sin / cos of w T may round Phi's float32 entries
one ulp away from NumPy's (tests/test_gatex_gpu.py says the same of the seam)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCORE_ATOL = 2e-5


def _close(a, b, rel=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    if a.size == 0:
        return True
    flat_a, flat_b = a.reshape(len(a), -1), b.reshape(len(b), -1)
    scale = np.maximum(np.abs(flat_b).max(axis=1, keepdims=True), 1.0)
    return bool(np.all(np.abs(flat_a - flat_b) <= rel * scale))


def _make(g, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import ct
    trk = Tracker(ct, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]), eta2=float(g["eta2"]),
                  useInitiator=False, **kw)
    acc = []
    for x in g["x0"]:
        n0 = trk.nTargets
        trk.initiateTarget(Target(float(g["t0"]), None, np.array(x, dtype=np.float64), ct.P0, status="preinitialized"))
        acc.append(trk.nTargets > n0)
    return trk, acc


@pytest.mark.parametrize("spill", [False, True])
def test_constant_turn_forest_replays_trace(gold_dir, spill, monkeypatch):
    """spill: the grow kernel of the constant-turn forest keeps a leaf's hits as bits over the target's candidate list (a few LDS words per leaf);
    a target with more candidates than those words hold takes full-width masks in a block of global memory -- forced for every target here
    (MHT_CT_SPILL=1, read when the forest is created)."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, "g23_trace_ct6.npz"))
    if spill:
        monkeypatch.setenv("MHT_CT_SPILL", "1")
    trk, acc = _make(g)
    monkeypatch.delenv("MHT_CT_SPILL", raising=False)
    assert trk.nx == 6 and acc == [bool(a) for a in g["accepted"]]
    n_ilp, worst = 0, 0.0
    for k in range(int(g["n_scans"])):
        p = "s%02d_" % k
        ids_before = [r.ID for r in trk.__targetList__]
        trk.addMeasurementList(MeasurementList(float(g["times"][k]), g[p + "z"]))
        st = trk.lastScanStats
        assert [st["L"], st["G"], st["M"]] == g[p + "LGM"].tolist(), "scan %d L/G/M" % k
        assert np.array_equal(st["unused"], g[p + "unused"]), k
        nodes = list(trk.getTrackNodes())
        assert np.array_equal([n.ID for n in nodes], g[p + "sel_ID"]), k
        assert np.array_equal([0 if n.measurementNumber is None else n.measurementNumber for n in nodes], g[p + "sel_meas"]), k
        assert _close(np.array([n.x_0 for n in nodes]).reshape(-1, 6), g[p + "sel_x"]), k
        assert np.allclose([float(n.cumulativeNLLR) for n in nodes], g[p + "sel_cnllr"], rtol=0, atol=SCORE_ATOL), k
        ids_after = np.array([r.ID for r in trk.__targetList__])
        assert np.array_equal(ids_after, g[p + "ids"]), k
        assert sorted(i for i in ids_before if i not in ids_after.tolist()) == g[p + "dead"].tolist()
        ptr, mem = g[p + "cl_ptr"], g[p + "cl_members"]
        cl = trk.__clusterList__
        assert len(cl) == len(ptr) - 1 and all(np.array_equal(np.asarray(c), mem[ptr[i]:ptr[i + 1]]) for i, c in enumerate(cl)), k
        assert trk.nOptimSolved == int(g[p + "n_ilp"])
        leaf = trk.leafBatch()
        assert leaf["x"].shape[1] == 6 and leaf["P"].shape[1:] == (6, 6)
        assert np.array_equal(leaf["ID"], g[p + "leaf_ID"]) and np.array_equal(leaf["meas"], g[p + "leaf_meas"]), k
        assert _close(leaf["x"], g[p + "leaf_x"]), (k, "leaf states")
        assert _close(leaf["P"], g[p + "leaf_P"]), (k, "leaf covariances")
        assert np.allclose(leaf["cnllr"], g[p + "leaf_cnllr"], rtol=0, atol=SCORE_ATOL), k
        worst = max(worst, float(np.abs(leaf["x"] - g[p + "leaf_x"]).max()))
        n_ilp += trk.nOptimSolved
    assert n_ilp > 20
    # the ancestors of a leaf come back with their own covariances (per-node storage: key -> parent layer)
    sel = list(trk.getTrackNodes())[0]
    chain = [sel]
    while chain[-1].parent is not None and len(chain) < 3:
        chain.append(chain[-1].parent)
    assert all(np.asarray(c.P_0).shape == (6, 6) and np.all(np.isfinite(c.P_0)) for c in chain)
    print("worst absolute leaf-state difference", worst)
    trk.close()


def test_constant_turn_forest_similar_state_pruning():
    """Similar-state pruning (tracker.py:1233-1239, pyTarget.py:358-412 are model-agnostic) in a constant-turn forest: the merged node takes the
    missed-detection child's slot and key, and a mean covariance that is not the hit children's own is written over the parent's P_bar entry
    (csrc/mht_similar.hip).  Random scenarios with the switch toggled per scan against the live oracle: decisions exact, states AND covariances of
    all leaves 1e-6 (fuzz_util.run_case_ct(similar=True))."""
    from fuzz_util import run_case_ct
    bad, merged_seen = [], 0
    for case in range(40):
        ok, desc, msg = run_case_ct(880000 + case, max_leaves=1500, budget_s=6.0, similar=True)
        if not ok:
            bad.append(desc + ' ' + msg)
    assert not bad, "\n".join(bad)
