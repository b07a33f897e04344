"""GPU: the six-state forest (libmht_amd6.so = the library's sources compiled with -DMHT_NX=6; BASELINE config 5's state dimension)
behind the same Tracker API, with the constant-acceleration model pymht_amd/models/ca.py.

The reference's tracker is hard-wired to its 4-state model; its kalman module is dimension-generic.  The fixture g17 was recorded
with the oracle tracker whose Kalman steps were the REFERENCE's own kalman.predict / precalc / z_tilde / NIS / numpyFilter / nllr
(oracle/gen_golden.py::gen_g17); it is replayed here scan by scan: gating counts, unused measurements, clusters, selections,
terminations exactly, states and covariances of ALL leaves bit for bit."""
import os

import numpy as np
import pytest

from trace_util import check_scan_against_fixture

pytestmark = pytest.mark.gpu
SCORE_ATOL = 2e-5


def make_tracker6(g, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import ca
    trk = Tracker(ca, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]), eta2=float(g["eta2"]),
                  useInitiator=False, **kw)
    acc = []
    for x in g["x0"]:
        n0 = len(trk.__targetList__)
        trk.initiateTarget(Target(float(g["t0"]), None, np.array(x, dtype=np.float64), ca.P0, status="preinitialized"))
        acc.append(len(trk.__targetList__) > n0)
    return trk, acc


def selected6(trk):
    nodes = list(trk.getTrackNodes())
    return dict(ID=np.array([n.ID for n in nodes], dtype=np.int64),
                x=np.array([np.asarray(n.x_0, dtype=np.float64) for n in nodes]).reshape(-1, 6),
                cnllr=np.array([float(n.cumulativeNLLR) for n in nodes]),
                meas=np.array([0 if n.measurementNumber is None else n.measurementNumber for n in nodes], dtype=np.int64))


def test_six_state_forest_replays_trace(gold_dir):
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, "g17_trace_6state.npz"))
    assert int(g["nx"]) == 6
    trk, acc = make_tracker6(g)
    assert trk.nx == 6 and acc == [bool(a) for a in g["accepted"]]
    n_ilp = 0
    for k in range(int(g["n_scans"])):
        p = "s%02d_" % k
        ids_before = [r.ID for r in trk.__targetList__]
        trk.addMeasurementList(MeasurementList(float(g["times"][k]), g[p + "z"]))
        st = trk.lastScanStats
        assert [st["L"], st["G"], st["M"]] == g[p + "LGM"].tolist(), "scan %d L/G/M" % k
        assert np.array_equal(st["unused"], g[p + "unused"]), k
        leaf = trk.leafBatch()
        assert leaf["x"].shape[1] == 6 and leaf["P"].shape[1:] == (6, 6)
        leaf_cmp = dict(ID=leaf["ID"].astype(np.int64), meas=leaf["meas"].astype(np.int64), x=leaf["x"], cnllr=leaf["cnllr"], P=leaf["P"])
        ids_after = np.array([r.ID for r in trk.__targetList__])
        check_scan_against_fixture(g, k, ids_after, selected6(trk), trk.__clusterList__, len(leaf["ID"]), leaf_cmp, score_atol=SCORE_ATOL)
        assert sorted(i for i in ids_before if i not in ids_after.tolist()) == g[p + "dead"].tolist()
        assert trk.nOptimSolved == int(g[p + "n_ilp"])
        n_ilp += trk.nOptimSolved
    assert n_ilp > 20
    trk.close()


def test_six_state_build_refuses_the_four_state_only_entries(gold_dir):
    """The M-of-N initiator is the reference's 4-state one (m_of_n.py imports models/pv): a six-state Tracker must be created without
    it, and the 4-state tile seam of the six-state library says so instead of computing with the wrong layout."""
    from pymht_amd.tracker import Tracker
    from pymht_amd.models import ca
    with pytest.raises(NotImplementedError):
        Tracker(ca, 2.5, 1e-6, 1e-4, useInitiator=True)
    from pymht_amd import _lib
    from pymht_amd.device import Context
    ctx = Context(0, nx=6)
    assert ctx.lib is not _lib.load(nx=4) and ctx.lib.mht_abi_version() == _lib.load(nx=4).mht_abi_version()
    assert ctx.lib.mht_gate_scan(ctx.handle, None, None, None, 0, None, 0, None, None, None, None, None) == _lib.MHT_E_INVALID
    ctx.close()
