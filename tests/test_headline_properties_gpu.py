"""GPU, BASELINE headline size (500 targets, ~500 measurements/scan, N-scan 5, ~13 k leaves): properties that do not need a
recorded trace -- the gate is sound and complete when re-derived from the leaf batch, the global hypothesis never uses a
measurement twice, leaf ranges stay in target order, and the run is reproducible bit for bit."""
import numpy as np
import pytest

import mht_oracle as orc

pytestmark = pytest.mark.gpu


def _run(n_scans, check_gate_at=()):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.scenario import make_config
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = make_config("cfg3", seed=5446, n_scans=n_scans)
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, maxTargets=2048, maxNodes=1 << 17,
                  maxMeasurements=1024)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    A, Q, Cm, R = orc.model_Phi(sc["period"]), orc.model_Q(sc["period"]), orc.model_C(), orc.model_R()
    digest = []
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        before = trk.leafBatch() if k in check_gate_at else None
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        if before is not None:
            # re-derive the gate of this scan from the leaves it started with (NumPy restatement of kalman.py, bulk call)
            f32 = (before["flags"] & 1).astype(bool)
            idx = [None] * len(f32)
            for mask, dt in ((~f32, np.float64), (f32, np.float32)):
                if mask.any():
                    o = orc.process_leaves(A, Q, Cm, R, 5.99, sc["lambda_phi"] + 1e-4, before["x"][mask].astype(dt), before["P"][mask],
                                           [sc["P_d"]] * int(mask.sum()), z)
                    for i, g in zip(np.where(mask)[0], o["idx"]):
                        idx[i] = g
            G = sum(len(g) for g in idx)
            assert (st["L"], st["G"]) == (len(f32), G), k
            used = np.zeros(len(z), bool)
            for g in idx:
                used[g] = True
            assert np.array_equal(st["unused"][:len(z)], ~used), k
        sel = trk._sel[0]
        hits = sel["sel_meas"][sel["sel_meas"] > 0]
        assert len(np.unique(hits)) == len(hits), "scan %d: a measurement is used by two selected leaves" % k
        lb = trk.leafBatch()
        assert np.all(np.diff(lb["target"]) >= 0), k                      # leaves grouped by target, in target-list order
        assert len(lb["ID"]) - st["leaves_out"] == trk.nTargets - len(sel), k     # + one root leaf per track born in this scan
        # (node indices are handles that depend on the order in which the targets' workgroups took their blocks: not part of the digest)
        digest.append((st["L"], st["G"], st["ilp"], int(sel["sel_meas"].sum()), float(sel["sel_cnllr"].sum()), float(sel["sel_x"].sum())))
    trk.close()
    return digest


def test_headline_gate_selection_properties_and_reproducibility():
    a = _run(12, check_gate_at=(3, 9))
    b = _run(12)
    assert a == b                                                          # results bit-reproducible (no atomics-order dependence)
    assert a[-1][0] > 10000 and a[-1][2] > 10                             # it really is the headline regime
