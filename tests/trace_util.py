"""Replay helpers for the golden scan traces recorded from the reference (tests/golden: the radar traces g2, g3, g3b, g6, g6b, g13*, g16,
the six-state g17, the AIS-aided g18-g18f) and for the oracle side of the live comparisons."""
import hashlib
import numpy as np

import mht_oracle as orc


class OracleInitiatorAdapter:
    """Lets OracleTracker drive the host-side M-of-N initiator (which is outside the hot path)."""

    def __init__(self, initiator, make_list):
        self.initiator, self.make_list = initiator, make_list

    def processMeasurements(self, time_, z, ais=()):
        return [(t.x_0, t.P_0, t.measurementNumber, t.measurement)
                for t in self.initiator.processMeasurements(self.make_list(time_, z), ais)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_oracle(g, with_initiator=True, model=None):
    from m_of_n_oracle import Initiator
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.models import pv
    init = None
    if with_initiator and model is None:
        init = OracleInitiatorAdapter(Initiator(2, 3, 20, pv.C_RADAR, pv.R_RADAR(), 4 * 2.5 ** 2), MeasurementList)
    o = orc.OracleTracker(float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]),
                          N=int(g["N"]), eta2=float(g["eta2"]), initiator=init, model=model)
    if model is not None:      # (a six-state model: pymht_amd.models.ca; roots carry the model's P0)
        for x, ok in zip(g["x0"], g["accepted"]):
            assert o.initiate_target(float(g["t0"]), x.copy(), model.P0.copy(), status="preinitialized") == bool(ok)
        return o
    custom = "P0s" in (g.files if hasattr(g, "files") else g)      # g16: roots with their own covariance, some with float32 states
    for i, (x, ok) in enumerate(zip(g["x0"], g["accepted"])):
        xr = x.astype(np.float32) if (custom and g["x0_f32"][i]) else x.copy()
        P0 = np.array(g["P0s"][i], dtype=np.float32) if custom else orc.model_P0()
        assert o.initiate_target(float(g["t0"]), xr, P0, status="preinitialized") == bool(ok)
    return o


def states_close(a, b, rel=1e-6):
    """The north star's state tolerance (1e-6 relative): relative to the largest component of each state vector -- a velocity near
    zero carries the rounding of the ~1e2..1e3 m positions it was differenced from.  Only used where a LIVE oracle on another host may
    have picked other BLAS kernels, and for the AIS-aided traces; the recorded radar traces -- the float32 chains of targets born from
    the device initiator included, since round 3 -- are compared with np.array_equal."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    nx = a.shape[-1] if a.ndim > 1 and a.shape[-1] in (4, 6) else 4
    a, b = a.reshape(-1, nx), b.reshape(-1, nx)
    if a.shape != b.shape:
        return False
    if a.size == 0:
        return True
    scale = np.maximum(np.abs(a).max(axis=1, keepdims=True), 1.0)
    return bool(np.all(np.abs(a - b) <= rel * scale))


def check_scan_against_fixture(g, k, ids, sel, clusters, n_leaves, leaf=None, score_atol=0.0):
    """Everything a fixture recorded from the reference holds for scan k.  States and covariances are compared BIT FOR BIT (both
    dtype chains, pre-initialised and initiator-born targets alike); `score_atol` only loosens the cumulative scores (the NLLR
    constant's float32 log is NumPy's SIMD polynomial, not a correctly rounded one: util.NLLR_ATOL)."""
    p = "s%02d_" % k
    assert np.array_equal(ids, g[p + "ids"]), "scan %d target ids" % k
    assert np.array_equal(sel["ID"], g[p + "sel_ID"]) and np.array_equal(sel["meas"], g[p + "sel_meas"]), "scan %d selection" % k
    assert np.array_equal(sel["x"], g[p + "sel_x"]), "scan %d selected states (bit for bit): %d of %d rows differ" % (
        k, int(np.any(np.asarray(sel["x"]) != g[p + "sel_x"], axis=1).sum()) if np.shape(sel["x"]) == g[p + "sel_x"].shape else -1, len(g[p + "sel_x"]))
    if score_atol == 0.0:
        assert np.array_equal(sel["cnllr"], g[p + "sel_cnllr"])
    else:
        assert np.allclose(sel["cnllr"], g[p + "sel_cnllr"], rtol=0, atol=score_atol)
    ptr, mem = g[p + "cl_ptr"], g[p + "cl_members"]
    assert len(clusters) == len(ptr) - 1
    for c, cl in enumerate(clusters):
        assert np.array_equal(np.asarray(cl), mem[ptr[c]:ptr[c + 1]]), "scan %d cluster %d" % (k, c)
    if p + "leaf_ID" in g:
        assert n_leaves == len(g[p + "leaf_ID"])
        if leaf is not None:
            assert np.array_equal(leaf["ID"], g[p + "leaf_ID"]) and np.array_equal(leaf["meas"], g[p + "leaf_meas"])
            assert np.array_equal(leaf["x"], g[p + "leaf_x"]), "scan %d leaf states (bit for bit): %d of %d rows differ" % (
                k, int(np.any(np.asarray(leaf["x"]) != g[p + "leaf_x"], axis=1).sum()), len(g[p + "leaf_x"]))
            assert np.array_equal(leaf["P"], g[p + "leaf_P"]), "scan %d leaf covariances (bit for bit)" % k
            if score_atol == 0.0:
                assert np.array_equal(leaf["cnllr"], g[p + "leaf_cnllr"])
            else:
                assert np.allclose(leaf["cnllr"], g[p + "leaf_cnllr"], rtol=0, atol=score_atol)
    else:
        assert n_leaves == int(g[p + "leaf_n"][0])
        if leaf is not None and p + "leaf_sha_x" in g:      # hashed traces: all leaf states / measurement numbers by checksum
            assert sha(np.asarray(leaf["meas"], dtype=np.int64)) == bytes(g[p + "leaf_sha_meas"]).hex(), "scan %d leaf measurement numbers (sha)" % k
            assert sha(np.asarray(leaf["x"], dtype=np.float64).reshape(-1, 4)) == bytes(g[p + "leaf_sha_x"]).hex(), "scan %d leaf states (sha-256 of all leaves)" % k


def replay_oracle(path):
    g = np.load(path)
    model = None
    if "nx" in g.files and int(g["nx"]) == 6:
        from pymht_amd.models import ca as model
    o = make_oracle(g, model=model)
    for k in range(int(g["n_scans"])):
        p = "s%02d_" % k
        info = o.add_scan(float(g["times"][k]), g[p + "z"], prune_similar=bool(g["prune_similar"]) if "prune_similar" in g else False)
        assert np.array_equal(info["unused"], g[p + "unused"])
        assert [info["L"], info["G"], info["M"]] == g[p + "LGM"].tolist()
        leaf = o.leaf_batch()
        check_scan_against_fixture(g, k, np.array([r.ID for r in o.targets]), o.selected(), o.clusters,
                                   len(leaf["ID"]), leaf, score_atol=2e-5)
        assert sorted(info["dead"]) == g[p + "dead"].tolist()
        assert list(info["new_ids"]) == g[p + "new_ids"].tolist()
    return o


# ---- AIS-aided traces (G18: Tracker.addMeasurementList(scan, aisList, aisInitialization=False), tracker.py:162-307 + :417-552) ----------
def ais_messages(g, k):
    """The AIS messages of scan k of a G18 fixture as oracle objects, in the order they were handed to the reference."""
    p = "s%02d_" % k
    return [orc.AisMessage(float(t), s.copy(), int(m), bool(h))
            for t, s, m, h in zip(g[p + "ais_time"], g[p + "ais_state"], g[p + "ais_mmsi"], g[p + "ais_high"])]


def make_oracle_ais(g):
    from m_of_n_oracle import Initiator
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.models import pv
    init = None
    if bool(g["with_initiator"]):
        init = OracleInitiatorAdapter(Initiator(2, 3, 20, pv.C_RADAR, pv.R_RADAR(), 4 * 2.5 ** 2), MeasurementList)
    o = orc.OracleTracker(float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]),
                          eta2=float(g["eta2"]), initiator=init, radarRange=float(g["radar_range"]), position=g["position"],
                          eta2_ais=float(g["eta2_ais"]))
    for x, ok in zip(g["x0"], g["accepted"]):
        assert o.initiate_target(float(g["t0"]), x.copy(), orc.model_P0(), status="preinitialized") == bool(ok)
    return o


def oracle_rows(nodes, nx=4):
    """Arrays of a list of oracle nodes, in the layout of the G18 fixtures (meas -1 = a pure AIS node, mmsi 0 = none)."""
    return dict(ID=np.array([n.ID for n in nodes], dtype=np.int64),
                x=np.array([np.asarray(n.x, dtype=np.float64) for n in nodes]).reshape(-1, nx),
                xf32=np.array([n.x.dtype == np.float32 for n in nodes], dtype=bool),
                P=np.array([np.asarray(n.P, dtype=np.float64) for n in nodes]).reshape(-1, nx, nx),
                Pf64=np.array([n.P.dtype == np.float64 for n in nodes], dtype=bool),
                cnllr=np.array([float(n.cnllr) for n in nodes], dtype=np.float64),
                meas=np.array([-1 if n.meas is None else n.meas for n in nodes], dtype=np.int64),
                mmsi=np.array([0 if n.mmsi is None else n.mmsi for n in nodes], dtype=np.int64))


def replay_oracle_ais(path):
    """The oracle alone on a G18 fixture: every array the reference produced, bit for bit (CPU suite)."""
    g = np.load(path)
    o = make_oracle_ais(g)
    for k in range(int(g["n_scans"])):
        p = "s%02d_" % k
        info = o.add_scan(float(g["times"][k]), g[p + "z"], ais=ais_messages(g, k), prune_similar=bool(g["prune_similar"]),
                          ais_initialization=bool(g["ais_init"]) if "ais_init" in g.files else False)
        leaves = oracle_rows([l for r in o.targets for l in r.leaves()])
        sel = oracle_rows(o.track_nodes)
        for key, v in leaves.items():
            assert np.array_equal(v, g[p + "leaf_" + key]), (k, "leaf", key)
        for key, v in sel.items():
            assert np.array_equal(v, g[p + "sel_" + key]), (k, "selected", key)
        assert np.array_equal([r.ID for r in o.targets], g[p + "ids"]) and sorted(info["dead"]) == list(g[p + "dead"])
        assert list(info["new_ids"]) == list(g[p + "new_ids"]) and np.array_equal(info["unused"], g[p + "unused"])
        assert list(info["used_mmsi"]) == list(g[p + "used_mmsi"])
        assert [info["L"], info["G"], info["M"], info["n_fused"]] == list(g[p + "LGM"])
        ptr, mem = g[p + "cl_ptr"], g[p + "cl_members"]
        assert len(o.clusters) == len(ptr) - 1 and all(np.array_equal(c, mem[ptr[i]:ptr[i + 1]]) for i, c in enumerate(o.clusters))
    return o
