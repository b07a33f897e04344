"""GPU: the OVERLAPPING / deferred path -- the code the headline number is measured on -- against traces recorded from the reference.

Every other trace test looks at the tracker after every scan, which flushes the deferred commit: the any-order grow launch, the per-target
publish records and the optimistic union-find never run there (FDyn::ovl = 0).  Here all scans of a fixture are streamed WITHOUT looking,
(i) through `Tracker.addMeasurementList` (drop-in API, device initiator) and (ii) through the raw `mht_forest_step` +
`mht_forest_add_targets_dev` replay that `bench.py` times (births replayed from an untimed pre-pass, like there), and only the LAST scan is
compared with what the reference produced (pymht/tracker.py:162-307): target ids, selections, clusters, the leaf set by sha-256 over all
leaf states / measurement numbers.  `uf_ovl` (host counters of the library) proves that the launches really overlapped."""
import ctypes as C
import os

import numpy as np
import pytest

from trace_util import sha

pytestmark = pytest.mark.gpu
SCORE_ATOL = 2e-5


def _uf_ovl(trk):
    v = np.zeros(2, dtype=np.int32)
    trk._lib.mht_forest_debug_read(trk._ctx.handle, b"uf_ovl", v.ctypes.data_as(C.c_void_p), 8)
    return int(v[0]), int(v[1])


def _make(g, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    trk = Tracker(pv, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]),
                  eta2=float(g["eta2"]), **kw)
    for x, ok in zip(g["x0"], g["accepted"]):
        n0 = trk.nTargets
        trk.initiateTarget(Target(float(g["t0"]), None, x.copy(), pv.P0, status="preinitialized"))
        assert (trk.nTargets > n0) == bool(ok)
    return trk


def _check_last(g, k, ids, sel_id, sel_meas, sel_x, sel_cnllr, clusters, leaf):
    p = "s%02d_" % k
    assert np.array_equal(ids, g[p + "ids"]), "target ids behind the last scan"
    assert np.array_equal(sel_id, g[p + "sel_ID"]) and np.array_equal(sel_meas, g[p + "sel_meas"]), "selections"
    assert np.array_equal(sel_x, g[p + "sel_x"]), "selected states (bit for bit)"
    assert np.allclose(sel_cnllr, g[p + "sel_cnllr"], rtol=0, atol=SCORE_ATOL)
    if clusters is not None:
        ptr, mem = g[p + "cl_ptr"], g[p + "cl_members"]
        assert len(clusters) == len(ptr) - 1
        for c, cl in enumerate(clusters):
            assert np.array_equal(np.asarray(cl), mem[ptr[c]:ptr[c + 1]]), "cluster %d" % c
    assert len(leaf["meas"]) == int(g[p + "leaf_n"][0])
    assert sha(np.asarray(leaf["meas"], dtype=np.int64)) == bytes(g[p + "leaf_sha_meas"]).hex(), "leaf measurement numbers (sha-256 of all leaves)"
    assert sha(np.asarray(leaf["x"], dtype=np.float64).reshape(-1, 4)) == bytes(g[p + "leaf_sha_x"]).hex(), "leaf states (sha-256 of all leaves)"


@pytest.mark.parametrize("name", ["g6b_trace_cfg3_long", "g6_trace_cfg3"])
def test_streamed_api_last_scan_equals_reference(name, gold_dir):
    """(i) the drop-in API, nobody looks between the scans: deferred commits, admissions riding in the next grow launch, reports folded two
    scans late, any-order grow launches next to the previous scan's ILP launch."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, name + ".npz"))
    trk = _make(g)
    n = int(g["n_scans"])
    for k in range(n):
        trk.addMeasurementList(MeasurementList(float(g["times"][k]), g["s%02d_z" % k]))
    uf, ovl = _uf_ovl(trk)      # (host counters: reading them does not touch the device)
    nodes = list(trk.getTrackNodes())
    leaf = trk.leafBatch()
    _check_last(g, n - 1, np.array([r.ID for r in trk.__targetList__]),
                np.array([t.ID for t in nodes], dtype=np.int64),
                np.array([0 if t.measurementNumber is None else t.measurementNumber for t in nodes], dtype=np.int64),
                np.array([np.asarray(t.x_0, dtype=np.float64) for t in nodes]).reshape(-1, 4),
                np.array([float(t.cumulativeNLLR) for t in nodes]), trk.__clusterList__, leaf)
    assert uf >= n - 2 and ovl > 0, "the scans were clustered by the union-find (%d) and grow launches overlapped the previous ILP launch (%d)" % (uf, ovl)
    trk.close()


@pytest.mark.parametrize("name,peek", [("g6b_trace_cfg3_long", 0), ("g6_trace_cfg3", 0), ("g6b_trace_cfg3_long", 5)])
def test_raw_replay_last_scan_equals_reference(name, peek, gold_dir):
    """(ii) what bench.py times: `mht_forest_step` on scans resident in HBM, the births of an untimed pre-pass through the API replayed with
    `mht_forest_add_targets_dev`, no report read until the end.
    peek: a report is read behind every peek-th scan -- the pending commit is flushed and the next step starts a fresh chain of
    overlapping launches."""
    import torch
    from pymht_amd import _lib
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, name + ".npz"))
    n = int(g["n_scans"])
    # pre-pass through the drop-in API: what the device initiator gave birth to, scan by scan
    pre = _make(g)
    births = [[] for _ in range(n)]
    orig = pre._apply_births

    def recording(b, scanTime, scanNumber, z_unused):
        orig(b, scanTime, scanNumber, z_unused)
        for r in b[b["id"] >= 0]:
            births[scanNumber - 1].append((r["x0"].astype(np.float32), r["P0"].reshape(4, 4).copy(), int(r["meas"])))
    pre._apply_births = recording
    for k in range(n):
        pre.addMeasurementList(MeasurementList(float(g["times"][k]), g["s%02d_z" % k]))
    pre.synchronize()
    assert [len(b) for b in births] == [len(g["s%02d_new_ids" % k]) for k in range(n)], "births of the pre-pass = the reference's"
    for k in range(n):      # ... and bit-identical to what the reference's initiator made (born_x / born_P of the fixture)
        if births[k]:
            assert np.array_equal(np.array([b[0] for b in births[k]], dtype=np.float64), g["s%02d_born_x" % k])
            assert np.array_equal(np.array([b[1] for b in births[k]], dtype=np.float64), g["s%02d_born_P" % k])
    pre.close()
    # raw replay
    trk = _make(g, useInitiator=False)
    lib, h, dev = trk._lib, trk._ctx.handle, trk._ctx.device
    zs = [np.ascontiguousarray(g["s%02d_z" % k], dtype=np.float32) for k in range(n)]
    zall = torch.from_numpy(np.concatenate(zs, axis=0)).to(dev)
    zoff = np.concatenate([[0], np.cumsum([len(z) for z in zs])]).astype(np.int64)
    flat = [b for per in births for b in per]
    boff = np.concatenate([[0], np.cumsum([len(per) for per in births])]).astype(np.int64)
    bx = torch.from_numpy(np.array([b[0] for b in flat], dtype=np.float64).reshape(-1, 4)).to(dev)
    bP = torch.from_numpy(np.array([b[1] for b in flat], dtype=np.float32).reshape(-1, 16)).to(dev)
    bf = torch.full((max(len(flat), 1),), 3, dtype=torch.uint8, device=dev)      # (float32 state and score chains: initiator-born)
    bm = torch.from_numpy(np.array([b[2] for b in flat], dtype=np.int32)).to(dev)
    bpd = torch.full((max(len(flat), 1),), float(g["P_d"]), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for k in range(n):
        _lib.check(lib.mht_forest_step(h, zall.data_ptr() + int(zoff[k]) * 8, len(zs[k])))
        nb = len(births[k])
        if nb:
            o = int(boff[k])
            _lib.check(lib.mht_forest_add_targets_dev(h, nb, bx.data_ptr() + o * 32, bP.data_ptr() + o * 64, bf.data_ptr() + o, bpd.data_ptr() + o * 8,
                                                      bm.data_ptr() + o * 4, 1, None, None))
        if peek and k % peek == peek - 1:
            rp = _lib.MhtScanReport()
            _lib.check(lib.mht_forest_report(h, C.byref(rp)))
            assert rp.error == 0 and rp.scan == k + 1
    uf, ovl = _uf_ovl(trk)
    rep = _lib.MhtScanReport()
    _lib.check(lib.mht_forest_report(h, C.byref(rep)))
    assert rep.error == 0
    dt = trk._REPORT_DTYPE
    recs = np.ctypeslib.as_array(C.cast(rep.targets, C.POINTER(C.c_uint8)), shape=(rep.n_targets * dt.itemsize,)).view(dt).copy()
    alive = recs[recs["status"] == 0]
    leaf = trk._leaf_export(trk._cfg.max_nodes)
    p = "s%02d_" % (n - 1)
    # (the report's rows are the targets the scan ran on: survivors in table order = the reference's list behind the scan, its last
    # scan gives birth to nothing)
    assert len(g[p + "new_ids"]) == 0
    _check_last(g, n - 1, alive["id"].astype(np.int64), alive["id"].astype(np.int64), alive["sel_meas"].astype(np.int64),
                np.array(alive["sel_x"], dtype=np.float64).reshape(-1, 4), alive["sel_cnllr"].astype(np.float64), None, leaf)
    assert uf >= n - 2 and (ovl > 0 or peek), (uf, ovl)
    trk.close()
