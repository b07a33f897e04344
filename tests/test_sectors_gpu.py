"""GPU: several independent sectors stepped as ONE batched launch set (mht_group_step, BASELINE config 4 on one device) must give,
sector by sector, exactly what the same trackers give when each is stepped on its own."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tracker(sc, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, maxTargets=1024,
                  maxNodes=1 << 18, maxMeasurements=1024, deviceTiming=False, **kw)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    return trk


def _sectors(n, n_scans, name="cfg3"):
    from pymht_amd.utils.scenario import make_config
    return [make_config(name, seed=5446 + 17 * q, n_scans=n_scans, centre=(0.0, 20000.0 * q)) for q in range(n)]


def _same_state(a, b, what):
    sa, sb = a._sel[0], b._sel[0]
    for name in ("id", "status", "sel_meas", "sel_x", "sel_cnllr", "score", "root_scan", "root_meas", "root_x", "n_leaves", "cluster"):
        assert np.array_equal(sa[name], sb[name]), (what, name)
    assert a.nTargets == b.nTargets, what
    for k in ("L", "G", "M", "leaves_out", "clusters", "ilp"):
        assert a.lastScanStats[k] == b.lastScanStats[k], (what, k)


@pytest.mark.parametrize("n_sec,n_scans", [(4, 9), (16, 7)])
def test_headline_sectors_batched_equal_single_trackers(n_sec, n_scans):
    """cfg4's workload on one device: n x (500 targets, ~500 measurements/scan, N-scan 5) through the drop-in API, with the
    initiator giving birth to tracks (so deferred and immediate commits both occur inside the batch).  16 sectors in one group take the
    wavefront-per-target grow kernel and the light ILP pass (wavefront per cluster, round-0 certificate) in front of the full solver:
    every sector must come out as its own single tracker does."""
    from pymht_amd.sectors import SectorGroup
    from pymht_amd.utils.classDefinitions import MeasurementList
    scs = _sectors(n_sec, n_scans)
    solo = [_tracker(sc) for sc in scs]
    grp_t = [_tracker(sc) for sc in scs]
    grp = SectorGroup(grp_t)
    for k in range(n_scans):
        lists = [MeasurementList(float(sc["times"][k]), sc["scans"][k]) for sc in scs]
        for t, sl in zip(solo, lists):
            t.addMeasurementList(sl)
        grp.addMeasurementLists(lists)
        for q in range(n_sec):
            _same_state(grp_t[q], solo[q], "scan %d sector %d" % (k, q))
    assert solo[0].lastScanStats["L"] > 10000 and solo[0].lastScanStats["ilp"] > 10      # the headline regime
    for q in range(n_sec):
        la, lb = grp_t[q].leafBatch(), solo[q].leafBatch()
        for key in la:
            if key != "node":      # node indices are handles (block taken with an atomic)
                assert np.array_equal(la[key], lb[key]), (q, key)
    grp.close()
    for t in solo + grp_t:
        t.close()


@pytest.mark.parametrize("light", ["0", "1"])
def test_group_raw_replay_uneven_sectors(light, monkeypatch):
    """Raw C-ABI replay (nothing fetched between the scans: every commit rides in the next batched grow launch): sectors of
    different sizes, one of them with empty scans in between, compared with single-forest replays at the end."""
    import torch
    from pymht_amd import _lib
    from pymht_amd.sectors import SectorGroup
    from pymht_amd.tracker import _REPORT_DTYPE
    from pymht_amd.utils.scenario import make_scenario
    monkeypatch.setenv("MHT_BLP_LIGHT", light)      # (the light ILP pass of a group, forced on / off whatever the group's size)
    params = [dict(T=60, radius=500.0, lambda_phi=3e-5), dict(T=7, radius=300.0, lambda_phi=1e-5), dict(T=200, radius=2500.0, lambda_phi=2e-6)]
    scs = []
    for q, pr in enumerate(params):
        sc = make_scenario(n_scans=10, P_d=0.85, seed=100 + q, centre=(0.0, 20000.0 * q), **pr)
        sc["N"] = 3
        scs.append(sc)
    scs[1]["scans"][4] = np.zeros((0, 2), np.float32)      # a sector that sees nothing for a scan
    solo = [_tracker(sc, useInitiator=False) for sc in scs]
    grp_t = [_tracker(sc, useInitiator=False) for sc in scs]
    grp = SectorGroup(grp_t)
    dev = grp_t[0]._ctx.device
    zd = [[torch.from_numpy(np.ascontiguousarray(z, np.float32).reshape(-1, 2) if len(z) else np.zeros((1, 2), np.float32)).to(dev)
           for z in sc["scans"]] for sc in scs]
    for k in range(10):
        for q, t in enumerate(solo):
            _lib.check(t._lib.mht_forest_step(t._ctx.handle, zd[q][k].data_ptr(), len(scs[q]["scans"][k])))
        grp.step_dev([zd[q][k].data_ptr() for q in range(3)], [len(scs[q]["scans"][k]) for q in range(3)])

    def report(trk):
        rep = _lib.MhtScanReport()
        _lib.check(trk._lib.mht_forest_report(trk._ctx.handle, C.byref(rep)))
        recs = np.ctypeslib.as_array(C.cast(rep.targets, C.POINTER(C.c_uint8)), shape=(rep.n_targets * _REPORT_DTYPE.itemsize,)) \
            .view(_REPORT_DTYPE).copy()
        return (rep.scan, rep.n_targets, rep.n_alive, rep.n_leaves_in, rep.n_children, rep.n_leaves_out, rep.n_clusters, rep.n_ilp), recs

    for q in range(3):
        ha, ra = report(grp_t[q])
        hb, rb = report(solo[q])
        assert ha == hb and ha[0] == 10, (q, ha, hb)
        for name in _REPORT_DTYPE.names:
            if name not in ("sel_node", "root_node"):
                assert np.array_equal(ra[name], rb[name]), (q, name)
        la, lb = grp_t[q].leafBatch(), solo[q].leafBatch()
        for key in la:
            if key != "node":
                assert np.array_equal(la[key], lb[key]), (q, key)
    grp.close()
    for t in solo + grp_t:
        t.close()


def test_value_table_generations_single_and_batched(monkeypatch):
    """The covariance-value table never recycles an id; three quarters full, the live leaves are re-keyed into the other generation
    (mht_forest.hip: vt_switch_generation).  With a table of 2 048 ids a 50-target stream switches every few scans: selections, scores,
    states and the leaves' covariances must be what a table that never fills gives -- stepped alone and as members of a group (whose
    cached argument blocks are rewritten at every switch)."""
    from pymht_amd import _lib
    from pymht_amd.sectors import SectorGroup
    from pymht_amd.utils.classDefinitions import MeasurementList
    n_scans = 60
    scs = _sectors(2, n_scans, name="cfg2")
    ref = [_tracker(sc) for sc in scs]                  # default capacity: no switch
    monkeypatch.setenv("MHT_VTAB_CAP", "2048")
    solo = [_tracker(sc) for sc in scs]
    grp_t = [_tracker(sc) for sc in scs]
    monkeypatch.delenv("MHT_VTAB_CAP")
    grp = SectorGroup(grp_t)
    for k in range(n_scans):
        lists = [MeasurementList(float(sc["times"][k]), sc["scans"][k]) for sc in scs]
        for t, sl in zip(ref + solo, lists + lists):
            t.addMeasurementList(sl)
        grp.addMeasurementLists(lists)
        for q in range(2):
            _same_state(solo[q], ref[q], "scan %d sector %d (alone)" % (k, q))
            _same_state(grp_t[q], ref[q], "scan %d sector %d (group)" % (k, q))
    for q in range(2):
        la, lb, lc = ref[q].leafBatch(), solo[q].leafBatch(), grp_t[q].leafBatch()
        for key in ("ID", "meas", "x", "cnllr", "P"):
            assert np.array_equal(la[key], lb[key]) and np.array_equal(la[key], lc[key]), (q, key)
    for t, least in zip(ref + solo + grp_t, [0, 0, 3, 3, 3, 3]):
        r = np.zeros(1, np.int32)
        _lib.check(t._lib.mht_forest_debug_read(t._ctx.handle, b"vt_rebuilds", r.ctypes.data_as(C.c_void_p), 4))
        assert (r[0] == 0) if least == 0 else (r[0] >= least), (least, int(r[0]))
    grp.close()
    for t in ref + solo + grp_t:
        t.close()


def test_value_table_generations_with_similar_state_pruning(monkeypatch):
    """Similar-state pruning gives merged hypotheses keys of their own (mht_similar.hip); the switch of the value table's generation
    re-keys them like every other live leaf, and skips nothing because of the emptied slots in the leaf ranges."""
    from pymht_amd import _lib
    from pymht_amd.utils.classDefinitions import MeasurementList
    n_scans = 40
    sc = _sectors(1, n_scans, name="cfg2")[0]
    ref = _tracker(sc)
    monkeypatch.setenv("MHT_VTAB_CAP", "2048")
    small = _tracker(sc)
    monkeypatch.delenv("MHT_VTAB_CAP")
    for k in range(n_scans):
        sl = MeasurementList(float(sc["times"][k]), sc["scans"][k])
        on = k % 7 != 5
        ref.addMeasurementList(sl, pruneSimilar=on)
        small.addMeasurementList(sl, pruneSimilar=on)
        _same_state(small, ref, "scan %d" % k)
    la, lb = ref.leafBatch(), small.leafBatch()
    for key in ("ID", "meas", "x", "cnllr", "P"):
        assert np.array_equal(la[key], lb[key]), key
    r = np.zeros(1, np.int32)
    _lib.check(small._lib.mht_forest_debug_read(small._ctx.handle, b"vt_rebuilds", r.ctypes.data_as(C.c_void_p), 4))
    assert r[0] >= 2, int(r[0])
    ref.close()
    small.close()


@pytest.mark.parametrize("light", ["0", "1"])
def test_grouped_sectors_with_similar_state_pruning(light, monkeypatch):
    """Similar-state pruning (tracker.py:230-231) inside a group: members 0 and 2 prune on most scans, members 1 and 3 never; every
    member must end up exactly where a single tracker with the same switch sequence does."""
    from pymht_amd.sectors import SectorGroup
    from pymht_amd.utils.classDefinitions import MeasurementList
    monkeypatch.setenv("MHT_BLP_LIGHT", light)
    n_scans, S = 14, 4
    scs = _sectors(S, n_scans, name="cfg2")
    solo = [_tracker(sc) for sc in scs]
    grouped = [_tracker(sc) for sc in scs]
    grp = SectorGroup(grouped)
    for k in range(n_scans):
        flags = [(q % 2 == 0) and (k % 4 != 1) for q in range(S)]
        lists = [MeasurementList(float(sc["times"][k]), sc["scans"][k]) for sc in scs]
        for q in range(S):
            solo[q].addMeasurementList(lists[q], pruneSimilar=flags[q])
        grp.addMeasurementLists(lists, pruneSimilar=flags)
        for q in range(S):
            _same_state(grouped[q], solo[q], "scan %d sector %d" % (k, q))
    for q in range(S):
        la, lb = solo[q].leafBatch(), grouped[q].leafBatch()
        for key in ("ID", "meas", "x", "cnllr", "P"):
            assert np.array_equal(la[key], lb[key]), (q, key)
    grp.close()
    for t in solo + grouped:
        t.close()


def test_group_with_an_ais_aided_member(gold_dir):
    """A group whose sectors are not all radar-only: two cfg2 sectors share the batched launches, the third is an AIS-aided tracker fed
    the reference's recorded G18b trace (tracker.py:417-552) -- stepped with launches of its own behind the group's.  The radar sectors
    must equal single trackers, the AIS sector its fixture (leaf identities, states and covariances bit for bit)."""
    import os
    from pymht_amd.sectors import SectorGroup
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.ais import AisMessage, AisMessageList
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(gold_dir, "g18b_trace_ais_dense.npz"))
    n_scans = int(g["n_scans"])
    scs = _sectors(2, n_scans, name="cfg2")
    solo = [_tracker(sc) for sc in scs]
    grouped = [_tracker(sc) for sc in scs]
    ais_trk = Tracker(pv, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]), eta2=float(g["eta2"]),
                      eta2_ais=float(g["eta2_ais"]), radarRange=float(g["radar_range"]), position=g["position"], aisAided=True,
                      useInitiator=bool(g["with_initiator"]), maxTargets=256, maxNodes=1 << 16, maxMeasurements=256)
    for x in g["x0"]:
        ais_trk.initiateTarget(Target(float(g["t0"]), None, x.copy(), pv.P0, status="preinitialized"))
    grp = SectorGroup([grouped[0], ais_trk, grouped[1]])
    try:
        with pytest.raises(NotImplementedError):      # (messages for a sector that was not made for them)
            grp.addMeasurementLists([MeasurementList(0.0, np.zeros((0, 2), np.float32))] * 3,
                                    aisLists=[AisMessageList([AisMessage(0.0, np.zeros(4), 1, True)]), None, None])
        for k in range(n_scans):
            p = "s%02d_" % k
            msgs = AisMessageList([AisMessage(float(t), s, int(m), bool(h)) for t, s, m, h in
                                   zip(g[p + "ais_time"], g[p + "ais_state"], g[p + "ais_mmsi"], g[p + "ais_high"])])
            lists = [MeasurementList(float(sc["times"][k]), sc["scans"][k]) for sc in scs]
            for q in range(2):
                solo[q].addMeasurementList(lists[q])
            grp.addMeasurementLists([lists[0], MeasurementList(float(g["times"][k]), g[p + "z"]), lists[1]], aisLists=[None, msgs, None],
                                    pruneSimilar=[False, bool(g["prune_similar"]), False],
                                    aisInitialization=bool(g["ais_init"]) if "ais_init" in g.files else False)
            for q in range(2):
                _same_state(grouped[q], solo[q], "scan %d sector %d" % (k, q))
            lb = ais_trk.leafBatch()
            assert np.array_equal(lb["ID"], g[p + "leaf_ID"]) and np.array_equal(lb["meas"], g[p + "leaf_meas"]) and np.array_equal(lb["mmsi"], g[p + "leaf_mmsi"]), k
            assert np.array_equal(lb["x"], g[p + "leaf_x"]) and np.array_equal(lb["P"], g[p + "leaf_P"]), k
    finally:
        grp.close()
        for t in solo + grouped + [ais_trk]:
            t.close()
