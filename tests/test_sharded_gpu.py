"""GPU: ONE tracker whose independent clusters are solved on several devices (pymht_amd.parallel.ClusterShardedTracker,
mht_forest_step_sharded_*) against the same tracker on one device, scan by scan.
 * two shards inside one process on one GPU (two contexts, the all-reduce replaced by an element-wise maximum);
 * two processes over RCCL (needs two GPUs: skipped on a one-GPU box)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tracker(sc, device=0, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, maxTargets=1024, maxNodes=1 << 18,
                  maxMeasurements=1024, device=device, **kw)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    return trk


def _same(a, b, what):
    sa, sb = a._sel[0], b._sel[0]
    for name in ("id", "status", "sel_meas", "sel_x", "sel_cnllr", "score", "root_scan", "root_meas", "root_x", "n_leaves", "cluster"):
        assert np.array_equal(sa[name], sb[name]), (what, name)
    assert a.nTargets == b.nTargets, what
    for k in ("L", "G", "M", "leaves_out", "clusters"):
        assert a.lastScanStats[k] == b.lastScanStats[k], (what, k)


def _rd(trk, name, n):
    import ctypes as C
    from pymht_amd import _lib
    a = np.zeros(max(int(n), 1), dtype=np.int32)
    _lib.check(trk._lib.mht_forest_debug_read(trk._ctx.handle, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a


def _owner_table(trk):
    trk._ctx.synchronize()
    cnt = _rd(trk, "cl_counts", 8)
    nC, nM = int(cnt[0]), int(cnt[1])
    ml, own = _rd(trk, "multi_list", nM)[:nM], _rd(trk, "cl_owner", nC)
    return sorted((int(c), int(own[c])) for c in ml)


def _check_lpt(trk, owners, shards):
    from pymht_amd.parallel import assign_clusters
    if not owners:
        return
    cnt = _rd(trk, "cl_counts", 8)
    nC = int(cnt[0])
    ptr = _rd(trk, "cl_ptr", nC + 1)
    nT = int(ptr[nC])
    mem, tch, tce = _rd(trk, "cl_members", nT), _rd(trk, "tchild", nT + 1), _rd(trk, "tcend", nT + 1)
    cs = [c for c, _ in owners]      # ascending cluster index = the tie order of the device's ranking
    sizes = [int(sum(int(tce[m]) - int(tch[m]) for m in mem[ptr[c]:ptr[c + 1]])) for c in cs]
    want = assign_clusters(sizes, shards)
    assert [o for _, o in owners] == want.tolist(), "the device's cluster -> device table is not the LPT assignment"
    load = np.bincount(want, weights=sizes, minlength=shards)
    assert load.max() <= load.mean() + max(sizes), "LPT bound"


@pytest.mark.parametrize("name,n_scans,shards,similar", [("cfg3", 8, 2, False), ("dense", 10, 3, False), ("cfg2", 9, 2, True)])
def test_cluster_sharded_equals_single_device(name, n_scans, shards, similar):
    import torch
    from pymht_amd.parallel import ClusterShardedTracker
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    sc = make_config(name, seed=5446, n_scans=n_scans)
    solo = _tracker(sc)
    parts = [ClusterShardedTracker(_tracker(sc), shards, i, exchange=lambda t: None) for i in range(shards)]
    n_ilp = 0
    for k in range(n_scans):
        sl = MeasurementList(float(sc["times"][k]), sc["scans"][k])
        solo.addMeasurementList(sl, pruneSimilar=similar)      # (similar-state pruning is replicated on every shard)
        for p in parts:
            p.begin(sl, pruneSimilar=similar)
        # what the all-reduce(MAX) over RCCL does between the ranks
        merged = torch.stack([p.sel_rel for p in parts]).max(dim=0).values
        solved = torch.stack([(p.sel_rel >= 0).int() for p in parts]).sum(dim=0)
        assert int(solved.max()) <= 1, "a target was solved by two shards"
        # which device solved which cluster: the device's own table (cluster kernel: LPT on the clusters' column counts) must be the
        # longest-processing-time assignment of pymht_amd.parallel.assign_clusters, and the same on every shard
        owners = [_owner_table(p.trk) for p in parts]
        for o in owners[1:]:
            assert o == owners[0], "the shards disagree on who solves which cluster"
        _check_lpt(parts[0].trk, owners[0], shards)
        for p in parts:
            p.sel_rel.copy_(merged)
            p.end()
        for i, p in enumerate(parts):
            _same(p.trk, solo, "scan %d shard %d" % (k, i))
        n_ilp += solo.lastScanStats["ilp"]
    assert n_ilp > 0
    la = solo.leafBatch()
    for p in parts:
        lb = p.trk.leafBatch()
        for key in la:
            if key != "node":
                assert np.array_equal(la[key], lb[key]), key
        p.trk.close()
    solo.close()


def _rccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from pymht_amd.parallel import ClusterShardedTracker
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    sc = make_config("cfg3", seed=5446, n_scans=6)
    part = ClusterShardedTracker(_tracker(sc, device=rank), world, rank, dist=dist, always_exchange=True)      # (world 1: the all-reduce is issued all the same)
    solo = _tracker(sc, device=rank) if rank == 0 else None
    ok = True
    for k in range(6):
        sl = MeasurementList(float(sc["times"][k]), sc["scans"][k])
        part.addMeasurementList(sl)
        if solo is not None:
            solo.addMeasurementList(sl)
            try:
                _same(part.trk, solo, "scan %d" % k)
            except AssertionError:
                ok = False
    q.put((rank, ok, part.trk.nTargets))
    dist.barrier()
    dist.destroy_process_group()


def test_cluster_sharded_two_processes_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the one-process two-shard test covers the same code path)")
    import torch.multiprocessing as mp
    world, port = 2, 29733 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res) and res[0][2] == res[1][2]


@pytest.mark.gpu
def test_cluster_sharded_rccl_world_size_one():
    """The RCCL path on the hardware there is: one process, `torch.distributed` over the `nccl` backend (= RCCL) with world_size 1 --
    communicator creation and the device-tensor all_reduce of `merge_selections` run on the MI355X (the two-process test above needs
    a second GPU); the sharded tracker must equal a single forest scan by scan."""
    import torch.multiprocessing as mp
    port = 29533 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(0, 1, port, q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert res[0] == 0 and res[1] and res[2] > 0


def _gloo_worker(rank, world, port, q):
    """Two PROCESSES on the one GPU there is: each rank its own HIP context and forest on device 0, the exchange over a gloo group with
    the selections staged through host tensors (RCCL refuses two ranks on one device).  The real multi-process product path --
    rendezvous, identical LPT tables on both ranks, begin -> exchange -> end -- against a single forest, scan by scan."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pymht_amd.parallel import ClusterShardedTracker
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    n_scans = 6
    sc = make_config("cfg3", seed=5446, n_scans=n_scans)

    def exchange(t):      # device tensor -> host -> all-reduce(MAX) over gloo -> device
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        t.copy_(h)
        return t
    part = ClusterShardedTracker(_tracker(sc, device=0), world, rank, exchange=exchange)
    solo = _tracker(sc, device=0) if rank == 0 else None
    ok, n_mine, owners = True, 0, []
    for k in range(n_scans):
        sl = MeasurementList(float(sc["times"][k]), sc["scans"][k])
        part.begin(sl)
        n_mine += int((part.sel_rel >= 0).sum().item())      # targets whose cluster this rank solved
        owners.append(_owner_table(part.trk))
        part.exchange(part.sel_rel)
        part.end()
        if solo is not None:
            solo.addMeasurementList(sl)
            try:
                _same(part.trk, solo, "scan %d" % k)
            except AssertionError:
                ok = False
    # both ranks must have derived the same cluster -> rank table on their own devices
    tabs = [None] * world
    dist.all_gather_object(tabs, owners)
    ok = ok and all(t == tabs[0] for t in tabs)
    q.put((rank, ok, part.trk.nTargets, n_mine))
    dist.barrier()
    part.trk.close()
    if solo is not None:
        solo.close()
    dist.destroy_process_group()


def test_cluster_sharded_two_processes_one_gpu_gloo():
    """The multi-process path on the hardware there is (a one-GPU box): two ranks, both on GPU 0, `ClusterShardedTracker` with the
    exchange over gloo.  Every rank solved SOME of the clusters, none solved all, and the sharded tracker equals a single forest on
    every scan (pymht/tracker.py:228-236: the per-cluster ILPs are independent)."""
    import torch.multiprocessing as mp
    world, port = 2, 29833 + os.getpid() % 100
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert res[0][2] == res[1][2] > 0
    assert res[0][3] > 0 and res[1][3] > 0, "both ranks solved clusters: %r" % (res,)


@pytest.mark.parametrize("name", ["g18b_trace_ais_dense", "g18f_trace_ais_init_dense"])
def test_cluster_sharded_ais_trace_equals_reference(name):
    """AIS-aided scans through the cluster-sharded step (two shards on one GPU): the fused children (tracker.py:417-552) are made on every
    shard like grow and clustering, the ILPs -- with their AIS rows -- are spread by cluster.  Every shard must reproduce the trace recorded
    from the reference bit for bit (states and covariances of all leaves in the reference's dtypes), AIS-started tracks included."""
    import torch
    from pymht_amd.parallel import ClusterShardedTracker
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.ais import AisMessage, AisMessageList
    from pymht_amd.utils.classDefinitions import MeasurementList
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))

    def mk():
        trk = Tracker(pv, float(g["period"]), float(g["lambda_phi"]), float(g["lambda_nu"]), P_d=float(g["P_d"]), N=int(g["N"]), eta2=float(g["eta2"]),
                      eta2_ais=float(g["eta2_ais"]), radarRange=float(g["radar_range"]), position=g["position"], aisAided=True,
                      useInitiator=bool(g["with_initiator"]), maxTargets=256, maxNodes=1 << 16, maxMeasurements=256)
        for x in g["x0"]:
            trk.initiateTarget(Target(float(g["t0"]), None, x.copy(), pv.P0, status="preinitialized"))
        return trk
    parts = [ClusterShardedTracker(mk(), 2, i, exchange=lambda t: None) for i in range(2)]
    solved = [0, 0]
    try:
        for k in range(int(g["n_scans"])):
            p = "s%02d_" % k
            kw = dict(aisInitialization=bool(g["ais_init"]) if "ais_init" in g.files else False, pruneSimilar=bool(g["prune_similar"]))
            for q in parts:
                msgs = AisMessageList([AisMessage(float(t), s, int(m), bool(h)) for t, s, m, h in
                                       zip(g[p + "ais_time"], g[p + "ais_state"], g[p + "ais_mmsi"], g[p + "ais_high"])])
                q.begin(MeasurementList(float(g["times"][k]), g[p + "z"]), msgs, **kw)
            both = torch.stack([(q.sel_rel >= 0).int() for q in parts])
            assert int(both.sum(dim=0).max()) <= 1
            for i in range(2):
                solved[i] += int(both[i].sum())
            merged = torch.stack([q.sel_rel for q in parts]).max(dim=0).values
            for q in parts:
                q.sel_rel.copy_(merged)
                q.end()
            for q in parts:
                trk = q.trk
                assert np.array_equal([r.ID for r in trk.__targetList__], g[p + "ids"]), k
                nodes = list(trk.getTrackNodes())
                assert np.array_equal([n.ID for n in nodes], g[p + "sel_ID"]), k
                assert np.array_equal([0 if n.mmsi is None else n.mmsi for n in nodes], g[p + "sel_mmsi"]), k
                lb = trk.leafBatch()
                assert np.array_equal(lb["ID"], g[p + "leaf_ID"]) and np.array_equal(lb["meas"], g[p + "leaf_meas"]) and np.array_equal(lb["mmsi"], g[p + "leaf_mmsi"]), k
                assert np.array_equal(lb["x"], g[p + "leaf_x"]) and np.array_equal(lb["Pf64"], g[p + "leaf_Pf64"]) and np.array_equal(lb["P"], g[p + "leaf_P"]), k
        assert min(solved) > 0, solved
    finally:
        for q in parts:
            q.trk.close()


@pytest.mark.parametrize("shards", [2, 3])
def test_giant_component_is_searched_by_every_shard(shards):
    """A gating graph that is one big component (tracker.py:1155-1217: one CBC call): the dense scenario of
    test_cluster_blp_gpu.py::test_blp_team_search_equals_single_workgroup_search, whose 30-60-target clusters branch.  With the exchange
    block of mht_forest_step_sharded_begin2 every shard searches its share of the subtrees and files its best selection; the merged
    block lets the smallest value win on every shard -- the same decisions as one device, scan by scan, and on the scans with a team
    cluster every shard has filed (nobody waited for an owner)."""
    import torch
    from pymht_amd.parallel import ClusterShardedTracker
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_scenario
    sc = make_scenario(T=66, radius=201.0, lambda_phi=1.5e-4, n_scans=7, P_d=0.73, period=2.5, seed=5494)

    def mk():
        trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=9.21, useInitiator=False, maxTargets=512, maxNodes=1 << 18)
        trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
        return trk

    solo = mk()
    parts = [ClusterShardedTracker(mk(), shards, i, exchange=lambda t: None) for i in range(shards)]
    T, XT = 512, 260
    filed_scans = branched = 0
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        sl = MeasurementList(float(t), z)
        solo.addMeasurementList(sl)
        for p in parts:
            p.begin(sl)
        blocks = torch.stack([p.sel_rel for p in parts])
        assert int((blocks >= 0).int().sum(dim=0).max()) <= 1, "two shards wrote the same word of the exchange block"
        files = blocks[:, T:].reshape(shards, shards, 8, XT).cpu().numpy()      # [written by shard][slot of shard][team slot][words]
        n_team = int(_rd(parts[0].trk, "cl_counts", 8)[5])
        for i in range(shards):
            assert (files[i, [j for j in range(shards) if j != i]] == -1).all(), "a shard wrote into another shard's slots"
            for q in range(n_team):
                assert files[i, i, q, 3] >= 24, "shard %d did not file for team cluster %d in scan %d" % (i, q, k)
        if n_team:
            filed_scans += 1
            # the members' entries of the selection part are empty: nobody's selection may leak past the vote
            ptr, lst = _rd(parts[0].trk, "cl_ptr", 513), _rd(parts[0].trk, "team_list", 8)
            mem = _rd(parts[0].trk, "cl_members", 512)
            for q in range(n_team):
                c = int(lst[q])
                assert (blocks[:, mem[ptr[c]:ptr[c + 1]]] == -1).all()
        merged = blocks.max(dim=0).values
        from pymht_amd.parallel import team_winners
        win = team_winners(merged.cpu().numpy(), T, shards)      # (the host's statement of the vote)
        assert sorted(win) == list(range(n_team))
        for p in parts:
            p.sel_rel.copy_(merged)
            p.end()
        if n_team:
            for p in parts:      # the device's vote (shard_team_resolve_kernel) wrote the winner's ordinals into the members' entries
                got = p.sel_rel.cpu().numpy()
                for q in range(n_team):
                    c = int(lst[q])
                    assert got[mem[ptr[c]:ptr[c + 1]]].tolist() == win[q][1], (k, q)
        for i, p in enumerate(parts):
            _same(p.trk, solo, "scan %d shard %d" % (k, i))
        branched += solo.lastScanStats["branched"]
    assert filed_scans >= 2 and branched > 0
    for p in parts:
        p.trk.close()
    solo.close()
