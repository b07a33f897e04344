"""CPU: the N>1 layout of the scan path with world_size-2 gloo: disjoint sectors, clock reduction, track gathering."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pymht_amd import parallel
    from pymht_amd.utils.scenario import make_config
    import mht_oracle as orc
    # every rank tracks its own sector with the CPU oracle (the GPU forest is exercised by the -m gpu tests)
    sc = make_config("cfg1", seed=parallel.sector_seed(172362, rank), centre=parallel.sector_centre(rank), n_scans=6)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99)
    for x in sc["x0"]:
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0())
    for z, t in zip(sc["scans"], sc["times"]):
        o.add_scan(float(t), z)
    sel = o.selected()
    elapsed, ok = parallel.reduce_clock(0.5 + rank, True, dist)
    tracks = parallel.gather_tracks(sel["ID"], sel["x"], dist)
    q.put((rank, elapsed, ok, [(i.tolist(), s.tolist()) for i, s in tracks], sel["x"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_sectors_over_gloo():
    world, port = 2, 29533 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(abs(r[1] - 1.5) < 1e-12 and r[2] for r in res)            # max over ranks, AND of the flags
    assert res[0][3] == res[1][3]                                           # every rank sees the same gathered picture
    for r in range(world):
        ids, xs = res[0][3][r]
        assert np.allclose(np.array(xs), np.array(res[r][4]))              # ...which holds rank r's own tracks at slot r
        assert np.all(np.abs(np.array(xs)[:, 0] - 20000.0 * r) < 5000.0)   # sectors are disjoint (20 km apart)


def test_cluster_assignment_balances():
    from pymht_amd.parallel import assign_clusters
    sizes = [900, 10, 10, 400, 395, 5, 300, 295]
    ranks = assign_clusters(sizes, 4)
    load = np.bincount(ranks, weights=sizes, minlength=4)
    assert set(ranks.tolist()) == {0, 1, 2, 3} and load.max() == 900 and load.min() >= 400


def _merge_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pymht_amd import parallel
    # the exchange step of a cluster-sharded scan (ClusterShardedTracker): rank r solved the clusters c with c % world == r; every
    # target carries its selection on exactly one rank, -1 elsewhere
    T, labels = 40, np.arange(40) // 3
    sel = np.where(labels % world == rank, 100 + np.arange(T), -1).astype(np.int32)
    out = parallel.merge_selections(torch.from_numpy(sel), dist)
    q.put((rank, out.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_selection_exchange_over_gloo():
    world, port = 2, 29433 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == (100 + np.arange(40)).tolist()      # every rank ends up with every selection


def _team_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pymht_amd import parallel
    # the exchange block of mht_forest_step_sharded_begin2: T selections, then every rank's files for the clusters searched by the teams
    # of ALL ranks.  Rank r solved the ordinary clusters c with c % world == r and files ITS best selection of two giant clusters
    # (targets 30..59 in slot 0, 70..99 in slot 1) with the value it reached; the ranks' values differ, one of them is negative,
    # two are one ulp apart
    T = 128
    block = np.full(T + world * parallel.TEAM_MAX * parallel.XT_WORDS, -1, dtype=np.int32)
    labels = np.arange(T) // 3
    giant = ((np.arange(T) >= 30) & (np.arange(T) < 60)) | ((np.arange(T) >= 70) & (np.arange(T) < 100))
    block[:T] = np.where((labels % world == rank) & ~giant, 100 + np.arange(T), -1)
    v0 = [412.5, np.nextafter(412.5, 0.0)][rank]          # slot 0: rank 1 is one ulp better
    v1 = [-3.25, -3.0][rank]                              # slot 1: rank 0 is better (negative values order the other way round in the raw bits)
    parallel.write_team_file(block, T, rank, 0, v0, 1000 * (rank + 1) + np.arange(30))
    parallel.write_team_file(block, T, rank, 1, v1, 2000 * (rank + 1) + np.arange(30))
    assert (block[T:][block[T:] != -1] >= 0).all()      # every filed word is non-negative: MAX against -1 is a gather
    out = parallel.merge_selections(torch.from_numpy(block), dist).numpy()
    win = parallel.team_winners(out, T, world)
    q.put((rank, out[:T].tolist(), {k: (v[0], v[1][:3]) for k, v in win.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_team_files_are_gathered_by_the_same_all_reduce():
    """The giant-component exchange (pymht_amd.parallel.write_team_file / team_winners = csrc/mht_blp.hip xteam_out /
    shard_team_resolve_kernel): ONE all-reduce(MAX) merges the selections and gathers every rank's file; every rank then takes the same
    vote -- smallest value, also across the sign and at one ulp."""
    world, port = 2, 29633 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_team_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    sel, win = res[0][1], res[0][2]
    giant = [t for t in range(128) if 30 <= t < 60 or 70 <= t < 100]
    assert all(sel[t] == -1 for t in giant) and all(sel[t] == 100 + t for t in range(128) if t not in giant)
    assert win == {0: (1, [2000, 2001, 2002]), 1: (0, [2000, 2001, 2002])}
