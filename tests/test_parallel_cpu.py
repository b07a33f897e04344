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
