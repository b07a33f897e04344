"""Multi-GPU layout of the scan path: one process per GPU, one independent sensor sector (or cluster group) per
process.  The path shards without any data-path collective when the gating graph partitions (BASELINE config 4:
disjoint sectors), so torch.distributed (RCCL over xGMI on the GPU box, gloo in the CPU tests) is used only for
 - the barrier / max-over-ranks clock of the benchmark, and
 - gathering the per-sector track lists into one picture after a scan (`gather_tracks`), KB-sized and latency bound.
SURVEY.md 8(e)."""
import numpy as np
import torch


def sector_centre(rank, spacing=20000.0):
    """Disjoint sectors 20 km apart."""
    return (spacing * rank, 0.0)


def sector_seed(base_seed, rank):
    return int(base_seed) + 1000 * int(rank)


def assign_clusters(cluster_sizes, world_size):
    """Longest-processing-time assignment of independent clusters (or sectors) to ranks: returns rank per cluster.
    Used when one scan's gating graph partitions into many clusters that are to be solved on different GPUs."""
    order = np.argsort(-np.asarray(cluster_sizes, dtype=np.int64), kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    out = np.zeros(len(cluster_sizes), dtype=np.int64)
    for c in order:
        r = int(np.argmin(load))
        out[c] = r
        load[r] += int(cluster_sizes[c])
    return out


def reduce_clock(elapsed_s, ok, dist=None, device="cpu"):
    """max over ranks of the elapsed time, logical AND of the per-rank consistency flags."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), bool(ok)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return float(t.item()), bool(f.item())


def gather_tracks(ids, states, dist=None, device="cpu", max_tracks=4096):
    """All-gather the per-rank track lists (ids (n,), states (n,4)) -> list over ranks of (ids, states)."""
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    states = np.asarray(states, dtype=np.float64).reshape(-1, 4)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [(ids, states)]
    n = len(ids)
    assert n <= max_tracks
    buf = torch.zeros((max_tracks, 5), dtype=torch.float64, device=device)
    buf[:n, 0] = torch.from_numpy(ids.astype(np.float64)).to(device)
    buf[:n, 1:] = torch.from_numpy(states).to(device)
    cnt = torch.tensor([n], dtype=torch.int64, device=device)
    world = dist.get_world_size()
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(bufs, buf)
    dist.all_gather(cnts, cnt)
    out = []
    for b, c in zip(bufs, cnts):
        k = int(c.item())
        a = b[:k].cpu().numpy()
        out.append((a[:, 0].astype(np.int64), a[:, 1:].copy()))
    return out
