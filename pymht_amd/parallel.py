"""Multi-GPU layout of the scan path: one process per GPU.

Two levels, as the north star allows them:
 * independent sensor sectors (BASELINE config 4): one tracker per rank, no data-path collective; torch.distributed (RCCL over
   xGMI on the GPU box, gloo in the CPU tests) only carries the barrier / max-over-ranks clock of the benchmark and the gathering
   of the per-sector track lists into one picture (`gather_tracks`, KB-sized, latency bound);
 * ONE tracker on several GPUs when its gating graph partitions (`ClusterShardedTracker`): every rank holds the same forest and is
   fed the same scans; the independent per-cluster ILPs (tracker.py:228-236) are spread over the ranks (by size: longest-processing-time
   first on the clusters' column counts, computed identically on every device -- `assign_clusters` below states the rule) and
   the selections travel in ONE all-reduce(MAX) of max_targets int32 per scan (child ordinals inside each target's block, -1 = not
   mine).  A graph that is one component is solved by one rank while the others wait: the one-GPU fallback."""
import numpy as np
import torch


def sector_centre(rank, spacing=20000.0):
    """Disjoint sectors 20 km apart."""
    return (spacing * rank, 0.0)


def sector_seed(base_seed, rank):
    return int(base_seed) + 1000 * int(rank)


def assign_clusters(cluster_sizes, world_size):
    """Longest-processing-time assignment of independent clusters to ranks: returns rank per cluster (largest first, each to the least
    loaded rank; ties: lower cluster index, lower rank).  The cluster-sharded step computes exactly this table ON THE DEVICE, on every
    rank identically, from the clusters' column counts (csrc/mht_cluster.hip: `cl_owner`); this host statement of the rule is what the
    tests compare the device's table with (tests/test_sharded_gpu.py, tests/test_parallel_cpu.py)."""
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


def merge_selections(sel_rel, dist=None, always=False):
    """The exchange step of a cluster-sharded scan: element-wise MAX over the ranks of the per-target selections (-1 = not solved
    here).  `sel_rel`: int32 tensor [max_targets] on the rank's device (cpu tensors with gloo).  In place.  always=True: the collective
    is issued with one rank too (tests / `bench.py` with MHT_BENCH_FORCE_DIST: the RCCL call itself on a one-GPU box)."""
    if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or always):
        dist.all_reduce(sel_rel, op=dist.ReduceOp.MAX)
    return sel_rel


# ---- the files of a giant component searched by the teams of all ranks (mht_forest_step_sharded_begin2; csrc/mht_blp.hip: xteam_out,
# shard_team_resolve_kernel) -- the host-side statement of the block's layout and of the vote, for tests and tooling ----------------------
TEAM_MAX, TEAM_SEL = 8, 256
XT_WORDS = 4 + TEAM_SEL


def value_key(v):
    """The order-preserving 64-bit key of a float64 objective (csrc/mht_blp.hip: enum_key) as the three non-negative chunks a file carries
    (22 + 21 + 21 bits): every word of a file is >= 0, so the all-reduce(MAX) over blocks whose foreign slots are -1 is a gather."""
    b = int(np.float64(v).view(np.uint64))
    k = (~b) & 0xFFFFFFFFFFFFFFFF if (b >> 63) else (b | 0x8000000000000000)
    return k >> 42, (k >> 21) & 0x1FFFFF, k & 0x1FFFFF


def write_team_file(block, max_targets, rank, slot, value, sel_rel):
    """What rank `rank` files for team slot `slot`: its best objective and the members' child ordinals (a numpy / torch int32 block)."""
    o = max_targets + (rank * TEAM_MAX + slot) * XT_WORDS
    k0, k1, k2 = value_key(value)
    block[o], block[o + 1], block[o + 2], block[o + 3] = k0, k1, k2, len(sel_rel)
    for i, s_ in enumerate(sel_rel):
        block[o + 4 + i] = int(s_)


def team_winners(block, max_targets, world):
    """The vote every rank takes on the merged block: per slot the smallest value, ties to the lowest rank -> {slot: (rank, selections)}."""
    b = np.asarray(block)
    out = {}
    for slot in range(TEAM_MAX):
        best = None
        for r in range(world):
            o = max_targets + (r * TEAM_MAX + slot) * XT_WORDS
            if b[o + 3] < 0:
                continue
            key = (int(b[o]) << 42) | (int(b[o + 1]) << 21) | int(b[o + 2])
            if best is None or key < best[0]:
                best = (key, r, b[o + 4:o + 4 + int(b[o + 3])].tolist())
        if best is not None:
            out[slot] = (best[1], best[2])
    return out


class ClusterShardedTracker:
    """ONE tracker on `shard_n` devices: wraps a `pymht_amd.tracker.Tracker` (rank `shard_i`'s copy of the forest) and steps it with
    `mht_forest_step_sharded_begin` -> exchange -> `mht_forest_step_sharded_end`.

    exchange(sel_rel): combines the ranks' selection arrays in place (default: `merge_selections` over torch.distributed; tests that
    run two shards inside one process pass a function that takes the element-wise maximum of the two tensors).
    Every rank must be given the same targets and the same scans; track initiation (step 7) runs replicated on every rank."""

    def __init__(self, tracker, shard_n, shard_i, exchange=None, dist=None, always_exchange=False):
        import torch
        self.trk, self.shard_n, self.shard_i = tracker, int(shard_n), int(shard_i)
        self.exchange = exchange if exchange is not None else (lambda t: merge_selections(t, dist, always_exchange))
        # the exchange block: a selection per target slot, then every rank's files for the clusters searched by teams ACROSS the ranks (a gating
        # graph that is one big component: mht_forest_step_sharded_begin2) -- one all-reduce(MAX) over the whole block merges the one and gathers the other
        import ctypes as C
        from . import _lib
        nw = C.c_int32(0)
        _lib.check(tracker._lib.mht_forest_sharded_words(tracker._ctx.handle, self.shard_n, C.byref(nw)))
        self.n_words = nw.value
        self.sel_rel = torch.full((self.n_words,), -1, dtype=torch.int32, device=tracker._ctx.device)

    def begin(self, scanList, aisList=None, **kwargs):
        """Grow, cluster and this rank's share of the ILPs (asynchronous).  `pruneSimilar=True` (tracker.py:230): similar-state
        pruning of the lone targets, replicated on every rank like grow and clustering.  `aisList` (a Tracker made with aisAided=True):
        the AIS-aided children (tracker.py:417-552) are made on every rank as well -- every rank must be given the same messages."""
        from . import _lib
        trk = self.trk
        trk._drain()
        self._tic = {'Total': __import__('time').time()}
        self._ais_list = aisList
        self._z = trk._accept_scan(scanList, aisList, kwargs)
        trk._set_prune_similar(bool(kwargs.get('pruneSimilar', False)))
        if aisList is not None and len(aisList) > 0:
            trk._arm_ais(scanList, aisList, self._z.shape[0], bool(kwargs.get('aisInitialization', True)))
            trk._last_ais_scan = len(trk.__scanHistory__) + 1
        zd = trk._upload_scan(self._z)
        self.sel_rel.fill_(-1)      # (the device resets the live targets' entries itself; this also clears slots of targets long gone)
        _lib.check(trk._lib.mht_forest_step_sharded_begin2(trk._ctx.handle, zd, self._z.shape[0], self.shard_n, self.shard_i,
                                                           self.sel_rel.data_ptr(), self.n_words))
        self._scan = scanList

    def end(self):
        """After the exchange: the per-target end of the scan for all targets, step 7, report."""
        from . import _lib
        trk = self.trk
        _lib.check(trk._lib.mht_forest_step_sharded_end(trk._ctx.handle, self.sel_rel.data_ptr()))
        trk._leaf_time = float(self._scan.time)      # (what the next scan's AIS messages are timed against, tracker.py:449)
        trk._after_step(self._scan, self._z, self._ais_list, self._tic)

    def addMeasurementList(self, scanList, aisList=None, **kwargs):
        self.begin(scanList, aisList, **kwargs)
        self.exchange(self.sel_rel)
        self.end()
