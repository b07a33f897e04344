"""GPU: seams (ii) clustering and (iii) the 0-1 ILP through the C ABI, against the oracle and the recorded instances."""
import ctypes as C
import os
import time
import numpy as np
import pytest
import torch

import mht_oracle as orc
from pymht_amd import _lib

pytestmark = pytest.mark.gpu


def gpu_cluster(ctx, sets, n_nodes):
    T = len(sets)
    words = (n_nodes + 63) // 64
    bits = np.zeros((T, words), dtype=np.uint64)
    for t, s in enumerate(sets):
        for m in s:
            bits[t, m >> 6] |= np.uint64(1) << np.uint64(m & 63)
    d = torch.from_numpy(bits.view(np.int64)).to(ctx.device)
    lab = torch.zeros(max(T, 1), dtype=torch.int32, device=ctx.device)
    _lib.check(ctx.lib.mht_cluster(ctx.handle, T, words, d.data_ptr(), lab.data_ptr()))
    return lab[:T].cpu().numpy()


@pytest.mark.parametrize("T,n_nodes,deg,seed", [(1, 64, 3, 0), (7, 100, 2, 1), (300, 3000, 4, 2), (2000, 9000, 3, 3),
                                                 (500, 200, 1, 4), (64, 64, 0, 5),
                                                 (3000, 9000, 9, 6),        # > 16384 edges: the LDS edge list spills to HBM
                                                 # tables beyond LDS (cluster_big_kernel, tables in HBM): 8 k targets x 4 k measurements, N-scan 6
                                                 # = 9 x 4096 measurement nodes (the round-2 review's size), sparse and dense; and the 16-bit limits
                                                 (8192, 36864, 5, 7), (8192, 36864, 24, 8), (6000, 65536, 3, 9), (20000, 4096, 2, 10),
                                                 # beyond every 16-bit limit (the seam's device-wide union-find, mht_uf.h: no edge records)
                                                 (20000, 131072, 4, 11), (30000, 70000, 3, 12)])
def test_cluster_matches_oracle(gpu_ctx, T, n_nodes, deg, seed):
    rng = np.random.default_rng(seed)
    sets = [set(int(v) for v in rng.integers(0, n_nodes, size=rng.integers(0, deg + 1))) for _ in range(T)]
    if T > 10:     # a long chain: worst case for label propagation
        for t in range(0, min(T, 120) - 1):
            sets[t].add(n_nodes - 1 - t)
            sets[t + 1].add(n_nodes - 1 - t)
    lab = gpu_cluster(gpu_ctx, sets, n_nodes)
    ref = orc.find_clusters(sets)
    want = np.zeros(T, dtype=np.int64)
    for cl in ref:
        want[cl] = cl[0]
    assert np.array_equal(lab, want)


def load_instances(path):
    g = np.load(path)
    out = []
    for i in range(int(g["n_inst"])):
        p = "i%03d_" % i
        ptr, rows = g[p + "col_ptr"], g[p + "col_rows"]
        out.append(dict(cols=[rows[ptr[c]:ptr[c + 1]] for c in range(len(ptr) - 1)], sizes=g[p + "sizes"],
                        cost=g[p + "cost"], sel=g[p + "sel"], obj=float(g[p + "obj"]), unique=bool(g[p + "unique"])))
    return out


def gpu_blp(ctx, inst, max_iter=200, node_limit=1 << 20):
    cols, sizes, cost = inst["cols"], inst["sizes"], inst["cost"]
    nH, nT = len(cols), len(sizes)
    depth = max(1, max(len(c) for c in cols))
    nrows = 1 + max((int(c.max()) for c in cols if len(c)), default=0)
    rows = -np.ones((depth, nH), dtype=np.int32)
    for h, c in enumerate(cols):
        rows[:len(c), h] = c
    dev = ctx.device
    gp = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(dev)
    rw = torch.from_numpy(rows).to(dev)
    cs = torch.from_numpy(np.asarray(cost, dtype=np.float64)).to(dev)
    sel = torch.zeros(nT, dtype=torch.int32, device=dev)
    obj, st, it, nd = C.c_double(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(ctx.lib.mht_solve_blp(ctx.handle, nH, nT, nrows, depth, gp.data_ptr(), rw.data_ptr(), cs.data_ptr(),
                                     max_iter, node_limit, sel.data_ptr(), C.byref(obj), C.byref(st), C.byref(it), C.byref(nd)))
    gpu_blp.last_call_s = time.perf_counter() - t0      # (the call synchronises: device time + launch, without the Python packing above)
    return sorted(sel.cpu().numpy().tolist()), obj.value, st.value, it.value, nd.value


@pytest.mark.parametrize("name", ["g4_ilp", "g6_ilp_cfg3", "g7_ilp_hard"])
@pytest.mark.parametrize("max_iter", [200, 0])
def test_blp_recorded_instances(gpu_ctx, gold_dir, name, max_iter):
    """Selections are bit-exact against the exact reference optimum (unique in every recorded instance); with
    max_iter=0 the dual ascent is skipped and the GPU branch and bound has to prove optimality on its own."""
    insts = load_instances(os.path.join(gold_dir, name + ".npz"))
    stats = {1: 0, 2: 0}
    for inst in insts:
        sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst, max_iter=max_iter)
        assert st in (1, 2)
        stats[st] += 1
        assert abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(obj)), (name, len(inst["cols"]))
        if inst["unique"]:
            assert sel == inst["sel"].tolist()
    print(name, "max_iter", max_iter, "certified/branched", stats)
    if max_iter == 0:
        assert stats[2] > 0
    # (the zig-zag pairs of g7 that the dual rounds cannot certify are settled by pair enumeration after the first round: status 1)


def test_blp_adversarial_needs_branching(gpu_ctx):
    """Odd cycle of pairwise conflicts: the LP relaxation is fractional (all 1/2), so the certificate cannot hold
    and branch and bound must close the gap.  Checked against exhaustive search."""
    rng = np.random.default_rng(11)
    for trial in range(6):
        nT = 5 + 2 * (trial % 2)
        cols, sizes, cost = [], [], []
        for t in range(nT):
            # column A uses rows (t, t+1 mod nT) -> neighbours conflict pairwise around an odd cycle
            cols += [np.array([t, (t + 1) % nT]), np.array([nT + t]), np.array([], dtype=np.int64)]
            cost += [-2.0 - 0.1 * rng.uniform(), -0.9 - 0.05 * rng.uniform(), 0.0]
            sizes.append(3)
        inst = dict(cols=cols, sizes=np.array(sizes), cost=np.array(cost))
        bs, bo, ties = orc.solve_blp_bruteforce([c.tolist() for c in cols], sizes, cost)
        sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst, max_iter=30)
        assert abs(obj - bo) < 1e-9
        if ties == 1:
            assert sel == sorted(bs)


def test_blp_hbm_storage_policy(gpu_ctx, gold_dir, monkeypatch):
    """Clusters that do not fit LDS run the same solver on HBM scratch; MHT_BLP_FORCE_HBM=1 sends every cluster there.
    Recorded instances (dual ascent and branch and bound) and a whole scan trace must still match bit for bit."""
    monkeypatch.setenv("MHT_BLP_FORCE_HBM", "1")
    for name in ("g4_ilp", "g6_ilp_cfg3", "g7_ilp_hard"):
        for max_iter in (200, 0):
            for inst in load_instances(os.path.join(gold_dir, name + ".npz"))[::3]:
                sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst, max_iter=max_iter)
                assert st in (1, 2) and abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(obj))
                assert sel == inst["sel"].tolist()
    from test_tracker_gpu import test_tracker_replays_reference_trace
    test_tracker_replays_reference_trace("g3b_trace_cfg2", gold_dir)


def test_blp_clusters_without_certificate(gpu_ctx, gold_dir):
    """tests/golden/g9_ilp_giant.npz, three clusters from dense scenarios the fuzzer found, none has a dual certificate:
    (1) 29 targets / 17 935 columns, too large for the LDS policy; (2) 34 targets / 3 039 columns with an LP gap of 0.59 spread
    over five targets -- a depth-first search with a static Lagrangian bound needs ~10^6 nodes there (it took 4.5 s and sat at
    the node limit); (3) 43 targets / 9 531 columns (HBM policy): 260 k nodes / 7 s with prices re-optimised on the first 12 levels
    only, and past the node limit inside the forest (other row numbering).  With re-optimised prices on 32 levels: 45, 32 and
    2 677 nodes."""
    for inst in load_instances(os.path.join(gold_dir, "g9_ilp_giant.npz")):
        sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst, max_iter=200)
        # (a cluster of >= 24 targets is searched by a TEAM of workgroups -- see the next test: nd sums the members' nodes, the levels above
        # the deal-out level are walked by all of them)
        assert st == 2 and 0 < nd < 15000, (len(inst["cols"]), st, it, nd)
        assert abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(obj)) and sel == inst["sel"].tolist()
        assert gpu_blp.last_call_s < 0.060, "a giant cluster took %.1f ms (round-3 bar: < 50 ms device time; the call adds launches and a read-back)" % (1e3 * gpu_blp.last_call_s)


def test_blp_team_on_hbm_scratch(gpu_ctx, gold_dir, monkeypatch):
    """G20: the giant cluster of fuzz scenario 90266 (44 targets, 7 831 columns) that reduced-cost fixing cannot cut down to what LDS
    holds, so that its branch and bound stays on HBM scratch (0.25 ms per node): a team of workgroups, each with its own copy of that
    scratch, replicates the dual phase and shares the search.  Exact optimum (HiGHS, unique), same selection with and without the team;
    the single workgroup needs 3.7 s (14.9 k nodes), the team well under a second."""
    inst = load_instances(os.path.join(gold_dir, "g20_ilp_hbm_team.npz"))[0]
    assert inst["unique"] and len(inst["sizes"]) == 44 and len(inst["cols"]) == 7831
    monkeypatch.setenv("MHT_BLP_NO_TEAMS", "0")
    sel, obj, status, iters, nodes = gpu_blp(gpu_ctx, inst, max_iter=200, node_limit=1 << 22)
    t_team = gpu_blp.last_call_s
    assert sel == inst["sel"].tolist() and abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(inst["obj"])) and status == 2
    assert t_team < 0.5, "the HBM team needed %.2f s" % t_team      # (0.23 s until round 5's batched sweeps, 0.11 s since: profiles/r05_ilp_tail.txt; the call adds launches and a read-back)
    monkeypatch.setenv("MHT_BLP_NO_TEAMS", "1")
    sel1, obj1, status1, _, nodes1 = gpu_blp(gpu_ctx, inst, max_iter=200, node_limit=1 << 22)
    assert sel1 == sel and status1 == 2
    assert t_team < 0.5 * gpu_blp.last_call_s, "team %.2f s vs single workgroup %.2f s" % (t_team, gpu_blp.last_call_s)


def test_blp_team_search_equals_single_workgroup_search(gpu_ctx, gold_dir, monkeypatch):
    """Branch and bound by a team of workgroups (csrc/mht_kernels.h: TEAM_*; tracker.py:1155-1217 is one CBC call): every member
    replicates the deterministic dual phase, the subtrees below the deal-out level are dealt out by a hash of their columns, the
    incumbent value is one shared atomic-min word, the last member to finish takes the best selection.  Same optimum and -- the
    recorded optima being unique -- the same selection as the single-workgroup search (MHT_BLP_NO_TEAMS=1), which needs 4x as long
    on the third G9 instance; and inside the forest: a dense scenario scan by scan with teams on and off."""
    insts = load_instances(os.path.join(gold_dir, "g9_ilp_giant.npz")) + load_instances(os.path.join(gold_dir, "g12_ilp_reduced.npz"))
    got = {}
    for off in ("0", "1"):
        monkeypatch.setenv("MHT_BLP_NO_TEAMS", off)
        got[off] = [gpu_blp(gpu_ctx, inst, max_iter=200) + (gpu_blp.last_call_s,) for inst in insts]
    for inst, a, b in zip(insts, got["0"], got["1"]):
        assert a[0] == b[0] == inst["sel"].tolist() and abs(a[1] - b[1]) <= 1e-12 * max(1.0, abs(b[1])) and a[2] == b[2] and a[2] in (1, 2)
    assert got["0"][2][5] < 0.6 * got["1"][2][5], "the team did not speed the 43-target instance up: %.1f vs %.1f ms" % (1e3 * got["0"][2][5], 1e3 * got["1"][2][5])
    # inside the forest (teams are formed by the launch's workgroups without a cluster of their own)
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.scenario import make_scenario
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = make_scenario(T=66, radius=201.0, lambda_phi=1.5e-4, n_scans=7, P_d=0.73, period=2.5, seed=5494)
    runs = {}
    for off in ("0", "1"):
        monkeypatch.setenv("MHT_BLP_NO_TEAMS", off)
        trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=3, eta2=9.21, useInitiator=False, maxTargets=512, maxNodes=1 << 18)
        trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
        out = []
        for z, t in zip(sc["scans"], sc["times"]):
            trk.addMeasurementList(MeasurementList(float(t), z))
            sel = trk._sel[0]
            out.append((sel["id"].tolist(), sel["sel_meas"].tolist(), sel["sel_x"].tobytes(), trk.lastScanStats["branched"], trk.lastScanStats["leaves_out"]))
        trk.close()
        runs[off] = out
    assert runs["0"] == runs["1"]
    assert sum(r[3] for r in runs["0"]) > 0      # (some cluster did branch)


def test_blp_time_limit_returns_a_feasible_selection(gpu_ctx, gold_dir, monkeypatch):
    """The wall-clock budget of a cluster's branch and bound (mht_forest_set_blp_time_limit; MHT_BLP_TIME_LIMIT_US for the stateless
    seam): the 43-target G9 cluster needs ~2 700 nodes / 150 ms for the proof; with 20 ms it comes back in time with MHT_E_LIMIT and
    a conflict-free selection within 3 % of the optimum (the incumbent of the dual phase and the first dives)."""
    inst = load_instances(os.path.join(gold_dir, "g9_ilp_giant.npz"))[2]
    monkeypatch.setenv("MHT_BLP_TIME_LIMIT_US", "20000")
    cols, sizes, cost = inst["cols"], inst["sizes"], inst["cost"]
    nH, nT = len(cols), len(sizes)
    depth = max(len(c) for c in cols)
    nrows = 1 + max(int(c.max()) for c in cols if len(c))
    rows = -np.ones((depth, nH), dtype=np.int32)
    for h, c in enumerate(cols):
        rows[:len(c), h] = c
    dev = gpu_ctx.device
    gp = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(dev)
    rw, cs = torch.from_numpy(rows).to(dev), torch.from_numpy(np.asarray(cost, dtype=np.float64)).to(dev)
    sel = torch.zeros(nT, dtype=torch.int32, device=dev)
    obj, st, it, nd = C.c_double(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = gpu_ctx.lib.mht_solve_blp(gpu_ctx.handle, nH, nT, nrows, depth, gp.data_ptr(), rw.data_ptr(), cs.data_ptr(), 200, 1 << 20, sel.data_ptr(),
                                   C.byref(obj), C.byref(st), C.byref(it), C.byref(nd))
    dt = time.perf_counter() - t0
    assert rc == _lib.MHT_E_LIMIT and st.value == 3, (rc, st.value)
    assert dt < 0.06, dt
    chosen = sel.cpu().numpy().tolist()
    used = set()
    for t, h in enumerate(chosen):
        assert gp[t].item() <= h < gp[t + 1].item()
        for m in cols[h]:
            assert int(m) not in used
            used.add(int(m))
    true_obj = float(sum(cost[h] for h in chosen))
    assert abs(true_obj - obj.value) < 1e-9 and inst["obj"] - 1e-9 <= true_obj <= inst["obj"] + 0.03 * abs(inst["obj"])


def test_blp_small_clusters_exact_search(gpu_ctx, gold_dir, monkeypatch):
    """tests/golden/g10_ilp_small_hard.npz: 3..9 near-duplicate tracks with a duality gap, the slowest ILPs of four headline
    streams (the coordinate rounds do not certify them).  The exact search over contested-row signatures (enumerate_small) must
    return the HiGHS optimum (unique in every instance), and so must the branch and bound it replaces (MHT_BLP_NO_ENUM=1)."""
    insts = load_instances(os.path.join(gold_dir, "g10_ilp_small_hard.npz"))
    searched = 0
    for inst in insts:
        sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst)
        assert st in (1, 2) and abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(obj)) and sel == inst["sel"].tolist(), (len(inst["cols"]), st, it, nd)
        searched += 1 if (st == 2 and it <= 16) else 0
    assert searched >= len(insts) // 2
    monkeypatch.setenv("MHT_BLP_NO_ENUM", "1")
    for inst in insts[::4]:
        sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst)
        assert st in (1, 2) and abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(obj)) and sel == inst["sel"].tolist()


def test_blp_giant_clusters_reduced_cost_fixing(gpu_ctx, gold_dir, monkeypatch):
    """Clusters too large for the LDS tables (G9: 29-43 targets, 3-18 k columns; G12: 26-29 targets, 2.8-4 k columns from dense fuzz
    scenarios) run their dual phase on HBM scratch; reduced-cost fixing at its prices leaves a few dozen columns, the cluster is
    rebuilt from them in LDS (with the HBM phase's incumbent) and solved there.  Same optimum as HiGHS, with the reduction and
    without it (MHT_BLP_NO_REDUCE=1: branch and bound on HBM scratch)."""
    insts = load_instances(os.path.join(gold_dir, "g12_ilp_reduced.npz")) + load_instances(os.path.join(gold_dir, "g9_ilp_giant.npz"))
    for flag in ("0", "1"):
        monkeypatch.setenv("MHT_BLP_NO_REDUCE", flag)
        for inst in insts:
            sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst)
            assert st in (1, 2) and abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(obj)), (flag, len(inst["cols"]), st, it, nd)
            if inst["unique"]:
                assert sel == inst["sel"].tolist(), (flag, len(inst["cols"]))


def test_prune_seam_matches_oracle_trees():
    """Seam (iv) mht_prune against the oracle's Node.prune_depth (pyTarget.py:343-356) on random trees: new roots and the exact
    set of surviving nodes, for windows shorter, equal to and longer than the tree."""
    import torch
    from pymht_amd import _lib
    from pymht_amd.device import Context
    rng = np.random.default_rng(11)
    ctx = Context(0)
    for trial in range(6):
        T = int(rng.integers(1, 40))
        nodes, roots = [], []

        def grow(node, depth, maxd):
            nodes.append(node)
            if depth >= maxd:
                return
            node.kids = [orc.Node(0.0, depth + 1, None, None, parent=node, meas=k) for k in range(int(rng.integers(1, 4)))]
            for k in node.kids:
                grow(k, depth + 1, maxd)
        for t in range(T):
            r = orc.Node(0.0, 0, None, None, ID=t)
            r.is_root = True
            roots.append(r)
            grow(r, 0, int(rng.integers(0, 6)))
        idx = {id(n): i for i, n in enumerate(nodes)}
        parent = np.array([-1 if n.parent is None else idx[id(n.parent)] for n in nodes], np.int32)
        sel_nodes = [r.leaves()[int(rng.integers(0, len(r.leaves())))] for r in roots]
        window = rng.integers(0, 7, size=T).astype(np.int32)
        sel = np.array([idx[id(s)] for s in sel_nodes], np.int32)
        want_root = np.array([idx[id(s.prune_depth(int(w)))] for s, w in zip(sel_nodes, window)], np.int32)
        alive = set()
        for r in roots:      # what is still reachable from the tops after the reference's pruning
            stack = [r]
            while stack:
                n = stack.pop()
                alive.add(idx[id(n)])
                if n.kids:
                    stack.extend(n.kids)
        dp, ds, dw = (torch.from_numpy(a).cuda() for a in (parent, sel, window))
        nr = torch.zeros(T, dtype=torch.int32, device="cuda")
        keep = torch.zeros(len(nodes), dtype=torch.uint8, device="cuda")
        _lib.check(ctx.lib.mht_prune(ctx.handle, len(nodes), dp.data_ptr(), T, ds.data_ptr(), dw.data_ptr(), nr.data_ptr(), keep.data_ptr()))
        ctx.synchronize()
        assert np.array_equal(nr.cpu().numpy(), want_root), trial
        assert set(np.nonzero(keep.cpu().numpy())[0].tolist()) == alive, trial
    ctx.close()


def test_blp_hbm_policy_both_row_passes_and_column_range_sources(gpu_ctx, gold_dir, monkeypatch):
    """(r5) The HBM storage policy has two forms of its passes over a cluster's rows (a thread per BIT of the row bitset up to 256 words,
    a thread per WORD beyond) and two sources of the members' column ranges (LDS for clusters of <= cap_k = 256 targets, the target
    tables in global memory beyond): recorded instances with their row numbers spread over > 16 384 nodes (the per-word form) give the
    recorded optima, and a chain of 300 conflicting targets (ranges from global memory) the exact optimum of the oracle's solver."""
    monkeypatch.setenv("MHT_BLP_FORCE_HBM", "1")
    for name in ("g6_ilp_cfg3", "g7_ilp_hard"):
        for inst in load_instances(os.path.join(gold_dir, name + ".npz"))[::5]:
            wide = dict(inst)
            wide["cols"] = [np.asarray(c, dtype=np.int64) * 131 + 7 for c in inst["cols"]]      # (injective: the same conflicts, > 256 bitset words)
            if max((int(c.max()) for c in wide["cols"] if len(c)), default=0) < 16384:
                continue
            for max_iter in (200, 0):
                sel, obj, st, it, nd = gpu_blp(gpu_ctx, wide, max_iter=max_iter)
                assert st in (1, 2) and abs(obj - inst["obj"]) <= 1e-9 * max(1.0, abs(obj)) and sel == inst["sel"].tolist()
    monkeypatch.delenv("MHT_BLP_FORCE_HBM")
    rng = np.random.default_rng(5)
    nT = 300
    cols, sizes, cost = [], [], []
    for t in range(nT):      # target t wants row t or row t + 1 (its neighbour's), or a private row, or nothing
        cols += [np.array([t]), np.array([t + 1]), np.array([nT + 1 + t]), np.array([], dtype=np.int64)]
        cost += [-3.0 - rng.uniform(), -3.0 - rng.uniform(), -1.0 - rng.uniform(), 0.0]
        sizes.append(4)
    inst = dict(cols=cols, sizes=np.array(sizes), cost=np.array(cost))
    ref_sel, ref_obj = orc.solve_blp_exact([c.tolist() for c in cols], sizes, cost)[:2]
    sel, obj, st, it, nd = gpu_blp(gpu_ctx, inst, max_iter=400)
    assert st in (1, 2), st
    assert abs(obj - float(ref_obj)) <= 1e-9 * max(1.0, abs(obj))
