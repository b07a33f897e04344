"""GPU: the dimension-generic gate seam `mht_gate_scan_x` through the C ABI -- a 6-state model against known-answer vectors made
with the reference's own dimension-generic kalman module (tests/golden/g11_kalman6.npz, BASELINE config 5's state dimension), the
4-state model against the G1 vectors, edge shapes against the oracle."""
import os
import numpy as np
import pytest

import mht_oracle as orc
from util import NLLR_ATOL, flags_for

pytestmark = pytest.mark.gpu


def _check(r, k, c):
    assert np.array_equal(r["row_ptr"], k("row_ptr")) and np.array_equal(r["col_idx"], k("col_idx")), c      # gating: bit-exact
    assert np.array_equal(r["x_bar"], k("x_bar").astype(np.float64)), c
    for name in ("P_bar", "P_hat", "S", "S_inv", "K"):
        assert np.array_equal(r[name], k(name)), (c, name)
    assert np.array_equal(r["x_hat"], k("x_hat").astype(np.float64)), c
    assert np.allclose(r["nllr"], k("nllr").astype(np.float64), rtol=0, atol=NLLR_ATOL), c


@pytest.mark.parametrize("fixture", ["g11_kalman6", "g1_kalman"])
def test_gate_x_matches_reference_vectors(gpu_ctx, gold_dir, fixture):
    from pymht_amd.device import process_leaf_nodes_x
    g = np.load(os.path.join(gold_dir, fixture + ".npz"))
    for c in range(int(g["n_cases"])):
        k = lambda s: g["c%d_%s" % (c, s)]
        x = k("x")
        r = process_leaf_nodes_x(gpu_ctx, g["A"], g["Q"], g["C"], g["R"], float(g["eta2"]), float(g["lambda_ex"]), x, k("P"),
                                 np.full(len(x), float(k("P_d"))), flags_for(x), k("z"))
        _check(r, k, (fixture, c))


def test_gate_x_constant_turn_matches_reference_vectors(gpu_ctx, gold_dir):
    """BASELINE config 5's model family -- a STATE-DEPENDENT transition (constant turn, six states; pymht_amd/models/ct.py): every leaf its own
    A and its own covariance chain.  Known answers made with the reference's per-hypothesis functions kalman.predict_single + kalman.precalc
    (g21, oracle/gen_golden.py::gen_g21), through `mht_gate_scan_x` with mht_model_x.transition = 1: the gating index sets exactly; the
    matrices and states bit for bit wherever the device's sin / cos agree with the host's to the last float32 bit of A (checked: A itself is
    compared first), and to 1e-6 otherwise."""
    from pymht_amd.device import process_leaf_nodes_x
    from pymht_amd.models import ct
    g = np.load(os.path.join(gold_dir, "g21_ct6.npz"))
    T = float(g["period"])
    n_exact = n_all = 0
    for c in range(int(g["n_cases"])):
        k = lambda s: g["c%d_%s" % (c, s)]
        x = k("x")
        r = process_leaf_nodes_x(gpu_ctx, ct.Phi(T, 0.0), g["Q"], g["C"], g["R"], float(g["eta2"]), float(g["lambda_ex"]), x, k("P"),
                                 np.full(len(x), float(k("P_d"))), flags_for(x), k("z"), ct_period=T)
        assert np.array_equal(r["row_ptr"], k("row_ptr")) and np.array_equal(r["col_idx"], k("col_idx")), c      # gating: exact
        exact = all(np.array_equal(r[name], k(name).astype(r[name].dtype)) for name in ("x_bar", "P_bar", "P_hat", "S", "S_inv", "K"))
        n_exact += int(exact)
        n_all += 1
        if exact:
            assert np.array_equal(r["x_hat"], k("x_hat").astype(np.float64)), c
        else:      # (a last-bit difference of sin / cos that survived the rounding of A to float32)
            assert np.allclose(r["x_bar"], k("x_bar").astype(np.float64), rtol=1e-6, atol=1e-9), c
            for name in ("P_bar", "P_hat", "S", "S_inv", "K"):
                assert np.allclose(r[name], k(name), rtol=1e-5, atol=1e-7), (c, name)
            assert np.allclose(r["x_hat"], k("x_hat").astype(np.float64), rtol=1e-6, atol=1e-6), c
        assert np.allclose(r["nllr"], k("nllr").astype(np.float64), rtol=0, atol=NLLR_ATOL if exact else 1e-5), c
    assert n_exact >= n_all - 2, (n_exact, n_all)


def test_gate_x_constant_turn_at_config_5_size(gpu_ctx):
    """The constant-turn seam at the size BASELINE config 5 names (2 000 targets' worth of leaves: 100 000 x 2 000 measurements, six
    states, every leaf its own transition and covariance chain), checked through what does not depend on the size: a random sample of
    300 leaves against the oracle's per-leaf restatement (same gating sets; values as in the known-answer test), and for EVERY gated pair
    of the call the gate's own inequality recomputed in float64 from the returned S^-1 and predicted measurement; the CSR is ordered;
    no pair of a sampled leaf is missing."""
    from pymht_amd.device import process_leaf_nodes_x
    from pymht_amd.models import ct
    rng = np.random.default_rng(4242)
    n, M, T, eta2, lam = 100_000, 2000, 2.5, 5.99, 2e-5
    x = np.concatenate([rng.uniform(-20000, 20000, size=(n, 2)), rng.normal(0, 8, size=(n, 2)), rng.normal(0, 0.05, size=(n, 1)), rng.normal(0, 1e-3, size=(n, 1))], axis=1)
    x[rng.uniform(size=n) < 0.15, 4] = 0.0
    P = np.repeat(ct.P0[None].astype(np.float32), n, axis=0) * rng.uniform(0.5, 2.0, size=(n, 1, 1)).astype(np.float32)
    P = (P + P.transpose(0, 2, 1)) / np.float32(2)
    z = rng.uniform(-20000, 20000, size=(M, 2)).astype(np.float32)
    near = rng.integers(0, n, size=M // 2)      # half of the measurements sit next to a predicted position
    xb = np.array([ct.Phi(T, x[i, 4]).astype(np.float64).dot(x[i]) for i in near])
    z[: M // 2] = (xb[:, :2] + rng.normal(0, 6.0, size=(M // 2, 2))).astype(np.float32)
    Q, Cm, R = ct.Q(T), ct.C_RADAR, ct.R_RADAR()
    pd = np.full(n, 0.9)
    r = process_leaf_nodes_x(gpu_ctx, ct.Phi(T, 0.0), Q, Cm, R, eta2, lam, x, P, pd, flags_for(x), z, ct_period=T)
    rp, ci = r["row_ptr"], r["col_idx"]
    assert rp[0] == 0 and np.all(np.diff(rp) >= 0) and rp[-1] == len(ci) > M // 4
    leaf_of = np.repeat(np.arange(n), np.diff(rp))
    assert np.all((np.diff(ci) > 0) | (np.diff(leaf_of) > 0))                      # measurements ascending inside a leaf
    # every gated pair passes the gate it was admitted by (float64 recomputation; a hair of slack for the float32 -> float64 path)
    zh = r["x_bar"][:, :2]                                                          # (C picks the position)
    zt = z.astype(np.float64)[ci] - zh[leaf_of]
    Si = r["S_inv"].astype(np.float64)[leaf_of]
    nis = np.einsum("pi,pij,pj->p", zt, Si, zt)
    assert np.all(nis <= eta2 * (1 + 1e-5)), float(nis.max())
    # a sample of leaves against the oracle's per-leaf restatement (kalman.predict_single + precalc on a batch of one)
    pick = np.sort(np.concatenate([rng.choice(n, 250, replace=False), near[:50]]))
    o = orc.process_leaves_ct(ct.Phi, T, Q, Cm, R, eta2, lam, x[pick], P[pick], [0.9] * len(pick), z)
    n_exact = 0
    for j, i in enumerate(pick):
        got = ci[rp[i]:rp[i + 1]]
        assert np.array_equal(got, np.asarray(o["idx"][j], dtype=np.int64)), (i, got, o["idx"][j])
        exact = all(np.array_equal(r[name][i], np.asarray(o[name][j]).astype(r[name].dtype)) for name in ("x_bar", "P_bar", "P_hat", "S", "S_inv", "K"))
        n_exact += int(exact)
        assert np.allclose(r["x_bar"][i], o["x_bar"][j], rtol=1e-6, atol=1e-9)
        for name in ("P_bar", "P_hat", "S", "S_inv", "K"):
            assert np.allclose(r[name][i], o[name][j], rtol=1e-5, atol=1e-7), (i, name)
        if len(got):
            assert np.allclose(r["x_hat"][rp[i]:rp[i + 1]], o["x_hat"][j], rtol=1e-6, atol=1e-6)
            assert np.allclose(r["nllr"][rp[i]:rp[i + 1]], o["nllr"][j], rtol=0, atol=1e-5)
    assert n_exact >= len(pick) * 0.8, (n_exact, len(pick))                        # (bit for bit wherever sin / cos round to the same float32 A)
    assert sum(len(v) for v in o["idx"]) >= 40                                     # (the sample does gate something)


@pytest.mark.parametrize("n,M,seed", [(0, 5, 1), (3, 0, 2), (1, 1, 3), (700, 65, 4), (129, 2048, 5)])
def test_gate_x_six_state_edge_shapes_vs_oracle(gpu_ctx, n, M, seed):
    from pymht_amd.device import process_leaf_nodes_x
    from pymht_amd.models import ca
    from trace_util import states_close
    rng = np.random.default_rng(seed)
    A, Q, Cm, R = ca.Phi(2.5), ca.Q(2.5), ca.C_RADAR, ca.R_RADAR()
    x = np.concatenate([rng.uniform(-500, 500, size=(n, 2)), rng.normal(0, 5, size=(n, 2)), rng.normal(0, 0.2, size=(n, 2))], axis=1)
    P = np.array([ca.P0] * n).reshape(n, 6, 6)
    z = rng.uniform(-500, 500, size=(M, 2))
    if n and M:
        xb = A.astype(np.float64).dot(x.T).T
        pick = rng.integers(0, n, size=M)
        near = rng.uniform(size=M) < 0.5
        z[near] = xb[pick[near], 0:2] + rng.normal(0, 7.0, size=(int(near.sum()), 2))
    z = z.astype(np.float32)
    r = process_leaf_nodes_x(gpu_ctx, A, Q, Cm, R, 5.99, 1.2e-4, x, P, np.full(n, 0.8), flags_for(x), z)
    if n == 0:
        assert r["row_ptr"].tolist() == [0]
        return
    o = orc.process_leaves(A, Q, Cm, R, 5.99, 1.2e-4, x, P, [0.8] * n, z.reshape(-1, 2))
    assert [tuple(r["col_idx"][r["row_ptr"][i]:r["row_ptr"][i + 1]].tolist()) for i in range(n)] == [tuple(i.tolist()) for i in o["idx"]]
    # (a live oracle on this box's CPU: states to 1e-6 relative, index sets exactly -- the golden vectors pin the bits)
    assert np.allclose(r["x_bar"], o["x_bar"], rtol=1e-6, atol=1e-9) and np.allclose(r["P_hat"], o["P_hat"], rtol=1e-6, atol=0)
    if len(r["col_idx"]):
        assert np.allclose(r["x_hat"], np.concatenate(o["x_hat"], axis=0), rtol=1e-6, atol=1e-6)
        assert np.allclose(r["nllr"], np.concatenate(o["nllr"]), rtol=0, atol=NLLR_ATOL)


def test_single_leaf_single_hit_orders_both_seams(gpu_ctx, gold_dir):
    """g15 (oracle/gen_golden.py::gen_g15): ONE leaf per call / ONE gated measurement per leaf -- the shapes NumPy hands to BLAS gemv
    instead of gemm (kalman.py:60, :88, :50), every target's first scan.  Known answers from the reference's kalman module, through
    `mht_gate_scan_x` (4 and 6 states) and, for the 4-state groups, `mht_gate_scan`: bit for bit, dense-R variants included (S^-1 of a
    non-diagonal S is numpy.linalg's float64 inverse rounded to float32)."""
    from pymht_amd.device import make_model, process_leaf_nodes, process_leaf_nodes_x
    g = np.load(os.path.join(gold_dir, "g15_single.npz"))
    eta2, lam, pd = float(g["eta2"]), float(g["lambda_ex"]), float(g["P_d"])
    n_one = 0
    for grp in range(int(g["n_groups"])):
        k = lambda s: g["g%d_%s" % (grp, s)]
        nx, X, Pin, Z, Mi = int(k("nx")), k("x"), k("P"), k("z"), k("M")
        model4 = make_model(k("A"), k("Q"), k("C"), k("R"), eta2, lam, pd) if nx == 4 else None
        for c in range(X.shape[0]):
            M = int(Mi[c])
            x, P, z = X[c:c + 1], Pin[c:c + 1], Z[c, :M]
            gate = k("gate")[c, :M]
            want_idx = np.nonzero(gate)[0]
            want_xhat = k("x_hat")[c, :M][gate].astype(np.float64)
            rs = [process_leaf_nodes_x(gpu_ctx, k("A"), k("Q"), k("C"), k("R"), eta2, lam, x, P, np.full(1, pd), flags_for(x), z)]
            if model4 is not None:
                rs.append(process_leaf_nodes(gpu_ctx, model4, x, P, np.zeros(1), np.full(1, pd), flags_for(x), z))
            for r in rs:
                assert np.array_equal(r["col_idx"], want_idx), (grp, c)
                assert np.array_equal(r["x_bar"][0], k("x_bar")[c].astype(np.float64)), (grp, c)
                assert np.array_equal(r["P_hat"][0], k("P_hat")[c]), (grp, c)
                assert np.array_equal(r["x_hat"], want_xhat), (grp, c, len(want_idx))
                assert np.allclose(r["nllr"], k("nllr")[c, :M][gate], rtol=0, atol=NLLR_ATOL), (grp, c)
            assert np.array_equal(rs[0]["S_inv"][0], k("S_inv")[c]) and np.array_equal(rs[0]["K"][0], k("K")[c]), (grp, c)
            n_one += int(len(want_idx) == 1)
    assert n_one > 300
