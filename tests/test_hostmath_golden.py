"""CPU: the kernel arithmetic (pymht_amd/csrc/mht_math.h, host build) against the golden vectors -- bit for bit
for everything that decides gating and for the Kalman states; 1 ulp(f32) on the NLLR constant."""
import ctypes as C
import os
import numpy as np

from util import NLLR_ATOL


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run_host(lib, g, x, P, z, P_d):
    f32 = x.dtype == np.float32
    n, M = x.shape[0], z.shape[0]
    xd, P, z = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(P), np.ascontiguousarray(z)
    o = dict(x_bar=np.zeros((n, 4)), P_bar=np.zeros((n, 4, 4), np.float32), P_hat=np.zeros((n, 4, 4), np.float32),
             S=np.zeros((n, 2, 2), np.float32), S_inv=np.zeros((n, 2, 2), np.float32), K=np.zeros((n, 4, 2), np.float32),
             nis=np.zeros((n, M)), gate=np.zeros((n, M), np.uint8), x_hat=np.zeros((n, M, 4)), nllr=np.zeros((n, M)))
    lib.mht_host_process(_p(g["A"]), _p(g["Q"]), _p(g["C"]), _p(g["R"]), C.c_double(float(g["eta2"])),
                         C.c_double(float(g["lambda_ex"])), int(f32), n, M, _p(xd), _p(P), _p(z), C.c_double(P_d),
                         *[_p(o[k]) for k in ("x_bar", "P_bar", "P_hat", "S", "S_inv", "K", "nis", "gate", "x_hat", "nllr")])
    return o


def test_hostmath_matches_reference_vectors(gold_dir, hostmath):
    g = np.load(os.path.join(gold_dir, "g1_kalman.npz"))
    g = {k: g[k] for k in g.files}
    for c in range(int(g["n_cases"])):
        k = lambda s: g["c%d_%s" % (c, s)]
        o = run_host(hostmath, g, k("x"), k("P"), k("z"), float(k("P_d")))
        n, M = o["gate"].shape
        assert np.array_equal(o["x_bar"], k("x_bar").astype(np.float64)), c
        for name in ("P_bar", "P_hat", "S", "S_inv", "K"):
            assert np.array_equal(o[name], k(name)), (c, name)
        rp, ci = k("row_ptr"), k("col_idx")
        mask = np.zeros((n, M), bool)
        for i in range(n):
            mask[i, ci[rp[i]:rp[i + 1]]] = True
        assert np.array_equal(mask, o["gate"].astype(bool)), c                       # gating: bit-exact
        assert np.array_equal(o["nis"][mask], k("nis_gated").astype(np.float64)), c  # NIS: bit-exact
        assert np.array_equal(o["x_hat"][mask], k("x_hat").astype(np.float64)), c    # states: bit-exact
        assert np.allclose(o["nllr"][mask], k("nllr").astype(np.float64), rtol=0, atol=NLLR_ATOL), c


def test_hostmath_six_state_matches_reference_vectors(gold_dir, hostmath):
    """The dimension-generic arithmetic (mht_math.h::predict_precalc_x, NX = 6) against known-answer vectors made with the reference's
    own dimension-generic kalman module (tests/golden/g11_kalman6.npz): bit for bit, NLLR constant to 1 ulp(f32)."""
    g = np.load(os.path.join(gold_dir, "g11_kalman6.npz"))
    g = {k: g[k] for k in g.files}
    for c in range(int(g["n_cases"])):
        k = lambda s: g["c%d_%s" % (c, s)]
        x, P, z = k("x"), np.ascontiguousarray(k("P")), np.ascontiguousarray(k("z"))
        n, M = x.shape[0], z.shape[0]
        o = dict(x_bar=np.zeros((n, 6)), P_bar=np.zeros((n, 6, 6), np.float32), P_hat=np.zeros((n, 6, 6), np.float32),
                 S=np.zeros((n, 2, 2), np.float32), S_inv=np.zeros((n, 2, 2), np.float32), K=np.zeros((n, 6, 2), np.float32),
                 gate=np.zeros((n, M), np.uint8), x_hat=np.zeros((n, M, 6)), nllr=np.zeros((n, M)))
        hostmath.mht_host_process_x6(_p(g["A"]), _p(g["Q"]), _p(g["C"]), _p(g["R"]), C.c_double(float(g["eta2"])), C.c_double(float(g["lambda_ex"])),
                                     int(x.dtype == np.float32), n, M, _p(np.ascontiguousarray(x, dtype=np.float64)), _p(P), _p(z), C.c_double(float(k("P_d"))),
                                     *[_p(o[q]) for q in ("x_bar", "P_bar", "P_hat", "S", "S_inv", "K", "gate", "x_hat", "nllr")])
        assert np.array_equal(o["x_bar"], k("x_bar").astype(np.float64)), c
        for name in ("P_bar", "P_hat", "S", "S_inv", "K"):
            assert np.array_equal(o[name], k(name)), (c, name)
        rp, ci = k("row_ptr"), k("col_idx")
        mask = np.zeros((n, M), bool)
        for i in range(n):
            mask[i, ci[rp[i]:rp[i + 1]]] = True
        assert np.array_equal(mask, o["gate"].astype(bool)), c
        assert np.array_equal(o["x_hat"][mask], k("x_hat").astype(np.float64)), c
        assert np.allclose(o["nllr"][mask], k("nllr").astype(np.float64), rtol=0, atol=NLLR_ATOL), c


def test_sum1d_is_numpys_reduction(hostmath):
    """The mean cumulativeNLLR of fused hypotheses (pyTarget.py:386-387) is np.mean of a 1-D array: NumPy adds < 8 elements one after
    the other and uses eight running sums from 8 on.  Sum1D (mht_math.h; prune_similar_kernel) must give the same bits for every
    length up to NumPy's block size, in both dtypes."""
    hostmath.mht_host_sum1d_f64.restype = C.c_double
    hostmath.mht_host_sum1d_f32.restype = C.c_float
    rng = np.random.default_rng(3)
    for n in list(range(1, 41)) + [63, 64, 65, 100, 127, 128]:
        for _ in range(40):
            a = rng.normal(size=n) * 10.0 ** rng.integers(-3, 4)
            assert hostmath.mht_host_sum1d_f64(_p(a), n) == float(np.add.reduce(a)), n
            b = a.astype(np.float32)
            assert np.float32(hostmath.mht_host_sum1d_f32(_p(b), n)) == np.add.reduce(b), n


def test_single_leaf_single_hit_orders(gold_dir, hostmath):
    """g15: calls with ONE leaf / leaves with ONE gated measurement, where NumPy hands the product to BLAS gemv (not gemm) and the rows
    are not FMA chains (mht_math.h::gemv_row).  Known answers from the reference's own kalman module, 4-state (both state dtypes, CV
    model and a dense-R variant) and 6-state (float64): x_bar, S^-1, K, P_hat, gating and x_hat bit for bit."""
    g = np.load(os.path.join(gold_dir, "g15_single.npz"))
    n_single_hit = 0
    for grp in range(int(g["n_groups"])):
        k = lambda s: g["g%d_%s" % (grp, s)]
        nx = int(k("nx"))
        X, Pin, Z, Mi = k("x"), k("P"), k("z"), k("M")
        f32 = X.dtype == np.float32
        for c in range(X.shape[0]):
            M = int(Mi[c])
            z = np.ascontiguousarray(Z[c, :M])
            xd = np.ascontiguousarray(X[c:c + 1], dtype=np.float64)
            P = np.ascontiguousarray(Pin[c:c + 1])
            o = dict(x_bar=np.zeros((1, nx)), P_bar=np.zeros((1, nx, nx), np.float32), P_hat=np.zeros((1, nx, nx), np.float32),
                     S=np.zeros((1, 2, 2), np.float32), S_inv=np.zeros((1, 2, 2), np.float32), K=np.zeros((1, nx, 2), np.float32),
                     nis=np.zeros((1, M)), gate=np.zeros((1, M), np.uint8), x_hat=np.zeros((1, M, nx)), nllr=np.zeros((1, M)))
            args = [_p(np.ascontiguousarray(k("A"))), _p(np.ascontiguousarray(k("Q"))), _p(np.ascontiguousarray(k("C"))), _p(np.ascontiguousarray(k("R"))),
                    C.c_double(float(g["eta2"])), C.c_double(float(g["lambda_ex"])), int(f32), 1, M, _p(xd), _p(P), _p(z), C.c_double(float(g["P_d"]))]
            if nx == 4:
                hostmath.mht_host_process(*args, *[_p(o[q]) for q in ("x_bar", "P_bar", "P_hat", "S", "S_inv", "K", "nis", "gate", "x_hat", "nllr")])
            else:
                hostmath.mht_host_process_x6(*args, *[_p(o[q]) for q in ("x_bar", "P_bar", "P_hat", "S", "S_inv", "K", "gate", "x_hat", "nllr")])
            assert np.array_equal(o["x_bar"][0], k("x_bar")[c].astype(np.float64)), (grp, c)
            assert np.array_equal(o["S_inv"][0], k("S_inv")[c]) and np.array_equal(o["K"][0], k("K")[c]) and np.array_equal(o["P_hat"][0], k("P_hat")[c]), (grp, c)
            gate = k("gate")[c, :M]
            assert np.array_equal(o["gate"][0].astype(bool), gate), (grp, c)
            assert np.array_equal(o["x_hat"][0][gate], k("x_hat")[c, :M][gate].astype(np.float64)), (grp, c, int(gate.sum()))
            assert np.allclose(o["nllr"][0][gate], k("nllr")[c, :M][gate], rtol=0, atol=NLLR_ATOL), (grp, c)
            n_single_hit += int(gate.sum() == 1)
    assert n_single_hit > 300
