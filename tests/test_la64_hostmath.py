"""CPU: the float64 covariance chain of a promoted target and the dgesv restatement (csrc/mht_la64.h, host build in tests/hostmath) against
known-answer vectors recorded from the reference's kalman.predict / kalman.precalc on float64 batches and from np.linalg.inv
(tests/golden/g22_cov64.npz, oracle/gen_golden.py::gen_g22) -- bit for bit."""
import ctypes as C
import os

import numpy as np


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_cov_chain64_matches_reference_vectors(gold_dir, hostmath):
    g = np.load(os.path.join(gold_dir, "g22_cov64.npz"))
    A, Q, Cm, R = (np.ascontiguousarray(g[k], dtype=np.float32) for k in ("A", "Q", "C", "R"))
    for ci in range(int(g["n_cases"])):
        p = "c%d_" % ci
        P = np.ascontiguousarray(g[p + "P"])
        n = len(P)
        oPb, oPh, oS, oSi, oK = np.zeros((n, 4, 4)), np.zeros((n, 4, 4)), np.zeros((n, 2, 2)), np.zeros((n, 2, 2)), np.zeros((n, 4, 2))
        hostmath.mht_host_cov_chain64(_p(A), _p(Q), _p(Cm), _p(R), n, _p(P), _p(oPb), _p(oPh), _p(oS), _p(oSi), _p(oK))
        for name, got in (("P_bar", oPb), ("P_hat", oPh), ("S", oS), ("S_inv", oSi), ("K", oK)):
            assert np.array_equal(got, g[p + name]), (ci, name)


def test_inv_lapack_matches_numpy_vectors(gold_dir, hostmath):
    g = np.load(os.path.join(gold_dir, "g22_cov64.npz"))
    hostmath.mht_host_inv_lapack.restype = C.c_double
    for n in (2, 4):
        mats, want = g["inv%d_in" % n], g["inv%d_out" % n]
        out = np.zeros((n, n))
        for m, w in zip(mats, want):
            det = hostmath.mht_host_inv_lapack(n, _p(np.ascontiguousarray(m)), _p(out))
            assert np.array_equal(out, w), n
            assert abs(det - np.linalg.det(m)) <= 1e-12 * abs(det)
