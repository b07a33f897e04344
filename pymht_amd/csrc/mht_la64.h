// float64 linear algebra of the reference's covariance chain where it runs in float64: AIS-updated targets (pymht/models/ais.py:4 --
// `C = np.eye(...)` is float64, so everything downstream of kalman.precalc(ais.C, ...) in Tracker.__fuseRadarAndAis, pymht/tracker.py:451-487,
// is float64, and np.array([...P_0...]) promotes the target's WHOLE batch from the next scan on, tracker.py:859-870).
//
// np.linalg.inv of a float64 N x N matrix (N = 2: the radar innovation covariance, N = 4: the AIS one) is LAPACK dgesv with the identity as
// right-hand side (numpy/linalg/umath_linalg.cpp: inv -> call_gesv).  What OpenBLAS 0.3.29 (the pinned numpy 2.2.6 wheel's) executes for a
// matrix this small, restated operation by operation -- probed against numpy itself on 10 000 random symmetric, nearly symmetric and
// general matrices per size, 0 differences (tools/probe/lapack_order_probe.py; SkylakeX kernel set, the one the fixtures were recorded with
// and -- AVX-512 hosts -- the GPU box's):
//   dgetf2 (lapack/getf2/getf2.c; dgetrf_single falls through to it below 2 x GEMM_UNROLL_N columns) is LEFT-looking, column by column:
//     the earlier row exchanges applied to the column; its upper part by DOT (one FMA chain from 0, k ascending: the first product is
//     rounded on its own) SUBTRACTED from the entry (a separate rounding -- not fused); its lower part by GEMV_N's scalar tail (rows < 4:
//     temp = FMA chain from 0 over the columns, then y += (-1) * temp); pivot = FIRST entry of maximal magnitude (IAMAX); the rows
//     exchanged over the columns 0 .. j; the sub-diagonal scaled by the RECIPROCAL of the pivot (SCAL with 1 / pivot).
//   dgetrs: the exchanges applied to the identity (LASWP), then dtrsm_LNLU / dtrsm_LNUN, whose inner "solve" is RIGHT-looking with a
//     pre-inverted diagonal (the packing routine stores 1 / u_ii): x_i = b_i * (1 / u_ii), then b_k = fma(-x_i, u_ki, b_k) for the rows
//     still to solve (the compiler contracts `c -= bb * a` of kernel/generic/trsm_kernel_L?.c for an FMA target).
// The Haswell / Zen kernel set of the same OpenBLAS differs in the last place (its trsm solve is not fused): the recorded fixtures
// (tests/golden: G18-G18f, G19, G22) are what pins this header; comparisons against a LIVE oracle are exact on hosts whose numpy
// runs the SkylakeX set and 1e-12 relative elsewhere (tests/util.py::live_numpy_f64_is_pinned).
#pragma once
#include "mht_math.h"

namespace mht {

// out = s^-1 (both row-major N x N); returns det(s) = sign * prod(u_ii) (numpy's det goes through exp(sum(log|u_ii|)): scores only, to tolerance)
template <int N>
MHT_HD double inv_lapack(const double* s, double* out) {
    double a[N][N];      // a[r][c]
    int piv[N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) a[r][c] = s[r * N + c];
    double det = 1.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // the row exchanges of the earlier columns, applied to this one
#pragma unroll
        for (int i = 0; i < j; ++i)
#pragma unroll
            for (int r = 0; r < N; ++r)
                if (r > i && r == piv[i]) { const double t = a[i][j]; a[i][j] = a[r][j]; a[r][j] = t; }
        // upper part of the column: b_i -= dot(L[i, 0:i], b[0:i])
#pragma unroll
        for (int i = 1; i < j; ++i) {
            double dot = a[i][0] * a[0][j];
#pragma unroll
            for (int k = 1; k < i; ++k) dot = fma(a[i][k], a[k][j], dot);
            a[i][j] = a[i][j] - dot;
        }
        // lower part: b[j:] -= A[j:, 0:j] b[0:j]
        if (j > 0) {
#pragma unroll
            for (int r = j; r < N; ++r) {
                double t = a[r][0] * a[0][j];
#pragma unroll
                for (int k = 1; k < j; ++k) t = fma(a[r][k], a[k][j], t);
                a[r][j] = a[r][j] - t;
            }
        }
        // pivot: the first entry of maximal magnitude
        int p = j;
        double best = fabs(a[j][j]);
#pragma unroll
        for (int r = j + 1; r < N; ++r)
            if (fabs(a[r][j]) > best) { best = fabs(a[r][j]); p = r; }
        piv[j] = p;
        if (p != j) {
            det = -det;
#pragma unroll
            for (int r = 0; r < N; ++r)
                if (r > j && r == p)
#pragma unroll
                    for (int c = 0; c <= j; ++c) { const double t = a[j][c]; a[j][c] = a[r][c]; a[r][c] = t; }
        }
        det *= a[j][j];
        const double rp = 1.0 / a[j][j];
#pragma unroll
        for (int r = j + 1; r < N; ++r) a[r][j] = a[r][j] * rp;
    }
    // right-hand side: the identity with the row exchanges applied
    double b[N][N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) b[r][c] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int r = 0; r < N; ++r)
            if (r > i && r == piv[i])
#pragma unroll
                for (int c = 0; c < N; ++c) { const double t = b[i][c]; b[i][c] = b[r][c]; b[r][c] = t; }
    double rd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) rd[i] = 1.0 / a[i][i];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        // L y = b (unit diagonal), right-looking
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = i + 1; k < N; ++k) b[k][c] = fma(-b[i][c], a[k][i], b[k][c]);
        // U x = y
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            const double x = b[i][c] * rd[i];
            b[i][c] = x;
#pragma unroll
            for (int k = 0; k < i; ++k) b[k][c] = fma(-x, a[k][i], b[k][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
        for (int c = 0; c < N; ++c) out[r * N + c] = b[r][c];
    return det;
}

// The measurement-independent covariance chain (kalman.py:62, :90-93) of a hypothesis whose batch NumPy has promoted to float64: the
// float32 model matrices cast (exactly) to float64, every product a dgemm -- one FMA chain per element, k ascending, like the float32
// chain of mht_math.h::cov_chain -- S^-1 by inv_lapack<2>.
struct CovChain64 {
    double P_bar[NP];
    double P_hat[NP];
    double K[NK];
    double S[4];
    double S_inv[4];
};
MHT_HD void cov_chain64(const Model& m, const double* P, CovChain64& o, bool with_phat = true) {
    double AP[NP], At[NP], APA[NP];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) At[i * NX + j] = (double)m.A[j * NX + i];
    gemm_chain<double, float, double, NX, NX, NX>(m.A, P, AP);
    gemm_chain<double, double, double, NX, NX, NX>(AP, At, APA);
#pragma unroll
    for (int i = 0; i < NP; ++i) o.P_bar[i] = APA[i] + (double)m.Q[i];
    double Ct[NK], CP[NK], CPC[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) Ct[j * 2 + i] = (double)m.C[i * NX + j];
    gemm_chain<double, float, double, 2, NX, NX>(m.C, o.P_bar, CP);
    gemm_chain<double, double, double, 2, NX, 2>(CP, Ct, CPC);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.S[i] = CPC[i] + (double)m.R[i];
    inv_lapack<2>(o.S, o.S_inv);
    double PCt[NK];
    gemm_chain<double, double, double, NX, NX, 2>(o.P_bar, Ct, PCt);
    gemm_chain<double, double, double, NX, 2, 2>(PCt, o.S_inv, o.K);
    if (!with_phat) return;
    double KC[NP], KCP[NP];
    gemm_chain<double, double, float, NX, 2, NX>(o.K, m.C, KC);
    gemm_chain<double, double, double, NX, NX, NX>(KC, o.P_bar, KCP);
#pragma unroll
    for (int i = 0; i < NP; ++i) o.P_hat[i] = o.P_bar[i] - KCP[i];
}

// kalman.py:19 for a float64 S: ln(lambda_ex sqrt(det(2 pi S)) / P_d), all float64 (numpy's det is exp(sum(log|u_ii|)) of the LU of 2 pi S:
// the product of the pivots used here differs from it by an ulp or two of float64 -- scores are compared to tolerance)
MHT_HD double nllr_const64(const double* S, double lambda_ex, double P_d) {
    const double two_pi = 2.0 * 3.141592653589793;
    double s2[4] = {two_pi * S[0], two_pi * S[1], two_pi * S[2], two_pi * S[3]}, inv[4];
    const double det = inv_lapack<2>(s2, inv);
    return log((lambda_ex * sqrt(det)) / P_d);
}

}  // namespace mht
