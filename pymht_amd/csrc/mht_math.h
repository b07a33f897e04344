// Per-hypothesis Kalman arithmetic of the pyMHT scan path, written so that it reproduces the
// reference's NumPy/OpenBLAS results BIT FOR BIT where that is achievable (SURVEY.md section 7
// "Hard parts"): every small matrix product is evaluated as OpenBLAS's gemm micro-kernels evaluate
// it -- one accumulator per output element, k ascending, fused multiply-add -- and everything else
// (element-wise adds, the NIS reduction) as separate IEEE operations.  Compile with
// -ffp-contract=off: every fma below is explicit, nothing else may be contracted.
//
// Reference lines restated here (file:line relative to /root/reference):
//   predict   pymht/utils/kalman.py:55-64      precalc  kalman.py:82-101
//   z_tilde   kalman.py:36-40                  NIS      kalman.py:25-28   gate tracker.py:829
//   update    kalman.py:43-52                  NLLR     kalman.py:14-22
//   miss hypothesis score  pymht/pyTarget.py:319-328, hit score pyTarget.py:250
//
// The header is shared by the HIP kernels (device) and by tests/hostmath (host build used ONLY by the
// CPU test-suite to check the arithmetic against the golden vectors without a GPU).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MHT_HD __host__ __device__ __forceinline__
#else
#define MHT_HD inline
#endif

// State dimension of this BUILD of the library: 4 (the reference's CV model, models/pv.py; the default, libmht_amd.so) or 6
// (BASELINE config 5 names a 6-state model; libmht_amd6.so is the same sources compiled with -DMHT_NX=6).  The reference's kalman
// module is dimension-generic (kalman.py:55-101); everything below is written for NX states and 2 measurements.
#ifndef MHT_NX
#define MHT_NX 4
#endif

namespace mht {

constexpr int NX = MHT_NX;      // states
constexpr int NP = NX * NX;     // covariance entries
constexpr int NK = 2 * NX;      // gain entries (NX x 2)
static_assert(NX == 4 || NX == 6, "MHT_NX: 4 or 6 states");

// The linear-Gaussian model, f32 like the reference's pv module (models/pv.py:7-34).
struct Model {
    float A[NP];   // state transition Phi(T), row-major NX x NX
    float Q[NP];   // process noise
    float C[NK];   // measurement matrix 2 x NX
    float R[4];    // measurement noise 2x2
    double eta2;   // gate threshold (chi-square, 2 dof)      tracker.py:110
    double lambda_ex;  // lambda_phi + lambda_nu                tracker.py:107
};

MHT_HD float fmaT(float a, float b, float c) { return fmaf(a, b, c); }
MHT_HD double fmaT(double a, double b, double c) { return fma(a, b, c); }

// c[m x n] = a[m x k] * b[k x n], all row-major, one fma chain per element (k ascending)
template <typename TO, typename TA, typename TB, int M_, int K_, int N_>
MHT_HD void gemm_chain(const TA* a, const TB* b, TO* c) {
#pragma unroll
    for (int i = 0; i < M_; ++i)
#pragma unroll
        for (int j = 0; j < N_; ++j) {
            TO acc = (TO)a[i * K_] * (TO)b[j];
#pragma unroll
            for (int k = 1; k < K_; ++k) acc = fmaT((TO)a[i * K_ + k], (TO)b[k * N_ + j], acc);
            c[i * N_ + j] = acc;
        }
}

// Per-leaf quantities that do not depend on the measurements.
template <typename TS>
struct Predicted {
    TS x_bar[NX];
    TS z_hat[2];
    float P_bar[NP];
    float P_hat[NP];
    float K[NK];     // NX x 2
    float S[4];
    float S_inv[4];
};

// np.linalg.inv / np.linalg.det of a float32 matrix: numpy.linalg computes BOTH in float64 (its gufunc signature is 'd->d' for
// every real dtype) and casts the result back to float32 (`ainv.astype(result_t)`), so S^-1 is the float64 LAPACK inverse of the
// float32 S, rounded once -- for the CV model's diagonal S simply the correctly rounded 1 / S00.  (Rounds 1-2 ran the LU in
// float32: identical for a diagonal S, off by ulps for a dense one -- found with the dense-R cases of tests/golden/g15_single.npz.)
// The float64 operation order below is dgetf2's / dgetrs's (pivot = first maximum, reciprocal pivot scaling); at float64 precision
// it cannot change the float32 result except on a ~1e-8 tie.
struct LU2 {
    double u00, u01, u11, l10;
    bool swapped;
};
MHT_HD LU2 lu2(const float* s) {
    LU2 f;
    double a = s[0], b = s[1], c = s[2], d = s[3];
    f.swapped = fabs(c) > fabs(a);
    if (f.swapped) { double t = a; a = c; c = t; t = b; b = d; d = t; }
    f.l10 = c * (1.0 / a);
    f.u00 = a;
    f.u01 = b;
    f.u11 = fma(-f.l10, b, d);
    return f;
}

// np.linalg.inv on one 2x2 (dgesv with the identity as right-hand side, result cast to float32)
MHT_HD void inv2(const float* s, float* out) {
    const LU2 f = lu2(s);
    const double r00 = 1.0 / f.u00, r11 = 1.0 / f.u11;
#pragma unroll
    for (int col = 0; col < 2; ++col) {
        double y0 = (col == 0) ? 1.0 : 0.0, y1 = (col == 1) ? 1.0 : 0.0;
        if (f.swapped) { const double t = y0; y0 = y1; y1 = t; }
        y1 = fma(-f.l10, y0, y1);
        const double x1 = y1 * r11;
        const double x0 = fma(-f.u01, x1, y0) * r00;
        out[0 + col] = (float)x0;
        out[2 + col] = (float)x1;
    }
}

// The measurement-independent covariance chain of one hypothesis (kalman.py:62, :90-93): P -> P_bar, S, S^-1, K, P_hat.
// All float32, like the reference's model matrices; `with_phat` = false stops after K (the gains of a node whose children's
// covariance is not needed yet).
struct CovChain {
    float P_bar[NP];
    float P_hat[NP];
    float K[NK];     // NX x 2
    float S[4];
    float S_inv[4];
};
MHT_HD void cov_chain(const Model& m, const float* P, CovChain& o, bool with_phat = true) {
    // kalman.py:62  P_bar = matmul(matmul(A, P), A.T) + Q
    float AP[NP], At[NP], APA[NP];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) At[i * NX + j] = m.A[j * NX + i];
    gemm_chain<float, float, float, NX, NX, NX>(m.A, P, AP);
    gemm_chain<float, float, float, NX, NX, NX>(AP, At, APA);
#pragma unroll
    for (int i = 0; i < NP; ++i) o.P_bar[i] = APA[i] + m.Q[i];
    // kalman.py:90  S = matmul(matmul(C, P_bar), C.T) + R
    float Ct[NK], CP[NK], CPC[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NX; ++j) Ct[j * 2 + i] = m.C[i * NX + j];
    gemm_chain<float, float, float, 2, NX, NX>(m.C, o.P_bar, CP);
    gemm_chain<float, float, float, 2, NX, 2>(CP, Ct, CPC);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.S[i] = CPC[i] + m.R[i];
    // kalman.py:91  S_inv = np.linalg.inv(S)
    inv2(o.S, o.S_inv);
    // kalman.py:92  K = matmul(matmul(P_bar, C.T), S_inv)
    float PCt[NK];
    gemm_chain<float, float, float, NX, NX, 2>(o.P_bar, Ct, PCt);
    gemm_chain<float, float, float, NX, 2, 2>(PCt, o.S_inv, o.K);
    if (!with_phat) return;
    // kalman.py:93  P_hat = P_bar - matmul(K.dot(C), P_bar)
    float KC[NP], KCP[NP];
    gemm_chain<float, float, float, NX, 2, NX>(o.K, m.C, KC);
    gemm_chain<float, float, float, NX, NX, NX>(KC, o.P_bar, KCP);
#pragma unroll
    for (int i = 0; i < NP; ++i) o.P_hat[i] = o.P_bar[i] - KCP[i];
}

// kalman.py:61 / :89  x_bar = A.dot(x.T).T (A promoted to the state dtype), z_hat = C.dot(x_bar.T).T
template <typename TS>
MHT_HD void state_predict(const Model& m, const TS* x, TS* x_bar, TS* z_hat) {
    gemm_chain<TS, float, TS, NX, NX, 1>(m.A, x, x_bar);
    gemm_chain<TS, float, TS, 2, NX, 1>(m.C, x_bar, z_hat);
}

// ---- ONE leaf / ONE hit: NumPy hands a matrix times a single column to BLAS gemv, not gemm --------------------------------------
// `A.dot(x_0_list.T)` (kalman.py:60, :88) with ONE leaf in the call -- a target whose tree is a single hypothesis: every target in
// its first scan, a target N-scan pruning or similar-state pruning cut down to one leaf -- is (4,4) x (4,1): numpy's
// cblas_matrixproduct calls ?gemv for a matrix times a column, and OpenBLAS's gemv kernels (Haswell / SkylakeX / Zen, the same
// code) do not accumulate a row in one FMA chain: every product is rounded on its own and the four are added pairwise, float64
// (p0 + p2) + (p1 + p3), float32 (p0 + p1) + (p2 + p3); six terms (float64): the first four as before, + fma(a4, x4, a5 * x5).
// Probed on the development host and on the GPU box's host (tools/probe/blas_order_probe.py); pinned by tests/golden/g15_single.npz,
// recorded from the reference's kalman module.  For the CV model (rows like [1, 0, T, 0]) this is x + round(T v) instead of
// fma(T, v, x): a last-bit difference in 3.5 % of the single-leaf predictions.
template <typename TS, int K_>
MHT_HD TS gemv_row(const float* a, const TS* x) {
    static_assert(K_ == 2 || K_ == 4 || K_ == 6, "gemv_row: 2, 4 or 6 terms");
    if (K_ == 2) {            // np.matmul(K_row, z_tilde.T) with one gated measurement (kalman.py:50): float64 fma(a0, x0, a1 * x1), float32 p0 + p1
        if (sizeof(TS) == 8) return fmaT((TS)a[0], x[0], (TS)a[1] * x[1]);
        return (TS)a[0] * x[0] + (TS)a[1] * x[1];
    }
    const TS p0 = (TS)a[0] * x[0], p1 = (TS)a[1] * x[1], p2 = (TS)a[2] * x[2], p3 = (TS)a[3] * x[3];
    if (K_ == 4) return (sizeof(TS) == 8) ? (p0 + p2) + (p1 + p3) : (p0 + p1) + (p2 + p3);
    // six terms.  float64: probed (all rows).  float32: the rows OpenBLAS's sgemv_t hands to its scalar tail loop add the six products
    // one after the other; the vector kernel's rows are NOT reproduced (no float32 six-state chain exists: the initiator is 4-state)
    if (sizeof(TS) == 8) return ((p0 + p2) + (p1 + p3)) + fmaT((TS)a[4], x[4], (TS)a[5] * x[5]);
    return ((((p0 + p1) + p2) + p3) + (TS)a[4] * x[4]) + (TS)a[5] * x[5];
}
// x_bar, z_hat of the ONLY leaf of a call (see above)
template <typename TS>
MHT_HD void state_predict_single(const Model& m, const TS* x, TS* x_bar, TS* z_hat) {
#pragma unroll
    for (int i = 0; i < NX; ++i) x_bar[i] = gemv_row<TS, NX>(m.A + i * NX, x);
#pragma unroll
    for (int i = 0; i < 2; ++i) z_hat[i] = gemv_row<TS, NX>(m.C + i * NX, x_bar);
}

template <typename TS>
MHT_HD void predict_precalc(const Model& m, const TS* x, const float* P, Predicted<TS>& o, bool single = false) {
    if (single) state_predict_single<TS>(m, x, o.x_bar, o.z_hat);
    else state_predict<TS>(m, x, o.x_bar, o.z_hat);
    CovChain c;
    cov_chain(m, P, c);
#pragma unroll
    for (int i = 0; i < NP; ++i) { o.P_bar[i] = c.P_bar[i]; o.P_hat[i] = c.P_hat[i]; }
#pragma unroll
    for (int i = 0; i < NK; ++i) o.K[i] = c.K[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { o.S[i] = c.S[i]; o.S_inv[i] = c.S_inv[i]; }
}

// kalman.py:19  ln( lambda_ex * sqrt(det(2 pi S)) / P_d ), evaluated in f32 exactly as NumPy evaluates it for an
// f32 S: python floats are weak scalars (f32 arithmetic), det = sign * u00 * u11 of the pivoted LU factors (in float64, cast to
// float32: numpy.linalg works in double), sqrt and the divide are correctly rounded.  The final f32 log is NumPy's own SIMD polynomial, which is not
// correctly rounded; here it is a double-precision log rounded to f32 (= correctly rounded), which differs
// from NumPy's by at most 1 ulp(f32) (measured: 5 % of inputs, |delta| <= 4.8e-7).  DESIGN.md "NLLR tolerance".
MHT_HD float nllr_const(const float* S, double lambda_ex, double P_d) {
    const float two_pi = (float)(2.0 * 3.141592653589793);
    float s2[4] = {S[0] * two_pi, S[1] * two_pi, S[2] * two_pi, S[3] * two_pi};
    const LU2 f = lu2(s2);                       // np.linalg.det: float64 LU of the float32 matrix, cast back to float32
    double det64 = f.u00 * f.u11;
    if (f.swapped) det64 = -det64;
    const float det = (float)det64;
    float r = sqrtf(det);
    r = (float)lambda_ex * r;
    r = r / (float)P_d;
    return (float)log((double)r);
}

// kalman.py:36-40 + :25-28 + tracker.py:829 for one (leaf, measurement) pair.
// (TG: float32 gains, or the float64 ones of a target NumPy has promoted -- mht_la64.h)
template <typename TS, typename TG = float>
MHT_HD bool gate_pair(const TS* z_hat, const TG* S_inv, float zx, float zy, TS eta2, TS* zt, TS& nis) {
    zt[0] = (TS)zx - z_hat[0];
    zt[1] = (TS)zy - z_hat[1];
    TS t0 = fmaT(zt[1], (TS)S_inv[2], zt[0] * (TS)S_inv[0]);
    TS t1 = fmaT(zt[1], (TS)S_inv[3], zt[0] * (TS)S_inv[1]);
    nis = t0 * zt[0] + t1 * zt[1];
    return nis <= eta2;
}

// kalman.py:43-52  x_hat = x_bar + K z_tilde   (one component / all four)
template <typename TS>
MHT_HD TS update_component(TS x_bar_i, float k0, float k1, const TS* zt) {
    TS acc = (TS)k0 * zt[0];
    acc = fmaT((TS)k1, zt[1], acc);
    return x_bar_i + acc;
}
// the same for a leaf with exactly ONE gated measurement: np.matmul(K, z_tilde.T) is then matrix x column = gemv (gemv_row<TS, 2>)
template <typename TS>
MHT_HD TS update_component_single(TS x_bar_i, float k0, float k1, const TS* zt) {
    const float kk[2] = {k0, k1};
    return x_bar_i + gemv_row<TS, 2>(kk, zt);
}
// either order from one multiply and one FMA (the kernels' emission is at the edge of its register budget: no second code path).
// gemm: fma(k1, z1, k0 * z0).  gemv, float64: fma(k0, z0, k1 * z1); float32: k0 * z0 + k1 * z1 = fma(1, k0 * z0, k1 * z1) -- exact product
// of 1 and an already rounded number, so the FMA rounds once, like the addition.
template <typename TS, typename TG = float>
MHT_HD TS update_component_n(TS x_bar_i, TG k0, TG k1, const TS* zt, bool single_hit) {
    const TS p = single_hit ? (TS)k1 * zt[1] : (TS)k0 * zt[0];
    TS a = single_hit ? (TS)k0 : (TS)k1, b = single_hit ? zt[0] : zt[1];
    if (sizeof(TS) == 4 && single_hit) { a = (TS)1; b = (TS)k0 * zt[0]; }
    return x_bar_i + fmaT(a, b, p);
}
template <typename TS>
MHT_HD void update_state(const TS* x_bar, const float* K, const TS* zt, TS* x_hat, bool single_hit = false) {
#pragma unroll
    for (int i = 0; i < NX; ++i) x_hat[i] = update_component_n<TS>(x_bar[i], K[i * 2], K[i * 2 + 1], zt, single_hit);
}

// ---- dimension-generic restatement (BASELINE config 5 names a 6-state model; the reference's kalman module is dimension-generic:
//      kalman.py:55-101).  Same evaluation order as the 4-state code above, NX states, 2 measurements.
template <int NXX>
struct ModelX {
    float A[NXX * NXX], Q[NXX * NXX], C[2 * NXX], R[4];
    double eta2, lambda_ex;
    int ct; double T;      // ct = 1: constant-turn transition, A rebuilt per leaf from its turn rate (ct_phi below; mht_model_x::transition)
};
// pymht_amd/models/ct.py::Phi(T, w), six states [x, y, vx, vy, w, a]: computed in float64, rounded to float32 like the model's matrices
MHT_HD void ct_phi(double T, double w, float* A) {
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = (i % 7 == 0) ? 1.0f : 0.0f;
    const double s = sin(w * T), c = cos(w * T);
    double sw, cw;
    if (fabs(w) < 1e-9) { sw = T; cw = 0.0; } else { sw = s / w; cw = (1.0 - c) / w; }
    A[0 * 6 + 2] = (float)sw; A[0 * 6 + 3] = (float)(-cw);
    A[1 * 6 + 2] = (float)cw; A[1 * 6 + 3] = (float)sw;
    A[2 * 6 + 2] = (float)c;  A[2 * 6 + 3] = (float)(-s);
    A[3 * 6 + 2] = (float)s;  A[3 * 6 + 3] = (float)c;
    A[4 * 6 + 5] = (float)T;
}
template <typename TS, int NXX>
MHT_HD void predict_precalc_x(const ModelX<NXX>& m, const TS* x, const float* P, TS* x_bar, TS* z_hat, float* P_bar, float* P_hat,
                              float* K, float* S, float* S_inv, bool single = false) {
    if (single) {             // one leaf in the call: gemv (gemv_row above)
#pragma unroll
        for (int i = 0; i < NXX; ++i) x_bar[i] = gemv_row<TS, NXX>(m.A + i * NXX, x);
#pragma unroll
        for (int i = 0; i < 2; ++i) z_hat[i] = gemv_row<TS, NXX>(m.C + i * NXX, x_bar);
    } else {
        gemm_chain<TS, float, TS, NXX, NXX, 1>(m.A, x, x_bar);              // kalman.py:61
        gemm_chain<TS, float, TS, 2, NXX, 1>(m.C, x_bar, z_hat);           // kalman.py:89
    }
    float AP[NXX * NXX], At[NXX * NXX], APA[NXX * NXX];
#pragma unroll
    for (int i = 0; i < NXX; ++i)
#pragma unroll
        for (int j = 0; j < NXX; ++j) At[i * NXX + j] = m.A[j * NXX + i];
    gemm_chain<float, float, float, NXX, NXX, NXX>(m.A, P, AP);          // kalman.py:62
    gemm_chain<float, float, float, NXX, NXX, NXX>(AP, At, APA);
#pragma unroll
    for (int i = 0; i < NXX * NXX; ++i) P_bar[i] = APA[i] + m.Q[i];
    float Ct[2 * NXX], CP[2 * NXX], CPC[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NXX; ++j) Ct[j * 2 + i] = m.C[i * NXX + j];
    gemm_chain<float, float, float, 2, NXX, NXX>(m.C, P_bar, CP);       // kalman.py:90
    gemm_chain<float, float, float, 2, NXX, 2>(CP, Ct, CPC);
#pragma unroll
    for (int i = 0; i < 4; ++i) S[i] = CPC[i] + m.R[i];
    inv2(S, S_inv);                                                   // kalman.py:91
    float PCt[2 * NXX];
    gemm_chain<float, float, float, NXX, NXX, 2>(P_bar, Ct, PCt);       // kalman.py:92
    gemm_chain<float, float, float, NXX, 2, 2>(PCt, S_inv, K);
    float KC[NXX * NXX], KCP[NXX * NXX];
    gemm_chain<float, float, float, NXX, 2, NXX>(K, m.C, KC);           // kalman.py:93
    gemm_chain<float, float, float, NXX, NXX, NXX>(KC, P_bar, KCP);
#pragma unroll
    for (int i = 0; i < NXX * NXX; ++i) P_hat[i] = P_bar[i] - KCP[i];
}

// np.add.reduce of a contiguous 1-D array, streamed: element i of n (pairwise_sum of NumPy's loops: n < 8 sequential; else eight
// running sums over the blocks of eight, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 leftovers one by one.
// Exact for n <= 128, NumPy's block size; beyond that NumPy recurses -- no radar scan puts 128 plots within 4 m of one prediction)
template <typename T> struct Sum1D {
    T r[8]; T res; int n, blocked;
    MHT_HD void begin(int n_) { n = n_; blocked = n_ - (n_ % 8); res = (T)0; }
    MHT_HD void add(int i, T v) {
        if (n < 8) { res = (i == 0) ? v : res + v; return; }
        if (i < 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j == i) r[j] = v;
        } else if (i < blocked) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j == (i & 7)) r[j] = r[j] + v;
        }
        if (i == blocked - 1) res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        if (i >= blocked) res = res + v;
    }
};

// Node flags (one byte per hypothesis)
enum : uint8_t {
    F_STATE_F32 = 1,   // state chain (x, z_hat, z_tilde, NIS, NLLR) is float32: tracks born from the initiator
    F_SCORE_F32 = 2,   // cumulativeNLLR currently holds a float32 value (all-hit path from an int-0 root)
    F_SCORE_INT0 = 4,  // cumulativeNLLR is the Python int 0 of a fresh root (weak scalar)
    F_DEAD = 8,        // taken out of the tree by similar-state pruning (mht_similar.hip): the slot stays, the hypothesis is gone
    F_COV_F64 = 16     // the covariance is float64 (AIS forests: an AIS-updated node, or a child of a batch NumPy promoted because one of its
                       // members was -- models/ais.py:4, tracker.py:859-870): the node's key names a float64 value (mht_vtab.h)
};

}  // namespace mht
