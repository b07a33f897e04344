// mht_gate_scan_x: the gate / update / score seam for a DIMENSION-GENERIC linear-Gaussian model (NX states, 2 measurements).
// BASELINE config 5 names a 6-state model; the reference has none, but its kalman module is dimension-generic
// (pymht/utils/kalman.py:55-101: predict, precalc; :36-52: z_tilde, numpyFilter; :14-28: NLLR, NIS; gate tracker.py:829), so this
// seam is what a 6-state model would run through -- pinned by known-answer vectors made with the reference's own functions
// (tests/golden/g11_kalman6.npz).  The forest (mht_fgrow.hip) is 4-state; this is the stateless, caller-owned-buffers form only.
//
// Three small launches, HBM bound like the 4-state kernel: nothing is reshaped into a GEMM, the matrices are NX x NX.
//   gatex_leaf_kernel    thread = leaf: predict + precalc in registers (arithmetic of mht_math.h::predict_precalc_x, the reference's
//                        evaluation order), x_bar / P_bar / P_hat / S / S^-1 / K / z_hat / score constant stored SoA (coalesced);
//   gatex_count_kernel   wavefront = leaf, lanes = measurements: exact NIS, hit bits by ballot into a [L][W] mask, hits per leaf;
//   (exclusive scan of the counts: one workgroup)
//   gatex_emit_kernel    wavefront = leaf: k-th set bit -> slot row_ptr[l] + k: measurement index (ascending), x_hat = x_bar + K z_tilde,
//                        NLLR = 0.5 NIS + ln(lambda_ex sqrt(det 2 pi S) / P_d).
#include "mht_kernels.h"

namespace mht {

template <int NX>
struct GateXArgs {
    ModelX<NX> model;
    int L, M, W, cap;
    const double* x; const uint8_t* flags; const float* P; const double* pd; const float* z;
    double* x_bar; float* P_bar; float* P_hat; float* S; float* S_inv; float* K;
    double* z_hat; float* lnc;                       // scratch: [2][L], [L]
    unsigned long long* mask; int32_t* cnt;          // scratch: [L][W], [L]
    int32_t* row_ptr; int32_t* col_idx; double* x_hat; double* nllr;
    DevStatus* status;
};

template <typename TS, int NX>
__device__ __forceinline__ void gatex_leaf(const GateXArgs<NX>& a, int l) {
    const size_t L = a.L;
    TS xs[NX], xb[NX], zh[2];
    float P[NX * NX], Pb[NX * NX], Ph[NX * NX], K[2 * NX], S[4], Si[4];
#pragma unroll
    for (int k = 0; k < NX; ++k) xs[k] = (TS)a.x[(size_t)k * L + l];
#pragma unroll
    for (int e = 0; e < NX * NX; ++e) P[e] = a.P[(size_t)e * L + l];
    bool done = false;
    if constexpr (NX == 6) { if (a.model.ct) {
        done = true;
        // a state-dependent transition: the leaf's own A, and the reference's per-hypothesis form (kalman.predict_single + kalman.precalc
        // on a batch of one: the matrix x vector products in gemv order)
        ModelX<6> m;
        for (int i = 0; i < 36; ++i) m.Q[i] = a.model.Q[i];
        for (int i = 0; i < 12; ++i) m.C[i] = a.model.C[i];
        for (int i = 0; i < 4; ++i) m.R[i] = a.model.R[i];
        m.eta2 = a.model.eta2; m.lambda_ex = a.model.lambda_ex; m.ct = 1; m.T = a.model.T;
        ct_phi(a.model.T, (double)xs[4 < NX ? 4 : 0], m.A);
        predict_precalc_x<TS, 6>(m, reinterpret_cast<const TS*>(xs), P, reinterpret_cast<TS*>(xb), zh, Pb, Ph, K, S, Si, true);
    } }
    if (!done) predict_precalc_x<TS, NX>(a.model, xs, P, xb, zh, Pb, Ph, K, S, Si, a.L == 1);      // (one leaf in the call: gemv order)
#pragma unroll
    for (int k = 0; k < NX; ++k) a.x_bar[(size_t)k * L + l] = (double)xb[k];
#pragma unroll
    for (int e = 0; e < NX * NX; ++e) { a.P_bar[(size_t)e * L + l] = Pb[e]; a.P_hat[(size_t)e * L + l] = Ph[e]; }
#pragma unroll
    for (int e = 0; e < 2 * NX; ++e) a.K[(size_t)e * L + l] = K[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a.S[(size_t)e * L + l] = S[e]; a.S_inv[(size_t)e * L + l] = Si[e]; }
    a.z_hat[l] = (double)zh[0];
    a.z_hat[L + l] = (double)zh[1];
    a.lnc[l] = nllr_const(S, a.model.lambda_ex, a.pd[l]);
}
template <int NX>
__global__ __launch_bounds__(128) void gatex_leaf_kernel(const GateXArgs<NX> a) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.L) return;
    if (a.flags[l] & F_STATE_F32) gatex_leaf<float, NX>(a, l);
    else gatex_leaf<double, NX>(a, l);
}

// (leaf, measurement) test in the state dtype of the leaf; returns the hit flag, z_tilde and NIS
template <int NX>
__device__ __forceinline__ bool gatex_pair(const GateXArgs<NX>& a, int l, bool f32, float zx, float zy, double* zt, double& nis) {
    const size_t L = a.L;
    const float Si[4] = {a.S_inv[l], a.S_inv[L + l], a.S_inv[2 * L + l], a.S_inv[3 * L + l]};
    if (f32) {
        float zh[2] = {(float)a.z_hat[l], (float)a.z_hat[L + l]}, t[2], n;
        const bool hit = gate_pair<float>(zh, Si, zx, zy, (float)a.model.eta2, t, n);
        zt[0] = t[0]; zt[1] = t[1]; nis = n;
        return hit;
    }
    double zh[2] = {a.z_hat[l], a.z_hat[L + l]};
    return gate_pair<double>(zh, Si, zx, zy, a.model.eta2, zt, nis);
}

template <int NX>
__global__ __launch_bounds__(256) void gatex_count_kernel(const GateXArgs<NX> a) {
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (l >= a.L) return;
    const bool f32 = (a.flags[l] & F_STATE_F32) != 0;
    int hits = 0;
    for (int w = 0; w < a.W; ++w) {
        const int j = w * 64 + lane;
        bool hit = false;
        if (j < a.M) {
            double zt[2], nis;
            hit = gatex_pair<NX>(a, l, f32, a.z[2 * j], a.z[2 * j + 1], zt, nis);
        }
        const unsigned long long bits = __ballot(hit);
        if (lane == 0) a.mask[(size_t)l * a.W + w] = bits;
        hits += __popcll(bits);
    }
    if (lane == 0) a.cnt[l] = hits;
}

// exclusive scan of cnt[0..L) -> row_ptr[0..L]; one workgroup of 1024 threads, chunks of 1024
__global__ __launch_bounds__(1024) void gatex_scan_kernel(const int32_t* cnt, int32_t* row_ptr, int L, int cap, DevStatus* status) {
    __shared__ int s_w[16], s_run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < L; base += 1024) {
        const int i = base + tid;
        const int v = i < L ? cnt[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int off = s_run;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (i < L) row_ptr[i] = off + incl - v;
        __syncthreads();
        if (tid == 1023) s_run = off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        row_ptr[L] = s_run;
        status->n_children = s_run;
        if (s_run > cap) status->overflow = 1;
    }
}

template <int NX>
__global__ __launch_bounds__(256) void gatex_emit_kernel(const GateXArgs<NX> a) {
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (l >= a.L) return;
    const size_t L = a.L;
    const bool f32 = (a.flags[l] & F_STATE_F32) != 0;
    int base = a.row_ptr[l];
    const bool one = a.row_ptr[l + 1] - base == 1;      // one gated measurement: np.matmul(K, z_tilde.T) is matrix x column = gemv
    double xb[NX];
    float K[2 * NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) xb[k] = a.x_bar[(size_t)k * L + l];
#pragma unroll
    for (int e = 0; e < 2 * NX; ++e) K[e] = a.K[(size_t)e * L + l];
    const float lnc = a.lnc[l];
    for (int w = 0; w < a.W; ++w) {
        const unsigned long long bits = a.mask[(size_t)l * a.W + w];
        if ((bits >> lane) & 1ull) {
            const int j = w * 64 + lane;
            const int c = base + __popcll(bits & ((1ull << lane) - 1ull));
            if (c < a.cap) {
                double zt[2], nis;
                gatex_pair<NX>(a, l, f32, a.z[2 * j], a.z[2 * j + 1], zt, nis);
                a.col_idx[c] = j;
                if (f32) {
                    const float ztf[2] = {(float)zt[0], (float)zt[1]};
#pragma unroll
                    for (int k = 0; k < NX; ++k) a.x_hat[(size_t)k * a.cap + c] = (double)update_component_n<float>((float)xb[k], K[2 * k], K[2 * k + 1], ztf, one);
                    a.nllr[c] = (double)(0.5f * (float)nis + lnc);                 // kalman.py:19 (float32 chain)
                } else {
#pragma unroll
                    for (int k = 0; k < NX; ++k) a.x_hat[(size_t)k * a.cap + c] = update_component_n<double>(xb[k], K[2 * k], K[2 * k + 1], zt, one);
                    a.nllr[c] = 0.5 * nis + (double)lnc;
                }
            }
        }
        base += __popcll(bits);
    }
}

template <int NX>
static int run_gate_x(mht_ctx* ctx, const mht_model_x* m, int32_t L, const double* x, const uint8_t* flags, const float* P, const double* pd,
                      const float* z, int32_t M, double* x_bar, float* P_bar, float* P_hat, float* S, float* S_inv, float* K,
                      int32_t* row_ptr, int32_t* col_idx, double* x_hat, double* nllr, int32_t cap, int32_t* n_pairs) {
    GateXArgs<NX> a = {};
    for (int i = 0; i < NX * NX; ++i) { a.model.A[i] = m->A ? m->A[i] : 0.f; a.model.Q[i] = m->Q[i]; }
    for (int i = 0; i < 2 * NX; ++i) a.model.C[i] = m->C[i];
    for (int i = 0; i < 4; ++i) a.model.R[i] = m->R[i];
    a.model.eta2 = m->eta2; a.model.lambda_ex = m->lambda_ex;
    a.model.ct = m->transition == 1 ? 1 : 0; a.model.T = m->period;
    a.L = L; a.M = M; a.W = (M + 63) / 64; a.cap = cap;
    a.x = x; a.flags = flags; a.P = P; a.pd = pd; a.z = z;
    a.x_bar = x_bar; a.P_bar = P_bar; a.P_hat = P_hat; a.S = S; a.S_inv = S_inv; a.K = K;
    a.row_ptr = row_ptr; a.col_idx = col_idx; a.x_hat = x_hat; a.nllr = nllr;
    const int Wn = a.W > 0 ? a.W : 1;
    // scratch: z_hat [2][L] f64 | mask [L][W] u64 | lnc [L] f32 | cnt [L] i32
    const size_t bytes = (size_t)2 * L * 8 + (size_t)L * Wn * 8 + (size_t)L * 4 + (size_t)L * 4 + 64;
    { const int rc = ctx->hitmask.ensure(bytes); if (rc) return rc; }
    char* q = static_cast<char*>(ctx->hitmask.ptr);
    a.z_hat = reinterpret_cast<double*>(q); q += (size_t)2 * L * 8;
    a.mask = reinterpret_cast<unsigned long long*>(q); q += (size_t)L * Wn * 8;
    a.lnc = reinterpret_cast<float*>(q); q += (size_t)L * 4;
    a.cnt = reinterpret_cast<int32_t*>(q);
    a.status = ctx->status;
    MHT_HIP_CHECK(hipMemsetAsync(ctx->status, 0, sizeof(DevStatus), ctx->stream));
    if (L > 0) {
        hipLaunchKernelGGL(gatex_leaf_kernel<NX>, dim3((L + 127) / 128), dim3(128), 0, ctx->stream, a);
        hipLaunchKernelGGL(gatex_count_kernel<NX>, dim3((L + 3) / 4), dim3(256), 0, ctx->stream, a);
    }
    hipLaunchKernelGGL(gatex_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, a.cnt, row_ptr, L, cap, ctx->status);
    if (L > 0) hipLaunchKernelGGL(gatex_emit_kernel<NX>, dim3((L + 3) / 4), dim3(256), 0, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    DevStatus st;
    MHT_HIP_CHECK(hipMemcpyAsync(&st, ctx->status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (n_pairs) *n_pairs = st.n_children;
    if (st.overflow) {
        set_error("mht_gate_scan_x: %d gated pairs exceed the capacity of the output arrays (%d)", st.n_children, cap);
        return MHT_E_CAPACITY;
    }
    return MHT_OK;
}

}  // namespace mht

using namespace mht;

extern "C" int mht_gate_scan_x(mht_ctx* ctx, const mht_model_x* model, int32_t L, const double* x, const uint8_t* flags, const float* P,
                               const double* pd, const float* z, int32_t M, double* x_bar, float* P_bar, float* P_hat, float* S,
                               float* S_inv, float* K, int32_t* row_ptr, int32_t* col_idx, double* x_hat, double* nllr, int32_t cap,
                               int32_t* n_pairs) {
    MHT_REQUIRE(ctx && model && (model->A || model->transition == 1) && model->Q && model->C && model->R, "mht_gate_scan_x: null argument");
    MHT_REQUIRE(model->nx == 4 || model->nx == 6, "mht_gate_scan_x: nx must be 4 or 6 (got %d)", model->nx);
    MHT_REQUIRE(model->transition == 0 || (model->transition == 1 && model->nx == 6), "mht_gate_scan_x: transition %d needs the six-state constant-turn layout", model->transition);
    MHT_REQUIRE(L >= 0 && M >= 0 && cap >= 0, "mht_gate_scan_x: negative size");
    MHT_REQUIRE(row_ptr && (L == 0 || (x && flags && P && pd && x_bar && P_bar && P_hat && S && S_inv && K)), "mht_gate_scan_x: null array");
    MHT_REQUIRE(M == 0 || z, "mht_gate_scan_x: z is null");
    MHT_REQUIRE(cap == 0 || (col_idx && x_hat && nllr), "mht_gate_scan_x: null output array");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    if (model->nx == 4)
        return run_gate_x<4>(ctx, model, L, x, flags, P, pd, z, M, x_bar, P_bar, P_hat, S, S_inv, K, row_ptr, col_idx, x_hat, nllr, cap, n_pairs);
    return run_gate_x<6>(ctx, model, L, x, flags, P, pd, z, M, x_bar, P_bar, P_hat, S, S_inv, K, row_ptr, col_idx, x_hat, nllr, cap, n_pairs);
}
