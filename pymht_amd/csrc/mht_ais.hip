// AIS-aided children: Tracker.__fuseRadarAndAis (pymht/tracker.py:417-552) on the device.
//
// Stateless seam mht_fuse_ais: for L leaves (caller-owned SoA arrays, as mht_gate_scan_x takes them) and the AIS messages of a scan
// in the reference's group order (pymht_amd/ais.py::group_messages) the fused / pure-AIS children of every leaf, CSR by leaf, in
// the reference's order.  One thread per leaf: the work per leaf is a handful of 4x4 float64 products per message that gates with
// it (a message gates with the ~30 leaves of its own ship's track and nobody else), then one pass over the scan's radar measurements
// per gated message -- a few thousand leaves x a few gated messages per scan: latency of the longest leaf, not throughput, and it
// runs only on scans that carry AIS messages.  Two passes (count, exclusive scan, emit) so that a leaf's children are contiguous.
//
// In the forest the same per-leaf function feeds the grow kernel (mht_forest.hip: forest_ais_*; fgrow_kernel<..., AIS>).
#include "mht_kernels.h"
#include "mht_ais_math.h"

namespace mht {

struct AisSeamArgs {
    Model model;                                   // C, R, eta2, lambda_ex are read
    int L; const double* x; const uint8_t* flags; const float* P; const double* P64; const double* pd; const int32_t* own;      // P64 != null (mht_fuse_ais_f64): [L][16] doubles, float64 covariances where flags carries F_COV_F64
    const AisGroup* groups; int nG; const AisMsg* msgs;
    double eta2_ais, lambda_ais;
    const float* z; int M;
    int32_t* cnt; const int32_t* child_ptr; int cap;
    double* out_x; double* out_P; int32_t* out_radar; double* out_nllr; int32_t* out_msg;
};

struct CountOnly {
    __device__ __forceinline__ void operator()(const double*, const double*, int, double, int) const {}
};
struct SeamEmit {
    const AisSeamArgs* a; int c;
    __device__ __forceinline__ void operator()(const double* x, const double* P, int radar, double nllr, int msg) {
        if (c < a->cap) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a->out_x[(size_t)k * a->cap + c] = x[k];
#pragma unroll
            for (int e = 0; e < 16; ++e) a->out_P[(size_t)c * 16 + e] = P[e];
            a->out_radar[c] = radar;
            a->out_nllr[c] = nllr;
            a->out_msg[c] = msg;
        }
        ++c;
    }
};

template <typename EMIT>
__device__ __forceinline__ int seam_leaf(const AisSeamArgs& a, int l, EMIT& e) {
    const double pd = a.pd[l];
    const int own = a.own ? a.own[l] : 0;
    if (a.P64 && (a.flags[l] & F_COV_F64)) {      // a node the reference carries in float64 (state and covariance: a promoted target's)
        double Pd[16], xd[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) Pd[i] = a.P64[(size_t)l * 16 + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) xd[k] = a.x[(size_t)k * a.L + l];
        return ais_fuse_leaf<double>(a.model, a.groups, a.nG, a.msgs, xd, Pd, pd, own, a.eta2_ais, a.lambda_ais, a.z, a.M, e);
    }
    float P[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = a.P64 ? (float)a.P64[(size_t)l * 16 + i] : a.P[(size_t)l * 16 + i];
    if (a.flags[l] & F_STATE_F32) {
        float xs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[k] = (float)a.x[(size_t)k * a.L + l];
        return ais_fuse_leaf<float>(a.model, a.groups, a.nG, a.msgs, xs, P, pd, own, a.eta2_ais, a.lambda_ais, a.z, a.M, e);
    }
    double xd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xd[k] = a.x[(size_t)k * a.L + l];
    return ais_fuse_leaf<double>(a.model, a.groups, a.nG, a.msgs, xd, P, pd, own, a.eta2_ais, a.lambda_ais, a.z, a.M, e);
}

__global__ __launch_bounds__(64) void ais_count_kernel(const AisSeamArgs a) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= a.L) return;
    CountOnly e;
    a.cnt[l] = seam_leaf(a, l, e);
}
__global__ __launch_bounds__(64) void ais_emit_kernel(const AisSeamArgs a) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= a.L) return;
    SeamEmit e{&a, a.child_ptr[l]};
    seam_leaf(a, l, e);
}
// exclusive scan of cnt[0..L) -> ptr[0..L]; one workgroup
__global__ __launch_bounds__(1024) void ais_scan_kernel(const int32_t* cnt, int32_t* ptr, int L, int cap, DevStatus* status) {
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
        if (i < L) ptr[i] = off + incl - v;
        __syncthreads();
        if (tid == 1023) s_run = off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        ptr[L] = s_run;
        status->n_children = s_run;
        if (s_run > cap) status->overflow = 1;
    }
}

// ---- the forest: fused children of every leaf of the newest layer, in front of the scan's grow launch ----------------------------
// One workgroup (a wavefront) per target slot of the COMMITTED table, lane = leaf.  Per leaf: count, take a slice of the record
// pool (one returning atomic), emit.  The covariance of the children of a (leaf, message) pair is a value like any other
// (mht_vtab.h) -- a FLOAT64 one, as the reference carries it (models/ais.py:4) -- found or inserted, and given a key of its own: a
// pseudo parent whose miss child it is, as a root's or a merged hypothesis' (mht_similar.hip), with the float64 gains the children
// need as leaves of the next scan.
struct ForestEmit {
    const AisForestArgs* a; int c; double pd; int last_msg, key;
    __device__ __forceinline__ void operator()(const double* x, const double* P, int radar, double nllr, int msg) {
        if (msg != last_msg) {      // (the children of one (leaf, message) pair share their covariance: float64, as the reference carries it)
            last_msg = msg;
            double P64[NP];
#pragma unroll
            for (int e = 0; e < NP; ++e) P64[e] = P[e < 16 ? e : 0];
            const int id0 = vt_find_or_insert64(a->vt, P64, pd);
            double row[GKF];
            vt_gains64(a->model, P64, pd, row);
            key = vt_pseudo_key64(a->vt, id0, row);
        }
        AisRec r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.x[k] = x[k];
        r.nllr = nllr; r.radar = radar; r.msg = msg; r.key = key; r.mmsi = a->msgs[msg].mmsi;
        a->rec[c++] = r;
    }
};

template <typename TP, typename EMIT>
__device__ __forceinline__ int forest_leaf(const AisForestArgs& a, int src, uint8_t fl, const TP* P, double pd, int own, EMIT& e) {
    if (fl & F_STATE_F32) {
        float xs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) xs[k] = (float)a.x[(size_t)k * a.cap + src];
        return ais_fuse_leaf<float>(a.model, a.groups, a.nG, a.msgs, xs, P, pd, own, a.eta2_ais, a.lambda_ais, a.z, a.M, e);
    }
    double xd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xd[k] = a.x[(size_t)k * a.cap + src];
    return ais_fuse_leaf<double>(a.model, a.groups, a.nG, a.msgs, xd, P, pd, own, a.eta2_ais, a.lambda_ais, a.z, a.M, e);
}
// count, take a slice of the record pool, emit -- in the dtype of the leaf's OWN covariance (kalman.predict_single on node.P_0, tracker.py:449-450)
template <typename TP>
__device__ __forceinline__ void forest_leaf_run(const AisForestArgs& a, int src, uint8_t fl, const TP* P, int& n, int& o) {
    const double pd = a.pd[src];
    const int own = a.hmmsi[src];
    CountOnly ce;
    n = forest_leaf(a, src, fl, P, pd, own, ce);
    if (n > 0) {
        o = (int)atomicAdd(a.rec_count, (unsigned)n);
        if (o + n > a.rec_cap) { a.status->overflow = 1; n = 0; }
    }
    if (n > 0) {
        ForestEmit fe{&a, o, pd, -1, 0};
        forest_leaf(a, src, fl, P, pd, own, fe);
    }
}

__global__ __launch_bounds__(64) void forest_ais_kernel(const AisForestArgs a) {
#if MHT_NX == 4
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.nT_dev[0]) return;
    const int first = a.t_first[t], cnt = a.t_leaf_off[t + 1] - a.t_leaf_off[t];
    for (int i = lane; i < cnt; i += 64) {
        const int src = first + i;
        const uint8_t fl = a.flags[src];
        int n = 0, o = 0;
        if (!(fl & F_DEAD)) {
            if (fl & F_COV_F64) {
                double P[NP];
                vt_load64(a.vt, a.vt.child[a.cov[src]], P);
                forest_leaf_run(a, src, fl, P, n, o);
            } else {
                float P[NP];
                vt_load(a.vt, a.vt.child[a.cov[src]], P);
                forest_leaf_run(a, src, fl, P, n, o);
            }
        }
        a.nf[src] = n;
        a.off[src] = o;
    }
#endif
}

int launch_forest_ais(mht_ctx* ctx, const AisForestArgs& a, int n_targets_ub) {
    const int grid = n_targets_ub < 1 ? 1 : n_targets_ub;
    hipLaunchKernelGGL(forest_ais_kernel, dim3(grid), dim3(64), 0, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

}  // namespace mht

using namespace mht;

static int fuse_ais_impl(mht_ctx* ctx, const mht_model* model, int32_t L, const double* x, const uint8_t* flags, const float* P, const double* P64, const double* pd,
                         const int32_t* own, const mht_ais_group* groups, int32_t nG, const mht_ais_msg* msgs, int32_t nA, double eta2_ais,
                         double lambda_ais, const float* z, int32_t M, int32_t* child_ptr, double* out_x, double* out_P, int32_t* out_radar,
                         double* out_nllr, int32_t* out_msg, int32_t cap, int32_t* n_children) {
    MHT_REQUIRE(NX == 4, "mht_fuse_ais: AIS messages report four states (models/ais.py); this is the %d-state build of the library", NX);
    MHT_REQUIRE(ctx && model && child_ptr, "mht_fuse_ais: null argument");
    MHT_REQUIRE(L >= 0 && M >= 0 && cap >= 0 && nG >= 0 && nA >= 0, "mht_fuse_ais: negative size");
    MHT_REQUIRE(L == 0 || (x && flags && (P || P64) && pd), "mht_fuse_ais: null leaf array");
    MHT_REQUIRE(nG == 0 || (groups && msgs), "mht_fuse_ais: null message array");
    MHT_REQUIRE(M == 0 || z, "mht_fuse_ais: z is null");
    MHT_REQUIRE(cap == 0 || (out_x && out_P && out_radar && out_nllr && out_msg), "mht_fuse_ais: null output array");
    MHT_REQUIRE(lambda_ais > 0.0, "mht_fuse_ais: lambda_ais must be positive (a tracker without a finite radarRange has none: tracker.py:438)");
    for (int g = 0; g < nG; ++g)
        MHT_REQUIRE(groups[g].first >= 0 && groups[g].count >= 0 && groups[g].first + groups[g].count <= nA, "mht_fuse_ais: group %d outside the message list", g);
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    static_assert(sizeof(mht_ais_group) == sizeof(AisGroup) && sizeof(mht_ais_msg) == sizeof(AisMsg), "ABI structs");
    AisSeamArgs a = {};
#if MHT_NX == 4
    for (int i = 0; i < 8; ++i) a.model.C[i] = model->C[i];
    for (int i = 0; i < 4; ++i) a.model.R[i] = model->R[i];
#endif
    a.model.eta2 = model->eta2; a.model.lambda_ex = model->lambda_ex;
    a.L = L; a.x = x; a.flags = flags; a.P = P; a.P64 = P64; a.pd = pd; a.own = own;
    a.nG = nG; a.eta2_ais = eta2_ais; a.lambda_ais = lambda_ais; a.z = z; a.M = M; a.cap = cap;
    a.child_ptr = child_ptr; a.out_x = out_x; a.out_P = out_P; a.out_radar = out_radar; a.out_nllr = out_nllr; a.out_msg = out_msg;
    // scratch: groups | messages | cnt[L]
    const size_t gb = ((size_t)(nG > 0 ? nG : 1) * sizeof(AisGroup) + 15) & ~(size_t)15, mb = ((size_t)(nA > 0 ? nA : 1) * sizeof(AisMsg) + 15) & ~(size_t)15;
    { const int rc = ctx->hitmask.ensure(gb + mb + (size_t)(L + 1) * 4 + 64); if (rc) return rc; }
    char* q = static_cast<char*>(ctx->hitmask.ptr);
    if (nG) MHT_HIP_CHECK(hipMemcpyAsync(q, groups, (size_t)nG * sizeof(AisGroup), hipMemcpyHostToDevice, ctx->stream));
    if (nA) MHT_HIP_CHECK(hipMemcpyAsync(q + gb, msgs, (size_t)nA * sizeof(AisMsg), hipMemcpyHostToDevice, ctx->stream));
    a.groups = reinterpret_cast<const AisGroup*>(q);
    a.msgs = reinterpret_cast<const AisMsg*>(q + gb);
    a.cnt = reinterpret_cast<int32_t*>(q + gb + mb);
    MHT_HIP_CHECK(hipMemsetAsync(ctx->status, 0, sizeof(DevStatus), ctx->stream));
#if MHT_NX == 4
    if (L > 0) hipLaunchKernelGGL(ais_count_kernel, dim3((L + 63) / 64), dim3(64), 0, ctx->stream, a);
    hipLaunchKernelGGL(ais_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, a.cnt, child_ptr, L, cap, ctx->status);
    if (L > 0) hipLaunchKernelGGL(ais_emit_kernel, dim3((L + 63) / 64), dim3(64), 0, ctx->stream, a);
#endif
    MHT_HIP_CHECK(hipGetLastError());
    DevStatus st;
    MHT_HIP_CHECK(hipMemcpyAsync(&st, ctx->status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // (also: the host arrays behind the two copies may be reused now)
    if (n_children) *n_children = st.n_children;
    if (st.overflow) {
        set_error("mht_fuse_ais: %d children exceed the capacity of the output arrays (%d)", st.n_children, cap);
        return MHT_E_CAPACITY;
    }
    return MHT_OK;
}

// ---- constant-turn forest (six-state build, MHT_FOREST_CT; mht_kernels.h: CtGrow) -------------------------------------------------
// Per leaf of the newest layer, in front of the grow launch: the leaf's own transition Phi(T, w) from its turn rate, the reference's
// per-hypothesis form kalman.predict_single (kalman.py:67-70: A.dot(x), A.dot(P).dot(A.T) + Q) and kalman.precalc on a batch of ONE
// (kalman.py:82-101) -- the arithmetic of the stateless seam mht_gate_scan_x with mht_model_x.transition = 1 (mht_gatex.hip,
// tests/golden/g21_ct6.npz) -- leaving the prediction, the gains row and the children's covariances by leaf node.
namespace mht {
#if MHT_NX == 6
__device__ __forceinline__ void ct_node_cov(const CtForestArgs& a, int key, float* P) {
    const float* src = key < 0 ? a.Proot + (size_t)(-2 - key) * NP : ((key & 1) ? a.Phat_prev : a.Pbar_prev) + (size_t)(key >> 1) * NP;
    // (a covariance is 144 contiguous bytes at a multiple of 16: nine 16-byte loads per lane instead of 36 scattered words)
    const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int q = 0; q < NP / 4; ++q) { const float4 v = s4[q]; P[4 * q] = v.x; P[4 * q + 1] = v.y; P[4 * q + 2] = v.z; P[4 * q + 3] = v.w; }
}
__global__ __launch_bounds__(64) void forest_ct_kernel(const CtForestArgs a) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.nT_dev[0]) return;
    int first, cnt;
    if (a.fused) { first = a.p_firstsurv[t]; cnt = (a.p_status[t] == 0) ? a.p_count[t] : 0; }      // (the previous scan's commit has not run: its per-target results stand in)
    else { first = a.t_first[t]; cnt = a.t_leaf_off[t + 1] - a.t_leaf_off[t]; }
    for (int i = lane; i < cnt; i += 64) {
        const int nd = first + i;
        if (a.flags[nd] & F_DEAD) continue;
        double xs[6], xb[6], zh[2];
        float P[36], Pb[36], Ph[36], K[12], S[4], Si[4];
#pragma unroll
        for (int k = 0; k < 6; ++k) xs[k] = a.x[(size_t)k * a.cap + nd];
        ct_node_cov(a, a.cov[nd], P);
        ModelX<6> m;
#pragma unroll
        for (int e = 0; e < 36; ++e) m.Q[e] = a.model.Q[e];
#pragma unroll
        for (int e = 0; e < 12; ++e) m.C[e] = a.model.C[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) m.R[e] = a.model.R[e];
        m.eta2 = a.model.eta2; m.lambda_ex = a.model.lambda_ex; m.ct = 1; m.T = a.T;
        ct_phi(a.T, xs[4], m.A);
        predict_precalc_x<double, 6>(m, xs, P, xb, zh, Pb, Ph, K, S, Si, true);
#pragma unroll
        for (int k = 0; k < 6; ++k) a.xbar[(size_t)k * a.cap + nd] = xb[k];
        a.zhat[nd] = zh[0]; a.zhat[(size_t)a.cap + nd] = zh[1];
        {
            float4* pb4 = reinterpret_cast<float4*>(a.Pbar + (size_t)nd * 36);
            float4* ph4 = reinterpret_cast<float4*>(a.Phat + (size_t)nd * 36);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                pb4[q] = make_float4(Pb[4 * q], Pb[4 * q + 1], Pb[4 * q + 2], Pb[4 * q + 3]);
                ph4[q] = make_float4(Ph[4 * q], Ph[4 * q + 1], Ph[4 * q + 2], Ph[4 * q + 3]);
            }
        }
        float row[GKF];
#pragma unroll
        for (int e = 0; e < GKF; ++e) row[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) row[e] = Si[e];
#pragma unroll
        for (int e = 0; e < 12; ++e) row[4 + e] = K[e];
        row[GK_LNC] = nllr_const(S, a.model.lambda_ex, a.pd[nd]);
        row[GK_RX] = sqrtf((float)a.model.eta2 * fabsf(S[0]));
        row[GK_RY] = sqrtf((float)a.model.eta2 * fabsf(S[3]));
#pragma unroll
        for (int q = 0; q < GKQ; ++q) a.gains[(size_t)nd * GKQ + q] = make_float4(row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]);
    }
}
#endif
int launch_forest_ct(mht_ctx* ctx, const CtForestArgs& a, int n_targets_ub) {
#if MHT_NX == 6
    const int grid = n_targets_ub < 1 ? 1 : n_targets_ub;
    hipLaunchKernelGGL(forest_ct_kernel, dim3(grid), dim3(64), 0, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
#else
    (void)ctx; (void)a; (void)n_targets_ub;
    set_error("the constant-turn forest needs the six-state build of the library");
    return MHT_E_INVALID;
#endif
}
}  // namespace mht

extern "C" int mht_fuse_ais(mht_ctx* ctx, const mht_model* model, int32_t L, const double* x, const uint8_t* flags, const float* P, const double* pd,
                            const int32_t* own, const mht_ais_group* groups, int32_t nG, const mht_ais_msg* msgs, int32_t nA, double eta2_ais,
                            double lambda_ais, const float* z, int32_t M, int32_t* child_ptr, double* out_x, double* out_P, int32_t* out_radar,
                            double* out_nllr, int32_t* out_msg, int32_t cap, int32_t* n_children) {
    return fuse_ais_impl(ctx, model, L, x, flags, P, nullptr, pd, own, groups, nG, msgs, nA, eta2_ais, lambda_ais, z, M, child_ptr, out_x, out_P, out_radar, out_nllr,
                         out_msg, cap, n_children);
}
extern "C" int mht_fuse_ais_f64(mht_ctx* ctx, const mht_model* model, int32_t L, const double* x, const uint8_t* flags, const double* P, const double* pd,
                                const int32_t* own, const mht_ais_group* groups, int32_t nG, const mht_ais_msg* msgs, int32_t nA, double eta2_ais,
                                double lambda_ais, const float* z, int32_t M, int32_t* child_ptr, double* out_x, double* out_P, int32_t* out_radar,
                                double* out_nllr, int32_t* out_msg, int32_t cap, int32_t* n_children) {
    return fuse_ais_impl(ctx, model, L, x, flags, nullptr, P, pd, own, groups, nG, msgs, nA, eta2_ais, lambda_ais, z, M, child_ptr, out_x, out_P, out_radar, out_nllr,
                         out_msg, cap, n_children);
}
