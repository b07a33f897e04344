// Gate / update / score kernels: the L x M fan-out of every leaf hypothesis against every measurement
// of a scan (reference: pymht/tracker.py:804-859 + pymht/utils/kalman.py, children of pyTarget.py:227-258).
//
// Two launches per scan:
//   gate_count_kernel  workgroup = 4 wavefronts, tile of 16 leaves.  Lanes 0..15 run predict+precalc
//                      (one leaf per lane, SoA loads/stores are coalesced) and stage the gate parameters in
//                      LDS; then each wavefront takes leaves of the tile and sweeps the scan 64 measurements
//                      per step (scan staged in LDS once per workgroup): a cheap conservative float32
//                      bounding-box test on all lanes, the exact reference-order NIS only on lanes that pass,
//                      `__ballot` turns the outcome into one 64-bit hit-mask word per step.  No (L,M) tensor
//                      is ever materialised (the reference builds a 40 MB z_tilde and a 20 MB NIS array).
//   emit_kernel        one wavefront per 64 leaves, one leaf per lane: exclusive scan of (1 + hits) gives the
//                      dense, DFS-ordered child index of every leaf; children are written in ascending
//                      measurement order by walking the hit mask.
// Matrices are 4x4 / 2x2: registers only, no MFMA; the bound is HBM + launch latency.
#include "mht_kernels.h"

namespace mht {


struct LeafGate {        // per-leaf gate parameters staged in LDS
    double zhat[2];
    float sinv[4];
    float bx, by;        // conservative half-widths of the gate's bounding box
    float zhx, zhy;      // float32 copy of z_hat for the pre-filter
    int f32state;
    int valid;
};

template <typename TS>
__device__ __forceinline__ void load_leaf(const GateArgs& a, int src, TS* xs, float* P) {
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[k] = (TS)a.x[(size_t)k * a.cap_in + src];
    const int c = a.cov[src];
#pragma unroll
    for (int e = 0; e < 16; ++e) P[e] = a.P[(size_t)e * a.capc_in + c];
}

// forest mode: position i in the implicit leaf list -> (target slot, node of the previous layer)
__device__ __forceinline__ void locate_leaf(const int* off, const int32_t* first, int nT, int i, int& tgt, int& src) {
    int lo = 0, hi = nT;                 // largest t with off[t] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    tgt = lo;
    src = first[lo] + (i - off[lo]);
}

__device__ __forceinline__ void box_from_S(const float* S, double eta2, float zhx, float zhy, float& bx, float& by) {
    // NIS <= eta2  =>  |dz_x| <= sqrt(eta2*S00), |dz_y| <= sqrt(eta2*S11); widen for float32 rounding of the
    // pre-filter subtraction (coordinates up to ~1e6 m) -- the exact test below decides, this only prunes.
    float rx = sqrtf((float)eta2 * fabsf(S[0])), ry = sqrtf((float)eta2 * fabsf(S[3]));
    bx = rx * 1.001f + 1e-6f * (fabsf(zhx) + rx) + 1e-3f;
    by = ry * 1.001f + 1e-6f * (fabsf(zhy) + ry) + 1e-3f;
}

__global__ __launch_bounds__(GATE_THREADS) void gate_count_kernel(const GateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = a.M, W = a.W;
    const int Mpad = W * 64;
    float* zx = reinterpret_cast<float*>(smem);
    float* zy = zx + Mpad;
    LeafGate* lg = reinterpret_cast<LeafGate*>(zy + Mpad);
    unsigned long long* used_l = reinterpret_cast<unsigned long long*>(lg + GATE_TILE);
    int* tile_total = reinterpret_cast<int*>(used_l + W);
    int* off = tile_total + 4;            // [nT+1] forest mode

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nT = a.t_leaf_off ? *a.nT_dev : 0;
    const int L = a.t_leaf_off ? a.t_leaf_off[nT] : a.L;
    const int ntiles = (L + GATE_TILE - 1) / GATE_TILE;
    if ((int)blockIdx.x >= ntiles) return;
    if (a.t_leaf_off)
        for (int j = tid; j <= nT; j += GATE_THREADS) off[j] = a.t_leaf_off[j];

    for (int j = tid; j < Mpad; j += GATE_THREADS) {
        float2 v = (j < M) ? reinterpret_cast<const float2*>(a.z)[j] : make_float2(3.0e38f, 3.0e38f);
        zx[j] = v.x;
        zy[j] = v.y;
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int w = tid; w < W; w += GATE_THREADS) used_l[w] = 0ull;
        if (tid == 0) *tile_total = 0;
        // ---- phase 1: predict + precalc, one leaf per lane ------------------------------------------------
        if (tid < GATE_TILE) {
            const int i = tile * GATE_TILE + tid;
            LeafGate g;
            g.valid = i < L;
            if (g.valid) {
                int src = a.leaf_src ? a.leaf_src[i] : i, tg_unused;
                if (a.t_leaf_off) locate_leaf(off, a.t_first, nT, i, tg_unused, src);
                const uint8_t fl = a.flags[src];
                float P[16];
                float Pb[16], Ph[16], S[4];
                if (fl & F_STATE_F32) {
                    float xs[4];
                    load_leaf<float>(a, src, xs, P);
                    Predicted<float> p;
                    predict_precalc<float>(a.model, xs, P, p);
                    g.zhat[0] = p.z_hat[0]; g.zhat[1] = p.z_hat[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { g.sinv[e] = p.S_inv[e]; S[e] = p.S[e]; }
#pragma unroll
                    for (int e = 0; e < 16; ++e) { Pb[e] = p.P_bar[e]; Ph[e] = p.P_hat[e]; }
                    g.f32state = 1;
                } else {
                    double xs[4];
                    load_leaf<double>(a, src, xs, P);
                    Predicted<double> p;
                    predict_precalc<double>(a.model, xs, P, p);
                    g.zhat[0] = p.z_hat[0]; g.zhat[1] = p.z_hat[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { g.sinv[e] = p.S_inv[e]; S[e] = p.S[e]; }
#pragma unroll
                    for (int e = 0; e < 16; ++e) { Pb[e] = p.P_bar[e]; Ph[e] = p.P_hat[e]; }
                    g.f32state = 0;
                }
                g.zhx = (float)g.zhat[0];
                g.zhy = (float)g.zhat[1];
                box_from_S(S, a.model.eta2, g.zhx, g.zhy, g.bx, g.by);
                if (2 * i + 1 < a.capc_out) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        a.oP[(size_t)e * a.capc_out + 2 * i] = Pb[e];
                        a.oP[(size_t)e * a.capc_out + 2 * i + 1] = Ph[e];
                    }
                } else {
                    a.status->overflow = 1;
                }
            }
            lg[tid] = g;
        }
        __syncthreads();
        // ---- phase 2: one wavefront per leaf, 64 measurements per step ----------------------------------------
        for (int t = wave; t < GATE_TILE; t += GATE_THREADS / 64) {
            const LeafGate g = lg[t];
            if (!g.valid) continue;
            const int i = tile * GATE_TILE + t;
            unsigned long long myword = 0ull;
            int cnt = 0;
            for (int s = 0; s < W; ++s) {
                const float mx = zx[s * 64 + lane], my = zy[s * 64 + lane];
                const bool cand = (fabsf(mx - g.zhx) <= g.bx) && (fabsf(my - g.zhy) <= g.by);
                bool hit = false;
                if (cand) {
                    if (g.f32state) {
                        float zh[2] = {(float)g.zhat[0], (float)g.zhat[1]}, zt[2], nis;
                        hit = gate_pair<float>(zh, g.sinv, mx, my, (float)a.model.eta2, zt, nis);
                    } else {
                        double zt[2], nis;
                        hit = gate_pair<double>(g.zhat, g.sinv, mx, my, a.model.eta2, zt, nis);
                    }
                }
                const unsigned long long word = __ballot(hit);
                if (lane == s) myword = word;
                cnt += __popcll(word);
            }
            if (lane < W) {
                a.hitmask[(size_t)i * W + lane] = myword;
                if (myword) atomicOr(&used_l[lane], myword);
            }
            if (lane == 0) {
                a.cnt[i] = cnt;
                atomicAdd(tile_total, cnt);
            }
        }
        __syncthreads();
        if (tid == 0) a.tile_cnt[tile] = *tile_total;
        if (a.used)
            for (int w = tid; w < W; w += GATE_THREADS)
                if (used_l[w]) atomicOr(&a.used[w], used_l[w]);
        __syncthreads();
    }
}

template <typename TS>
__device__ __forceinline__ void emit_children(const GateArgs& a, int i, int src, int tgt, uint8_t fl, int base, double cn, double pd) {
    TS xs[4];
    float P[16];
    load_leaf<TS>(a, src, xs, P);
    Predicted<TS> p;
    predict_precalc<TS>(a.model, xs, P, p);
    const float lnc = nllr_const(p.S, a.model.lambda_ex, pd);
    const TS eta2 = (TS)a.model.eta2;
    const size_t cap = a.cap_out;
    // forest extras: path of the parent leaf, shifted by the root advance of its target
    int depth = 0, shift = 0;
    int ppath[MAXPD];
    if (tgt >= 0) {
        depth = a.tgt_depth[tgt];
        shift = a.tgt_shift[tgt];
#pragma unroll
        for (int d = 0; d < MAXPD; ++d) ppath[d] = (d < depth) ? a.in_path[(size_t)(d + shift) * a.cap_in + src] : -1;
    }
    double rootc = 0.0;
    bool root_f32 = false;
    if (a.ocost) { rootc = a.t_root_cnllr[tgt]; root_f32 = a.t_root_f32[tgt] != 0; }
    auto write_common = [&](int c, int meas, int covcol, uint8_t cfl, double cnl, double inc) {
        if (a.ocost) {
            // getScore()/N (pyTarget.py:124, tracker.py:1127) with NumPy's scalar promotion: float32 - float32
            // and float32 / int stay float32
            if ((cfl & F_SCORE_F32) && root_f32) a.ocost[c] = (double)(((float)cnl - (float)rootc) / (float)a.Nwin);
            else a.ocost[c] = (cnl - rootc) / (double)a.Nwin;
        }
        a.ocnllr[c] = cnl;
        a.opd[c] = pd;
        a.oparent[c] = src;
        a.omeas[c] = meas;
        a.ocov[c] = covcol;
        a.oflags[c] = cfl;
        if (a.nllr) a.nllr[c] = inc;
        if (a.out_path) {
#pragma unroll
            for (int d = 0; d < MAXPD; ++d)
                if (d < a.PD) {
                    int v = ppath[d];
                    if (d == depth && meas > 0) v = a.cur_slot_base + meas - 1;
                    a.out_path[(size_t)d * cap + c] = v;
                }
            a.out_tgt[c] = tgt;
            if (meas > 0) a.used_bytes[meas - 1] = 1;
        }
    };
    // missed-detection child (pyTarget.py:319-328)
    {
        const int c = base;
        if (c < (int)cap) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.ox[(size_t)k * cap + c] = (double)p.x_bar[k];
            const double inc = (pd == a.default_pd) ? a.default_miss_nllr : -log(1.0 - pd);
            write_common(c, 0, 2 * i, (uint8_t)(fl & F_STATE_F32), cn + inc, inc);
        }
    }
    // one child per gated measurement, ascending index (pyTarget.py:242-254)
    int c = base + 1;
    const float2* z2 = reinterpret_cast<const float2*>(a.z);
    for (int w = 0; w < a.W; ++w) {
        unsigned long long bits = a.hitmask[(size_t)i * a.W + w];
        while (bits) {
            const int b = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const int j = w * 64 + b;
            if (c < (int)cap) {
                const float2 m = z2[j];
                TS zt[2], nis, xh[4];
                gate_pair<TS>(p.z_hat, p.S_inv, m.x, m.y, eta2, zt, nis);
                update_state<TS>(p.x_bar, p.K, zt, xh);
#pragma unroll
                for (int k = 0; k < 4; ++k) a.ox[(size_t)k * cap + c] = (double)xh[k];
                const TS inc = (TS)0.5 * nis + (TS)lnc;      // kalman.py:19
                double cnl;
                uint8_t cfl = (uint8_t)(fl & F_STATE_F32);
                if (sizeof(TS) == 4 && (fl & F_SCORE_F32)) {   // float32 + float32 stays float32 (NumPy scalar rules)
                    cnl = (double)((float)cn + (float)inc);
                    cfl |= F_SCORE_F32;
                } else {
                    cnl = cn + (double)inc;
                }
                write_common(c, j + 1, 2 * i + 1, cfl, cnl, (double)inc);
            }
            ++c;
        }
    }
    // association bitset of the target: ancestors below the root + everything gated now (tracker.py:255-258)
    if (a.assoc && tgt >= 0) {
        unsigned long long* row = a.assoc + (size_t)tgt * a.assoc_words;
        // Every tree node with a real measurement is contributed exactly once: by the leaf reached from it through
        // missed detections only, i.e. each leaf contributes the LAST real measurement on its path.
        int last = -1;
#pragma unroll
        for (int d = 0; d < MAXPD; ++d)
            if (d < depth && ppath[d] >= 0) last = ppath[d];
        if (last >= 0) atomicOr(&row[last >> 6], 1ull << (last & 63));
        for (int w = 0; w < a.W; ++w) {
            const unsigned long long bits = a.hitmask[(size_t)i * a.W + w];
            if (bits) {
                // measurement-node ids of this scan start at cur_slot_base (a multiple of 64)
                atomicOr(&row[(a.cur_slot_base >> 6) + w], bits);
            }
        }
    }
}

__global__ __launch_bounds__(EMIT_THREADS) void emit_kernel(const GateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* off = reinterpret_cast<int*>(smem);
    const int lane = threadIdx.x;
    const int nT = a.t_leaf_off ? *a.nT_dev : 0;
    const int L = a.t_leaf_off ? a.t_leaf_off[nT] : a.L;
    if (a.t_leaf_off) {
        for (int j = lane; j <= nT; j += EMIT_THREADS) off[j] = a.t_leaf_off[j];
        __syncthreads();
    }
    const int nblocks = (L + EMIT_THREADS - 1) / EMIT_THREADS;
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int i = blk * EMIT_THREADS + lane;
        // children of all leaves before this block: (#leaves before) + (hits before), hits summed per gate tile
        const int tiles_before = blk * (EMIT_THREADS / GATE_TILE);
        int part = 0;
        for (int t = lane; t < tiles_before; t += 64) part += a.tile_cnt[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        const int mine = (i < L) ? 1 + a.cnt[i] : 0;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int base = blk * EMIT_THREADS + part + incl - mine;
        if (i < L) {
            a.child_ptr[i] = base;
            if (i == L - 1) {
                const int total = base + mine;
                a.child_ptr[L] = total;
                a.status->n_children = total;
                if (total > a.cap_out) a.status->overflow = 1;
            }
            int src = a.leaf_src ? a.leaf_src[i] : i, tg = -1;
            if (a.t_leaf_off) {
                locate_leaf(off, a.t_first, nT, i, tg, src);
                if (i == off[tg]) a.tchild[tg] = base;
                if (i == L - 1) a.tchild[tg + 1] = base + mine;
            }
            const uint8_t fl = a.flags[src];
            const double cn = a.cnllr[src], pd = a.pd[src];
            if (fl & F_STATE_F32) emit_children<float>(a, i, src, tg, fl, base, cn, pd);
            else emit_children<double>(a, i, src, tg, fl, base, cn, pd);
        }
    }
    if (L == 0 && blockIdx.x == 0 && lane == 0) {
        a.child_ptr[0] = 0;
        a.status->n_children = 0;
        if (a.tchild) a.tchild[0] = 0;
    }
}

static inline size_t gate_lds_bytes(int W, int Tcap) {
    return (size_t)2 * W * 64 * sizeof(float) + GATE_TILE * sizeof(LeafGate) + (size_t)W * 8 + 16 + (size_t)(Tcap + 1) * 4;
}

int launch_gate(mht_ctx* ctx, GateArgs& a, int grid_leaves_hint) {
    const int L = grid_leaves_hint > a.L ? grid_leaves_hint : a.L, W = a.W;
    a.status = ctx->status;
    if (L <= 0) {
        hipLaunchKernelGGL(emit_kernel, dim3(1), dim3(EMIT_THREADS), (size_t)((a.t_leaf_off ? a.Tcap : 0) + 1) * 4, ctx->stream, a);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
    const int ntiles = (L + GATE_TILE - 1) / GATE_TILE;
    int rc = ctx->hitmask.ensure((size_t)L * W * 8);
    if (rc) return rc;
    rc = ctx->counts.ensure(((size_t)L + ntiles + 8) * 4);
    if (rc) return rc;
    a.hitmask = static_cast<unsigned long long*>(ctx->hitmask.ptr);
    a.cnt = static_cast<int32_t*>(ctx->counts.ptr);
    a.tile_cnt = a.cnt + L;
    a.status = ctx->status;
    const int Tl = a.t_leaf_off ? a.Tcap : 0;
    const int gate_blocks = ntiles < 1024 ? ntiles : 1024;
    hipLaunchKernelGGL(gate_count_kernel, dim3(gate_blocks), dim3(GATE_THREADS), gate_lds_bytes(W, Tl), ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    const int eblocks = (L + EMIT_THREADS - 1) / EMIT_THREADS;
    hipLaunchKernelGGL(emit_kernel, dim3(eblocks < 1024 ? eblocks : 1024), dim3(EMIT_THREADS), (size_t)(Tl + 1) * 4, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

void fill_model(GateArgs& a, const mht_model* m) {
    for (int i = 0; i < 16; ++i) { a.model.A[i] = m->A[i]; a.model.Q[i] = m->Q[i]; }
    for (int i = 0; i < 8; ++i) a.model.C[i] = m->C[i];
    for (int i = 0; i < 4; ++i) a.model.R[i] = m->R[i];
    a.model.eta2 = m->eta2;
    a.model.lambda_ex = m->lambda_ex;
    a.default_pd = m->default_pd;
    a.default_miss_nllr = m->default_miss_nllr;
}

}  // namespace mht

using namespace mht;

extern "C" int mht_gate_scan(mht_ctx* ctx, const mht_model* model, const mht_nodes* in, const int32_t* leaf_src,
                             int32_t L, const float* z, int32_t M, const mht_nodes* out, int32_t* child_ptr,
                             double* nllr, uint64_t* used, int32_t* n_children) {
    MHT_REQUIRE(ctx && model && in && out && child_ptr, "mht_gate_scan: null argument");
    MHT_REQUIRE(L >= 0 && M >= 0 && M <= MAX_MEAS, "mht_gate_scan: need 0 <= M <= %d, L >= 0 (M=%d L=%d)", MAX_MEAS, M, L);
    MHT_REQUIRE(z || M == 0, "mht_gate_scan: z is null");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    GateArgs a = {};
    fill_model(a, model);
    a.x = in->x; a.cnllr = in->cnllr; a.pd = in->pd; a.cov = in->cov; a.flags = in->flags; a.P = in->P;
    a.cap_in = in->cap; a.capc_in = in->cap_cov;
    a.leaf_src = leaf_src; a.L_dev = nullptr; a.L = L;
    a.z = z; a.M = M; a.W = (M + 63) / 64;
    a.ox = out->x; a.ocnllr = out->cnllr; a.opd = out->pd; a.oparent = out->parent; a.omeas = out->meas;
    a.ocov = out->cov; a.oflags = out->flags; a.oP = out->P; a.cap_out = out->cap; a.capc_out = out->cap_cov;
    a.child_ptr = child_ptr; a.nllr = nllr; a.used = reinterpret_cast<unsigned long long*>(used);
    MHT_HIP_CHECK(hipMemsetAsync(ctx->status, 0, sizeof(DevStatus), ctx->stream));
    int rc = launch_gate(ctx, a, L);
    if (rc) return rc;
    if (n_children) {
        DevStatus st;
        MHT_HIP_CHECK(hipMemcpyAsync(&st, ctx->status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *n_children = st.n_children;
        if (st.overflow) {
            set_error("mht_gate_scan: output layer too small (need cap >= %d, cap_cov >= %d)", st.n_children, 2 * L);
            return MHT_E_CAPACITY;
        }
    }
    return MHT_OK;
}
