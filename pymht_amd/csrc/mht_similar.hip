// Similar-state pruning of the forest: Tracker._pruneSimilarState (pymht/tracker.py:1233-1239) -> Target.pruneSimilarState
// (pymht/pyTarget.py:358-412), asked for per scan with addMeasurementList(pruneSimilar=True) (tracker.py:230-231).
//
// For every target that is ALONE in its cluster, and for every node that got children in this scan: the hit children whose
// position lies within `threshold` of the missed-detection child's (= the predicted) position are taken out of the tree, and the
// missed-detection child is REPLACED by one measurement-less node carrying their mean state, mean covariance and mean score.
// The arithmetic is NumPy's, in NumPy's order:
//   distance   float32: positions rounded to float32, dx*dx + dy*dy and its correctly rounded root (np.linalg.norm, axis=1)
//   mean x, P  np.mean(axis=0): rows added one after the other in the array's dtype (float64 or float32 chain; P is float32),
//              then ONE division by the count in that dtype
//   mean score np.mean of a 1-D array: sequential below 8 elements, NumPy's 8-accumulator block sum from 8 on
// In the forest the children of a leaf are contiguous (missed detection first), so a node's children are found from the
// missed-detection child: one lane per such child walks its siblings.  Fused siblings stay where they are and get F_DEAD: the
// next scan's grow kernel, the leaf exports and the neighbour test of initiateTarget skip them; the merged node takes the slot
// (and the path / ancestor records) of the missed-detection child.  Its covariance is a VALUE like any other (mht_vtab.h): when
// the mean of n identical matrices gives the matrix back (n = 1, 2, 4 always) it keeps the hit children's key, otherwise the
// mean is found / inserted and gets a key of its own, as a root does.
#include "mht_kernels.h"

namespace mht {

namespace {

__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ double div_rn(double a, double b) { return a / b; }

__device__ __forceinline__ bool is_near(const SimilarArgs& a, int g, float p0x, float p0y) {
    const float dx = __fsub_rn((float)a.x[g], p0x), dy = __fsub_rn((float)a.x[(size_t)a.cap + g], p0y);
    const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    return d < a.thr;
}

// TP: the dtype of the children's covariance -- float32, or the float64 of a promoted target (AIS forests, mht_vtab.h: F_COV_F64; the hit
// children of one node share their dtypes, np.mean works in the array's)
__device__ __forceinline__ void sim_load(const VTab& vt, int id, float* P) { vt_load(vt, id, P); }
__device__ __forceinline__ void sim_load(const VTab& vt, int id, double* P) { vt_load64(vt, id, P); }
__device__ __forceinline__ bool sim_same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
__device__ __forceinline__ bool sim_same(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }
template <typename TS, typename TP> __device__ __forceinline__ void fuse_group(const SimilarArgs& a, int t, int h, int ce, int n, float p0x, float p0y) {
    const size_t cap = a.cap;
    TS xs[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) xs[k] = (TS)0;
    TP Ps[NP], P1[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) Ps[e] = (TP)0;
    int first = -1, i = 0;
    uint8_t fl = 0;
    bool score_f32 = false;
    Sum1D<double> sd; Sum1D<float> sf;
    for (int g = h + 1; g < ce && a.meas[g] > 0; ++g) {
        if (a.mmsi && a.mmsi[g] != 0) break;      // (fused children come behind the radar children and are never merged)
        if (!is_near(a, g, p0x, p0y)) continue;
        const uint8_t fg = a.flags[g];
        TP P[NP];
        if (sizeof(TP) == 4 && a.ct_Phat) {      // constant-turn forest: the hit children of node (key >> 1) share its P_hat
            const float* src = a.ct_Phat + (size_t)(a.cov[g] >> 1) * NP;
#pragma unroll
            for (int e = 0; e < NP; ++e) P[e] = (TP)src[e];
        } else
        sim_load(a.vt, a.vt.child[a.cov[g]], P);
        if (first < 0) {
            first = g; fl = fg;
            score_f32 = (fg & F_SCORE_F32) != 0;      // (the hit children of one node share their dtypes)
            if (score_f32) sf.begin(n); else sd.begin(n);
#pragma unroll
            for (int k = 0; k < NX; ++k) xs[k] = (TS)a.x[(size_t)k * cap + g];
#pragma unroll
            for (int e = 0; e < NP; ++e) { Ps[e] = P[e]; P1[e] = P[e]; }
        } else {
#pragma unroll
            for (int k = 0; k < NX; ++k) xs[k] = xs[k] + (TS)a.x[(size_t)k * cap + g];
#pragma unroll
            for (int e = 0; e < NP; ++e) Ps[e] = Ps[e] + P[e];
        }
        if (score_f32) sf.add(i, (float)a.cnllr[g]); else sd.add(i, a.cnllr[g]);
        a.flags[g] = (uint8_t)(fg | F_DEAD);
        ++i;
    }
    // the merged node, in the slot of the missed-detection child (pyTarget.py:392-412)
    const TS cnt = (TS)n;
#pragma unroll
    for (int k = 0; k < NX; ++k) a.x[(size_t)k * cap + h] = (double)div_rn(xs[k], cnt);
    bool same = true;
#pragma unroll
    for (int e = 0; e < NP; ++e) {
        Ps[e] = div_rn(Ps[e], (TP)n);
        same = same && sim_same(Ps[e], P1[e]);
    }
    const double cn = score_f32 ? (double)__fdiv_rn(sf.res, (float)n) : sd.res / (double)n;
    a.cnllr[h] = cn;
    const uint8_t mfl = (uint8_t)(fl & (F_STATE_F32 | F_SCORE_F32 | F_COV_F64));
    a.flags[h] = mfl;
    if (same) {
        a.cov[h] = a.cov[first];
    } else if (sizeof(TP) == 4 && a.ct_Pbar) {      // constant-turn forest: the mean under the missed-detection child's own key (see SimilarArgs)
        float* dst = a.ct_Pbar + (size_t)(a.cov[h] >> 1) * NP;
#pragma unroll
        for (int e = 0; e < NP; ++e) dst[e] = (float)Ps[e];
    } else {
        const double pd = a.pd[h];
        if (sizeof(TP) == 8) {      // a float64 mean: two ids for the value, two for its pseudo parent (mht_vtab.h)
            double P64[NP], row[GKF];
#pragma unroll
            for (int e = 0; e < NP; ++e) P64[e] = (double)Ps[e];
            const int id0 = vt_find_or_insert64(a.vt, P64, pd);
            vt_gains64(a.model, P64, pd, row);
            a.cov[h] = vt_pseudo_key64(a.vt, id0, row);
        } else {
            float P32[NP];
#pragma unroll
            for (int e = 0; e < NP; ++e) P32[e] = (float)Ps[e];
            const int id0 = vt_find_or_insert(a.vt, P32, pd);
            const unsigned pid = atomicAdd(a.vt.count, 1u);      // a key of its own: a pseudo parent whose miss child is the mean
            if (pid >= (unsigned)a.vt.vcap) { *a.vt.overflow = 1; return; }
            const int key = 2 * (int)pid;
            float4 rec[GKQ];
            vt_gains(a.model, P32, pd, rec);
#pragma unroll
            for (int e = 0; e < GKQ; ++e) a.vt.Gk[(size_t)key * GKQ + e] = rec[e];
            a.vt.child[key] = id0;
            a.cov[h] = key;
        }
    }
    // its ILP cost, like any child's (fgrow_kernel; nothing reads it before the next scan overwrites the array, kept consistent)
    const double rootc = a.t_root_cnllr[t];
    if ((mfl & F_SCORE_F32) && a.t_root_f32[t]) a.cost[h] = (double)(((float)cn - (float)rootc) / (float)a.Nwin);
    else a.cost[h] = (cn - rootc) / (double)a.Nwin;
}

}  // namespace

__global__ __launch_bounds__(256) void prune_similar_kernel(const SimilarArgs a) {
    if (a.status && a.status->overflow) return;
    if (a.status && blockIdx.x == 0 && threadIdx.x == 0) const_cast<DevStatus*>(a.status)->t[3] = wall_clock64();      // stage stamp
    const int nSingle = a.counts[2];
    const int lane = threadIdx.x & 63, nw = gridDim.x * (blockDim.x >> 6);
    for (int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < nSingle; i += nw) {
        const int t = a.single_list[i];
        // (AIS forest: the reference rebuilds the association set of every target that is alone in its cluster from its tree here,
        // tracker.py:1233-1239 -- the children of this scan included; mht_kernels.h: WIN_REBUILT_*)
        if (a.t_window && lane == 0) {
            const int w = a.t_window[t];
            if ((w >> WIN_REBUILT_SHIFT) != WIN_REBUILT_ALL) a.t_window[t] = (w & 0xff) | ((a.t_depth[t] + 1) << WIN_REBUILT_SHIFT);
        }
        const int cb = a.tchild[t], ce = a.tcend[t];
        for (int h = cb + lane; h < ce; h += 64) {
            if (a.meas[h] != 0) continue;              // (a node's children start with its missed-detection child)
            if (a.mmsi && a.mmsi[h] != 0) continue;    // (a child with an AIS message and no radar measurement is not one)
            const float p0x = (float)a.x[h], p0y = (float)a.x[(size_t)a.cap + h];
            int n = 0;
            for (int g = h + 1; g < ce && a.meas[g] > 0 && !(a.mmsi && a.mmsi[g] != 0); ++g) n += is_near(a, g, p0x, p0y) ? 1 : 0;
            if (n == 0) continue;
            // (dtypes of the HIT children: a promoted target's are float64 throughout, mht_fgrow.hip)
            const uint8_t fh = a.flags[h];
            if (fh & F_COV_F64) fuse_group<double, double>(a, t, h, ce, n, p0x, p0y);
            else if (fh & F_STATE_F32) fuse_group<float, float>(a, t, h, ce, n, p0x, p0y);
            else fuse_group<double, float>(a, t, h, ce, n, p0x, p0y);
        }
    }
}

int launch_prune_similar(mht_ctx* ctx, const SimilarArgs& a, int n_targets_ub) {
    int grid = (n_targets_ub + 3) / 4;
    if (grid < 1) grid = 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(prune_similar_kernel, dim3(grid), dim3(256), 0, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

}  // namespace mht
