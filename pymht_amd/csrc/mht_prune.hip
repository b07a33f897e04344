// Seam (iv): N-scan pruning of host-owned hypothesis trees -- Tracker._nScanPruning (pymht/tracker.py:1229-1231) ->
// _pruneTargetIndex (:1219-1227) -> Target.pruneDepth (pymht/pyTarget.py:343-356) -> _pruneAllHypothesisExceptThis(backtrack=True)
// (:330-337) -- for a caller that keeps its own node arrays (the device forest does the same inside blp_kernel's epilogue).
//
// Trees are given by parent pointers.  Per target: walk `window` parents up from the selected leaf; that ancestor is the new root;
// every node that is neither on the chain root-of-time .. new root nor below the new root is deleted.  If the walk reaches the top
// of the tree first, nothing is deleted and the top stays the root (pyTarget.py:353-356).
#include "mht_kernels.h"

namespace mht {

// one thread per target: new root + marks on the kept chain (2 = new root, 1 = its ancestors)
__global__ void prune_roots_kernel(int T, const int32_t* __restrict__ parent, const int32_t* __restrict__ sel,
                                   const int32_t* __restrict__ window, int n_nodes, int32_t* new_root, uint8_t* mark) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int n = sel[t], steps = window[t];
    if (n < 0 || n >= n_nodes) { new_root[t] = -1; return; }
    while (steps > 0) {
        const int p = parent[n];
        if (p < 0) break;
        n = p;
        --steps;
    }
    new_root[t] = n;
    mark[n] = 2;
    int guard = n_nodes;
    for (int p = parent[n]; p >= 0 && guard-- > 0; p = parent[p]) mark[p] = 1;
}

// one thread per node: survives iff it is marked, or the first marked node above it is a new root
__global__ void prune_keep_kernel(int n_nodes, const int32_t* __restrict__ parent, const uint8_t* __restrict__ mark, uint8_t* keep) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    int cur = v, guard = n_nodes;
    uint8_t m = mark[cur];
    while (m == 0 && guard-- > 0) {
        cur = parent[cur];
        if (cur < 0) break;
        m = mark[cur];
    }
    keep[v] = (cur >= 0 && (m == 2 || (m == 1 && cur == v))) ? 1 : 0;
}

}  // namespace mht

using namespace mht;

extern "C" int mht_prune(mht_ctx* ctx, int32_t n_nodes, const int32_t* parent, int32_t T, const int32_t* sel, const int32_t* window,
                         int32_t* new_root, uint8_t* keep) {
    MHT_REQUIRE(ctx && (parent || n_nodes == 0) && (T == 0 || (sel && window && new_root)) && (keep || n_nodes == 0), "mht_prune: null argument");
    MHT_REQUIRE(n_nodes >= 0 && T >= 0, "mht_prune: negative size");
    if (n_nodes == 0 || T == 0) {
        if (n_nodes > 0) MHT_HIP_CHECK(hipMemsetAsync(keep, 0, (size_t)n_nodes, ctx->stream));
        return MHT_OK;
    }
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ctx->counts.ensure((size_t)n_nodes + 16);
    if (rc) return rc;
    uint8_t* mark = static_cast<uint8_t*>(ctx->counts.ptr);
    MHT_HIP_CHECK(hipMemsetAsync(mark, 0, (size_t)n_nodes, ctx->stream));
    hipLaunchKernelGGL(prune_roots_kernel, dim3((T + 255) / 256), dim3(256), 0, ctx->stream, T, parent, sel, window, n_nodes, new_root, mark);
    MHT_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(prune_keep_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, ctx->stream, n_nodes, parent, mark, keep);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}
