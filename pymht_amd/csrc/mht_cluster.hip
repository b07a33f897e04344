// Clustering: connected components of the bipartite graph  targets <-> measurements-of-the-window
// (reference: Tracker._findClustersFromSets, pymht/tracker.py:961-974, which builds a dense
// (T+|superSet|)^2 adjacency matrix in a Python double loop and calls scipy connected_components).
//
// Input is one bitset per target over the "measurement nodes" of the N-scan window (bit = ring_slot*Mpad + m),
// filled by the emit kernel (ancestors below the root + everything gated in this scan) -- exactly the
// reference's __associatedMeasurements__ sets.  One workgroup: expand the bitsets to an edge list, then
// min-label propagation with pointer jumping in LDS until a fixed point.  Labels are target indices, the
// fixed point is the smallest member of each component, so clusters come out ordered by smallest member with
// ascending members -- the order scipy's labelling + np.where gives the reference (tracker.py:972-974).
#include "mht_kernels.h"

namespace mht {


constexpr int CL_THREADS = 1024;
constexpr int CL_ELDS = 16384;     // edges kept in LDS (packed target<<16 | node); the rest spills to HBM scratch

__device__ __forceinline__ unsigned cl_edge(const unsigned* eL, const ClusterArgs& a, int e) {
    if (e < CL_ELDS) return eL[e];
    return ((unsigned)a.edge_t[e - CL_ELDS] << 16) | (unsigned)a.edge_m[e - CL_ELDS];
}

__global__ __launch_bounds__(CL_THREADS) void cluster_kernel(const ClusterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* tlabel = reinterpret_cast<int*>(smem);          // [Tcap]
    int* aux = tlabel + a.Tcap;                          // [Tcap]  cluster index of a head
    int* mlabel = aux + a.Tcap;                          // [n_mnodes]
    const int mslots = a.n_mnodes > 3 * a.Tcap + 2 ? a.n_mnodes : 3 * a.Tcap + 2;   // the slot is re-used for the cluster tables
    unsigned* eL = reinterpret_cast<unsigned*>(mlabel + mslots);   // [CL_ELDS]
    __shared__ int s_edges, s_changed, s_big, s_scan[CL_THREADS / 64], s_total;
    const int tid = threadIdx.x;
    if (a.status && a.status->overflow) return;       // a pool overflowed in grow_kernel: the scan is void (commit reports it)
    const int T = *a.nT_dev;
    const unsigned long long t0 = wall_clock64();
#define CL_STAMP(q) do { if (a.dbg && tid == 0) a.dbg[q] = (int)(wall_clock64() - t0); } while (0)
    for (int t = tid; t < T; t += CL_THREADS) tlabel[t] = t;
    if (tid == 0) { s_edges = 0; a.counts[3] = 0; }
    __syncthreads();
    int E;
    unsigned long long* rows = const_cast<unsigned long long*>(a.assoc);
    if (a.edges_in) {
        // forest mode: grow_kernel already produced the deduplicated edge list, in EDGE_SEGS counted segments
        __shared__ int s_segoff[EDGE_SEGS + 1];
        if (tid < 64) {
            int n = a.edge_count[tid];
            if (n > a.seg_cap) { n = a.seg_cap; a.counts[3] = 1; }
            int incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (tid >= o) incl += v;
            }
            s_segoff[tid] = incl - n;
            if (tid == 63) s_segoff[64] = incl;
        }
        __syncthreads();
        E = s_segoff[EDGE_SEGS];
        if (E > CL_ELDS + a.Ecap) { if (tid == 0) a.counts[3] = 1; E = CL_ELDS + a.Ecap; }
        // flat gather: thread -> dense edge index -> (segment, offset) by binary search over the 65 offsets
        for (int e = tid; e < E; e += CL_THREADS) {
            int lo = 0, hi = EDGE_SEGS;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_segoff[mid] <= e) lo = mid; else hi = mid; }
            const unsigned pk = a.edges_in[(size_t)lo * a.seg_cap + (e - s_segoff[lo])];
            if (e < CL_ELDS) eL[e] = pk;
            else { a.edge_t[e - CL_ELDS] = (int)(pk >> 16); a.edge_m[e - CL_ELDS] = (int)(pk & 0xffff); }
        }
        __threadfence_block();
        __syncthreads();
    } else {
        // bitsets -> edge list; the rows are cleared on the way (no memset between scans).  4 independent loads in
        // flight per thread: the sweep is latency bound otherwise.
        const long long nwords = (long long)T * a.AW;
        for (long long base = 0; base < nwords; base += 4LL * CL_THREADS) {
            unsigned long long v[4];
    #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long idx = base + (long long)q * CL_THREADS + tid;
                v[q] = idx < nwords ? rows[idx] : 0ull;
            }
    #pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned long long bits = v[q];
                if (!bits) continue;
                const long long idx = base + (long long)q * CL_THREADS + tid;
                if (a.clear_rows) rows[idx] = 0ull;
                const int t = (int)(idx / a.AW), w = (int)(idx % a.AW);
                int pos = atomicAdd(&s_edges, __popcll(bits));
                while (bits) {
                    const int b = __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    const int m = w * 64 + b;
                    if (pos < CL_ELDS) eL[pos] = ((unsigned)t << 16) | (unsigned)m;
                    else if (pos - CL_ELDS < a.Ecap) { a.edge_t[pos - CL_ELDS] = t; a.edge_m[pos - CL_ELDS] = m; }
                    ++pos;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        E = s_edges;
        if (E > CL_ELDS + a.Ecap) {
            if (tid == 0) a.counts[3] = 1;
            E = CL_ELDS + a.Ecap;
        }
    }
    CL_STAMP(0);
    // connected components by lock-free union-find in LDS: vertices = targets (0..T-1) and measurement nodes (mlabel[]
    // doubles as their parent array, value = vertex id, T + m).  Every edge hooks the larger root under the smaller one
    // with atomicCAS; targets have the smaller ids, so the root of a component is its smallest target -- the label the
    // reference's scipy labelling + np.where ordering implies (tracker.py:972-974).  One pass over the edges, one
    // compression pass: no iteration to a fixed point.
    int iters_done = 1;
    for (int m = tid; m < a.n_mnodes; m += CL_THREADS) mlabel[m] = T + m;
    __syncthreads();
    auto parent_of = [&](int v) -> int { return v < T ? tlabel[v] : mlabel[v - T]; };
    auto find_root = [&](int v) -> int {
        int p = parent_of(v);
        while (p != v) { v = p; p = parent_of(v); }
        return v;
    };
    for (int e = tid; e < E; e += CL_THREADS) {
        const unsigned pk = cl_edge(eL, a, e);
        int ra = find_root((int)(pk >> 16)), rb = find_root(T + (int)(pk & 0xffff));
        while (ra != rb) {
            if (ra > rb) { const int tmp = ra; ra = rb; rb = tmp; }      // ra < rb: hook rb under ra
            int* slot = rb < T ? &tlabel[rb] : &mlabel[rb - T];
            const int old = atomicCAS(slot, rb, ra);
            if (old == rb) break;
            rb = find_root(old);          // somebody else moved rb meanwhile: retry from its new root
            ra = find_root(ra);
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += CL_THREADS) {
        const int r = find_root(t);
        aux[t] = r;                       // stash: writing tlabel here would race with other threads' find_root
    }
    __syncthreads();
    for (int t = tid; t < T; t += CL_THREADS) tlabel[t] = aux[t];
    __syncthreads();
    CL_STAMP(1);
    if (a.dbg && tid == 0) a.dbg[6] = iters_done;
    if (a.edges_in) {      // clear the dedup bitsets for the next scan and hand the counters back
        for (int e = tid; e < E; e += CL_THREADS) {
            const unsigned pk = cl_edge(eL, a, e);
            rows[(size_t)(pk >> 16) * a.AW + ((pk & 0xffff) >> 6)] = 0ull;
        }
        if (tid < EDGE_SEGS) a.edge_count[tid] = 0;
    }
    CL_STAMP(2);
    // heads -> cluster indices (exclusive scan over targets, chunked)
    int running = 0;
    for (int base = 0; base < T; base += CL_THREADS) {
        const int t = base + tid;
        const int head = (t < T && tlabel[t] == t) ? 1 : 0;
        int incl = head;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_scan[wv] = incl;
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int i = 0; i < CL_THREADS / 64; ++i) { const int v = s_scan[i]; s_scan[i] = acc; acc += v; }
            s_total = acc;
        }
        __syncthreads();
        if (head) aux[t] = running + s_scan[wv] + incl - 1;     // cluster index of head t
        running += s_total;
        __syncthreads();
    }
    const int nC = running;
    CL_STAMP(3);
    // cluster sizes and offsets in LDS (csize/cptr alias the measurement-label array, which is dead by now)
    int* csize = mlabel;                 // [nC]
    int* cptr = mlabel + a.Tcap;         // [nC+1]
    int* mheads = mlabel + 2 * a.Tcap + 1;   // [<= nC] heads of the multi-target clusters
    for (int c = tid; c < nC; c += CL_THREADS) csize[c] = 0;
    if (tid == 0) { s_edges = 0; s_changed = 0; s_big = 0; }
    __syncthreads();
    for (int t = tid; t < T; t += CL_THREADS) {
        a.t_label[t] = tlabel[t];
        a.t_cluster[t] = aux[tlabel[t]];
        atomicAdd(&csize[aux[tlabel[t]]], 1);
    }
    __syncthreads();
    running = 0;
    for (int base = 0; base < nC; base += CL_THREADS) {       // exclusive scan of the sizes
        const int c = base + tid;
        const int v = (c < nC) ? csize[c] : 0;
        int incl = v;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) s_scan[wv] = incl;
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int i = 0; i < CL_THREADS / 64; ++i) { const int x = s_scan[i]; s_scan[i] = acc; acc += x; }
            s_total = acc;
        }
        __syncthreads();
        if (c < nC) {
            const int p0 = running + s_scan[wv] + incl - v;
            cptr[c] = p0;
            a.cl_ptr[c] = p0;
        }
        running += s_total;
        __syncthreads();
    }
    if (tid == 0) { cptr[nC] = running; a.cl_ptr[nC] = running; }
    CL_STAMP(4);
    // per-cluster linked lists of the non-head members (push order is arbitrary, the lists are sorted below)
    // (the three tables live in the LDS edge list, which is dead by now; 3*Tcap <= CL_ELDS is checked at launch)
    int* lhead = reinterpret_cast<int*>(eL);   // [T] first list element of the cluster headed by t, -1 = none
    int* lnext = lhead + a.Tcap;               // [T]
    int* tmp = lnext + a.Tcap;                 // [T] scratch: every cluster owns the segment [cptr[c], cptr[c+1])
    for (int t = tid; t < T; t += CL_THREADS) lhead[t] = -1;
    __syncthreads();
    for (int t = tid; t < T; t += CL_THREADS) {
        const int r = tlabel[t];
        if (r != t) lnext[t] = atomicExch(&lhead[r], t);
    }
    __syncthreads();
    // work lists: heads of multi-target clusters (solved by blp_kernel), targets alone in their cluster; the members of a
    // multi-target cluster are collected by its head, sorted ascending (insertion sort in LDS: clusters are small) and
    // written out; clusters with more than 64 members are left to the wavefront sweep below
    for (int t = tid; t < T; t += CL_THREADS) {
        const int c = aux[tlabel[t]];
        const int K = csize[c];
        if (K == 1) {
            a.cl_members[cptr[c]] = t;
            a.single_list[atomicAdd(&s_changed, 1)] = t;
        } else if (tlabel[t] == t) {
            a.multi_list[atomicAdd(&s_edges, 1)] = c;
            if (K <= 64) {
                int* seg = tmp + cptr[c];
                seg[0] = t;                               // the head is the smallest member
                int n = 1;
                for (int q = lhead[t]; q >= 0 && n < K; q = lnext[q]) {
                    int pos = n++;
                    while (pos > 1 && seg[pos - 1] > q) { seg[pos] = seg[pos - 1]; --pos; }
                    seg[pos] = q;
                }
                for (int k = 0; k < K; ++k) a.cl_members[cptr[c] + k] = seg[k];
            } else {
                mheads[atomicAdd(&s_big, 1)] = t;
            }
        }
    }
    __syncthreads();
    {   // big clusters: one wavefront per cluster sweeps the labels in ascending order
        const int lane = tid & 63, wv = tid >> 6;
        const int nB = s_big;
        for (int i = wv; i < nB; i += CL_THREADS / 64) {
            const int l = mheads[i];
            const int base = cptr[aux[l]];
            int run = 0;
            for (int t0 = l & ~63; t0 < T; t0 += 64) {
                const int t = t0 + lane;
                const bool m = t < T && tlabel[t] == l;
                const unsigned long long bal = __ballot(m);
                if (m) a.cl_members[base + run + __popcll(bal & ((1ull << lane) - 1ull))] = t;
                run += __popcll(bal);
            }
        }
    }
    CL_STAMP(5);
    if (a.dbg && tid == 0) a.dbg[7] = E;
    __syncthreads();
    if (tid == 0) {
        a.counts[0] = nC;
        a.counts[1] = s_edges;
        a.counts[2] = s_changed;
    }
}

size_t cluster_lds_bytes(int Tcap, int n_mnodes) {
    const int mslots = n_mnodes > 3 * Tcap + 2 ? n_mnodes : 3 * Tcap + 2;
    return (size_t)(2 * Tcap + mslots + CL_ELDS) * 4;
}

int launch_cluster(mht_ctx* ctx, const ClusterArgs& a) {
    size_t& attr_bytes = ctx->lds_attr_cluster;
    const size_t lds = cluster_lds_bytes(a.Tcap, a.n_mnodes);
    if (lds > 150 * 1024 || a.n_mnodes > 65536 || 3 * a.Tcap > CL_ELDS) {
        set_error("cluster: Tcap=%d and %d measurement nodes need %zu B of LDS (> 150 KiB)", a.Tcap, a.n_mnodes, lds);
        return MHT_E_CAPACITY;
    }
    if (lds > 48 * 1024 && lds > attr_bytes) {
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    hipLaunchKernelGGL(cluster_kernel, dim3(1), dim3(CL_THREADS), lds, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

}  // namespace mht

using namespace mht;

extern "C" int mht_cluster(mht_ctx* ctx, int32_t T, int32_t words, const uint64_t* assoc, int32_t* label) {
    MHT_REQUIRE(ctx && label && (assoc || T == 0), "mht_cluster: null argument");
    MHT_REQUIRE(T >= 0 && words >= 1, "mht_cluster: bad sizes");
    if (T == 0) return MHT_OK;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    size_t ecap = (size_t)T * words * 64;
    if (ecap > (1u << 22)) ecap = 1u << 22;
    const size_t ints = 2 * ecap + 6 * (size_t)T + 32;
    int rc = ctx->counts.ensure(ints * 4);
    if (rc) return rc;
    int32_t* base = static_cast<int32_t*>(ctx->counts.ptr);
    ClusterArgs a = {};
    a.assoc = reinterpret_cast<const unsigned long long*>(assoc);
    a.AW = words;
    a.Tcap = T;
    a.Ecap = (int)ecap;
    a.clear_rows = 0;
    a.n_mnodes = words * 64;
    a.edge_t = base; a.edge_m = base + ecap;
    int32_t* q = base + 2 * ecap;
    a.t_label = label; a.t_cluster = q; a.cl_ptr = q + T; a.cl_members = q + 2 * T + 1; a.multi_list = q + 3 * T + 1;
    a.single_list = q + 4 * T + 1; a.counts = q + 5 * T + 8;
    int32_t* nT_dev = q + 5 * T + 16;
    a.nT_dev = nT_dev;
    MHT_HIP_CHECK(hipMemsetAsync(a.counts, 0, 8 * 4, ctx->stream));
    MHT_HIP_CHECK(hipMemcpyAsync(nT_dev, &T, 4, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_cluster(ctx, a);
    if (rc) return rc;
    int32_t counts[4];
    MHT_HIP_CHECK(hipMemcpyAsync(counts, a.counts, 16, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (counts[3]) {
        set_error("mht_cluster: more than %zu associations", ecap);
        return MHT_E_CAPACITY;
    }
    return MHT_OK;
}
