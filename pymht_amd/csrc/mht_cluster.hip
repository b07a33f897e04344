// Clustering: connected components of the bipartite graph  targets <-> measurements-of-the-window
// (reference: Tracker._findClustersFromSets, pymht/tracker.py:961-974, which builds a dense
// (T+|superSet|)^2 adjacency matrix in a Python double loop and calls scipy connected_components).
//
// The vertices on the measurement side are the "measurement nodes" of the N-scan window (id = ring_slot*Mpad + m); the
// edges are the reference's __associatedMeasurements__ sets.  In the forest grow_kernel hands over a de-duplicated
// (target, node) edge list in 64 counted segments (the per-target bitsets only serve as its de-duplication filter and are
// cleared here through that list); the stateless seam mht_cluster expands bitsets into the edge list first.
// One workgroup, everything in LDS, ten barrier-separated phases: gather the edges (one global round trip, speculative);
// every node learns its smallest user; the other users are united with it (lock-free union-find over the targets, a
// component's root is its smallest target); root search + member counts; one dual block scan for cluster index and cluster
// offset; slot + rank for the ascending member lists.  Clusters come out ordered by smallest
// member with ascending members -- the order scipy's labelling + np.where gives the reference (tracker.py:972-974).
#include "mht_kernels.h"
#include "mht_init_dev.h"
#include "mht_uf.h"

namespace mht {


constexpr int CL_THREADS = 1024;
static_assert(CL_THREADS / 16 == EDGE_SEGS, "speculative gather: 16 threads per edge segment");
constexpr int CL_SPEC = 128;       // edges per segment fetched speculatively in the first round trip (CL_THREADS / 64 threads x 8)
constexpr int CL_PEND_MAX = 4096;  // (owner, user) target pairs waiting for their union (ClusterArgs::pcap); more are united straight away
constexpr int CL_ELDS_MAX = 16384; // edges kept in LDS (packed target<<16 | node), fewer if the tables need the room (ClusterArgs::elds);
                                   // the rest spills to HBM scratch

__device__ __forceinline__ unsigned cl_edge(const unsigned* eL, const ClusterArgs& a, int e) {
    if (e < a.elds) return eL[e];
    return ((unsigned)a.edge_t[e - a.elds] << 16) | (unsigned)a.edge_m[e - a.elds];
}

// BIG: the tables (4 ints per target + one per measurement node) do not fit LDS -- thousands of targets x tens of thousands of
// measurement nodes -- and live in HBM scratch (ClusterArgs::gtab); the edge list, the pending pairs and the member scratch stay in
// LDS.  Same phases, same results; the barriers between the phases carry an agent-scope fence (the tables are written with
// atomics that live in L2 and read with plain loads that may sit in the CU's L1), and within a phase the union-find tolerates stale
// parents (a parent is only ever replaced by a smaller member of the same component, and every hook is a compare-and-swap in L2).
template <bool BIG>
__device__ __forceinline__ void cluster_body(const ClusterArgs& a, unsigned char* smem) {
#define __syncthreads() do { if (BIG) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __syncthreads(); if (BIG) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); } while (0)
    int* tlabel = BIG ? a.gtab : reinterpret_cast<int*>(smem);          // [Tcap]  union-find parent of a target
    int* lab = tlabel + a.Tcap;                          // [Tcap]  final label = smallest target of the component
    int* cnt = lab + a.Tcap;                             // [Tcap]  by head: members of its cluster
    int* fill = cnt + a.Tcap;                            // [Tcap]  by head: member slots handed out
    int* mlabel = fill + a.Tcap;                         // [n_mnodes] smallest target that uses the measurement node
    const int mslots = a.n_mnodes > 2 * a.Tcap ? a.n_mnodes : 2 * a.Tcap;   // re-used for the cluster tables afterwards
    unsigned* eL = BIG ? reinterpret_cast<unsigned*>(smem) : reinterpret_cast<unsigned*>(mlabel + mslots);   // [a.elds]
    unsigned* pend = eL + a.elds;                                  // [a.pcap] (owner << 16 | user) pairs still to be united
    __shared__ int s_pend;
    __shared__ int s_edges, s_changed, s_scan[CL_THREADS / 64], s_scan2[CL_THREADS / 64], s_total, s_total2;
    const int tid = threadIdx.x;
    // First round trip, everything at once: the status word, the target count, the 64 segment lengths and -- speculatively --
    // the first CL_SPEC edges of every segment (16 threads per segment; a segment holds ~55 edges on the headline config, its
    // capacity is >= 1024, so the loads are always in bounds; what lies beyond a segment's length is masked below).
    const int s_over = a.status ? a.status->overflow : 0;
    const int T = *a.nT_dev;
    int my_n = 0;
    unsigned spec[CL_SPEC / 16];
    if (a.edges_in) {
        if (tid < EDGE_SEGS) my_n = a.edge_count[tid];
        const int sg = tid >> 4, l16 = tid & 15;
#pragma unroll
        for (int q = 0; q < CL_SPEC / 16; ++q) spec[q] = a.edges_in[(size_t)sg * a.seg_cap + l16 + 16 * q];
    }
    if (s_over) return;       // a pool overflowed in grow_kernel: the scan is void (commit reports it)
    if (a.status && tid == 0) const_cast<DevStatus*>(a.status)->t[1] = wall_clock64();      // stage stamp: clustering starts
    if (a.status_other && tid == 0) { a.status_other->overflow = 0; a.status_other->n_children = 0; a.status_other->n_dead = 0; }      // the scan after this one starts from a clean word
    const unsigned long long t0 = wall_clock64();
#define CL_STAMP(q) do { if (a.dbg && tid == 0) a.dbg[q] = (int)(wall_clock64() - t0); } while (0)
    for (int t = tid; t < T; t += CL_THREADS) { tlabel[t] = t; cnt[t] = 0; fill[t] = 0; }
    if (a.sel_rel_reset) for (int t = tid; t < T; t += CL_THREADS) a.sel_rel_reset[t] = -1;
    for (int m = tid; m < a.n_mnodes; m += CL_THREADS) mlabel[m] = 0x7fffffff;
    __shared__ int s_team;
    if (tid == 0) { s_edges = 0; s_changed = 0; s_pend = 0; s_team = 0; a.counts[3] = 0; a.counts[4] = 0; }
    if (a.team_state && tid < TEAM_MAX) { a.team_state[tid].gub = ~0ull; a.team_state[tid].done = 0; }
    __syncthreads();
    int E;
    unsigned long long* rows = const_cast<unsigned long long*>(a.assoc);
    if (a.edges_in) {
        // forest mode: grow_kernel already produced the deduplicated edge list, in EDGE_SEGS counted segments
        __shared__ int s_segoff[EDGE_SEGS + 1], s_long;
        if (tid < 64) {
            int n = my_n;
            if (n > a.seg_cap) { n = a.seg_cap; a.counts[3] = 1; }
            const int any_long = __any(n > CL_SPEC) ? 1 : 0;
            if (tid == 0) s_long = any_long;
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
        if (E > a.elds + a.Ecap) { if (tid == 0) a.counts[3] = 1; E = a.elds + a.Ecap; }
        {   // the speculative loads: entry l16 + 16 q of segment sg, if the segment is that long
            const int sg = tid >> 4, l16 = tid & 15;
            const int sb = s_segoff[sg], sn = s_segoff[sg + 1] - sb;
#pragma unroll
            for (int q = 0; q < CL_SPEC / 16; ++q) {
                const int idx = l16 + 16 * q, e = sb + idx;
                if (idx < sn && e < E) {
                    if (e < a.elds) eL[e] = spec[q];
                    else { a.edge_t[e - a.elds] = (int)(spec[q] >> 16); a.edge_m[e - a.elds] = (int)(spec[q] & 0xffff); }
                }
            }
        }
        // segments longer than CL_SPEC (rare): flat gather of the rest: thread -> dense edge index -> (segment, offset) by
        // binary search over the 65 offsets, four edges per thread and pass (the four global loads are issued before the
        // first result is stored: one round trip per pass)
        for (int e0 = tid; e0 < (s_long ? E : 0); e0 += 4 * CL_THREADS) {
            unsigned pk[4];
            bool rest[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * CL_THREADS;
                const int ec = e < E ? e : 0;
                int lo = 0, hi = EDGE_SEGS;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_segoff[mid] <= ec) lo = mid; else hi = mid; }
                pk[q] = a.edges_in[(size_t)lo * a.seg_cap + (ec - s_segoff[lo])];
                rest[q] = ec - s_segoff[lo] >= CL_SPEC;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * CL_THREADS;
                if (e >= E || !rest[q]) continue;
                if (e < a.elds) eL[e] = pk[q];
                else { a.edge_t[e - a.elds] = (int)(pk[q] >> 16); a.edge_m[e - a.elds] = (int)(pk[q] & 0xffff); }
            }
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
                    if (pos < a.elds) eL[pos] = ((unsigned)t << 16) | (unsigned)m;
                    else if (pos - a.elds < a.Ecap) { a.edge_t[pos - a.elds] = t; a.edge_m[pos - a.elds] = m; }
                    ++pos;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        E = s_edges;
        if (E > a.elds + a.Ecap) {
            if (tid == 0) a.counts[3] = 1;
            E = a.elds + a.Ecap;
        }
    }
    CL_STAMP(0);
    // Connected components.  Pass A: every measurement node learns its smallest user (one fire-and-forget LDS atomicMin per
    // edge; most nodes have a single user and are done).  Pass B: every other user of a node must end up in its owner's
    // component: those (owner, user) pairs are compacted into a list (ballot + one LDS atomic per wavefront) and united, one
    // pair per thread, by a lock-free union-find over the TARGETS only (hook the larger root under the smaller with
    // atomicCAS) -- the root of a component is its smallest target, the label the reference's scipy labelling + np.where
    // ordering implies (tracker.py:972-974).  Running the unions straight from the edge loop made every wavefront wait, in
    // every pass, for its few lanes on the slow path; compacted, the slow path runs once, fully occupied.
    int iters_done = 1;
    auto find_root = [&](int v) -> int {
        int p = tlabel[v];
        while (p != v) { v = p; p = tlabel[v]; }
        return v;
    };
    auto unite = [&](int x, int y) {
        int ra = find_root(x), rb = find_root(y);
        while (ra != rb) {
            if (ra > rb) { const int tmp = ra; ra = rb; rb = tmp; }      // ra < rb: hook rb under ra
            const int old = atomicCAS(&tlabel[rb], rb, ra);
            if (old == rb) break;
            rb = find_root(old);          // somebody else moved rb meanwhile: retry from its new root
            ra = find_root(ra);
        }
    };
    // (a thread keeps its first four edges -- all of them up to 4096 edges -- in registers across both passes: with sixteen
    // wavefronts on one LDS every access counts, not only the dependent ones)
    constexpr unsigned NO_EDGE = 0xffffffffu;
    unsigned pkr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + q * CL_THREADS;
        pkr[q] = e < E ? cl_edge(eL, a, e) : NO_EDGE;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (pkr[q] != NO_EDGE) atomicMin(&mlabel[pkr[q] & 0xffff], (int)(pkr[q] >> 16));
    for (int e = tid + 4 * CL_THREADS; e < E; e += CL_THREADS) {
        const unsigned pk = cl_edge(eL, a, e);
        atomicMin(&mlabel[pk & 0xffff], (int)(pk >> 16));
    }
    __syncthreads();
    if (tid == 0) s_edges = 0;      // (the stateless seam counted its edges here; from now on it counts the multi-target clusters)
    {
        // pass B for the edges in registers: one list reservation per wavefront for all four
        const int lane = tid & 63;
        int owner[4];
        unsigned long long fm[4];
        int total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) owner[q] = (pkr[q] != NO_EDGE) ? mlabel[pkr[q] & 0xffff] : -1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fm[q] = __ballot(pkr[q] != NO_EDGE && owner[q] != (int)(pkr[q] >> 16));
            total += __popcll(fm[q]);
        }
        if (a.edges_in && a.clear_rows) {      // clear the dedup bitsets for the next scan (forest mode), while the edge is at hand
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (pkr[q] != NO_EDGE) rows[(size_t)(pkr[q] >> 16) * a.AW + ((pkr[q] & 0xffff) >> 6)] = 0ull;
        }
        if (total) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_pend, total);
            base = __shfl(base, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if ((fm[q] >> lane) & 1ull) {
                    const int pos = base + __popcll(fm[q] & ((1ull << lane) - 1ull));
                    const int et = (int)(pkr[q] >> 16);
                    if (pos < a.pcap) pend[pos] = ((unsigned)owner[q] << 16) | (unsigned)et;
                    else unite(owner[q], et);          // list full (huge graphs): straight from here
                }
                base += __popcll(fm[q]);
            }
        }
    }
    for (int e0 = 4 * CL_THREADS; e0 < E; e0 += CL_THREADS) {      // more than 4096 edges: the rest, one at a time
        const int e = e0 + tid;
        int owner = -1, et = -1;
        if (e < E) {
            const unsigned pk = cl_edge(eL, a, e);
            et = (int)(pk >> 16);
            owner = mlabel[pk & 0xffff];
            if (a.edges_in && a.clear_rows) rows[(size_t)et * a.AW + ((pk & 0xffff) >> 6)] = 0ull;
        }
        const bool foreign = owner != et;      // owner < et
        const unsigned long long fm = __ballot(foreign);
        if (fm) {
            const int lane = tid & 63, leader = __ffsll((long long)fm) - 1;
            int pos = 0;
            if (lane == leader) pos = atomicAdd(&s_pend, __popcll(fm));
            pos = __shfl(pos, leader) + __popcll(fm & ((1ull << lane) - 1ull));
            if (foreign) {
                if (pos < a.pcap) pend[pos] = ((unsigned)owner << 16) | (unsigned)et;
                else unite(owner, et);
            }
        }
    }
    __syncthreads();
    {
        const int np = s_pend < a.pcap ? s_pend : a.pcap;
        for (int i = tid; i < np; i += CL_THREADS) {
            const unsigned pr = pend[i];
            unite((int)(pr >> 16), (int)(pr & 0xffff));
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += CL_THREADS) {
        const int r = find_root(t);
        lab[t] = r;                       // (writing tlabel here would race with other threads' find_root)
        atomicAdd(&cnt[r], 1);
    }
    if (a.edges_in) {      // hand the counters back (the dedup bitsets were cleared in pass B)
        if (tid < EDGE_SEGS) a.edge_count[tid] = 0;
        if (tid == 0 && a.ticket_reset) *a.ticket_reset = 0;      // grow_kernel's tile ticket (used when its grid is not co-resident)
        if (a.alloc_reset && tid >= 64 && tid < 64 + FG_REGIONS) a.alloc_reset[(tid - 64) * 32] = 0u;      // fgrow_kernel's child counters
    }
    __syncthreads();
    CL_STAMP(1);
    if (a.dbg && tid == 0) a.dbg[6] = iters_done;
    CL_STAMP(2);
    // heads -> cluster index (scan of the head flags) and cluster offset (scan of the head's member count), one pass
    int* cidx = mlabel;                  // [T] by head; the measurement parents are dead by now
    int* cstart = mlabel + a.Tcap;       // [T] by head
    int running = 0, running2 = 0;
    for (int base = 0; base < T; base += CL_THREADS) {
        const int t = base + tid;
        const int head = (t < T && lab[t] == t) ? 1 : 0;
        const int sz = head ? cnt[t] : 0;
        int incl = head, incl2 = sz;
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o), v2 = __shfl_up(incl2, o);
            if (lane >= o) { incl += v; incl2 += v2; }
        }
        if (lane == 63) { s_scan[wv] = incl; s_scan2[wv] = incl2; }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, acc2 = 0;
            for (int i = 0; i < CL_THREADS / 64; ++i) {
                const int v = s_scan[i], v2 = s_scan2[i];
                s_scan[i] = acc; s_scan2[i] = acc2;
                acc += v; acc2 += v2;
            }
            s_total = acc; s_total2 = acc2;
        }
        __syncthreads();
        if (head) {
            const int c = running + s_scan[wv] + incl - 1, p0 = running2 + s_scan2[wv] + incl2 - sz;
            cidx[t] = c;
            cstart[t] = p0;
            a.cl_ptr[c] = p0;
        }
        running += s_total;
        running2 += s_total2;
        __syncthreads();
    }
    const int nC = running;
    if (tid == 0) a.cl_ptr[nC] = running2;
    CL_STAMP(3);
    CL_STAMP(4);
    // work lists (heads of multi-target clusters -> blp_kernel's ILPs, targets alone in their cluster) and the member lists
    // in ascending target order: every member takes a slot of its cluster's segment with an LDS atomic (arbitrary order),
    // then finds its rank by counting the smaller members -- O(cluster size) per thread, all targets in parallel.
    // (the scratch lives in the LDS edge list, which is dead by now; Tcap <= elds is checked at launch)
    int* tmp = BIG ? mlabel + mslots : reinterpret_cast<int*>(eL);     // [T] every cluster owns the segment [cstart, cstart + cnt)
    for (int t = tid; t < T; t += CL_THREADS) {
        const int h = lab[t];
        const int c = cidx[h], base = cstart[h], K = cnt[h];
        a.t_label[t] = h;
        a.t_cluster[t] = c;
        if (K == 1) {
            a.cl_members[base] = t;
            a.single_list[atomicAdd(&s_changed, 1)] = t;
        } else {
            tmp[base + atomicAdd(&fill[h], 1)] = t;
            if (h == t) {
                a.multi_list[atomicAdd(&s_edges, 1)] = c;
                if (a.team_list && K >= TEAM_MIN_K) {      // large cluster: its branch and bound (if it needs one) is shared by a team
                    // slot = rank among the large clusters' heads: the same list on every device of a cluster-sharded step (the members of a team
                    // that spans devices file their results by slot), also when there are more than TEAM_MAX of them.  Rare: a loop over the heads below
                    int q = 0;
                    for (int u = 0; u < t; ++u) q += (lab[u] == u && cnt[u] >= TEAM_MIN_K) ? 1 : 0;
                    atomicAdd(&s_team, 1);
                    if (q < TEAM_MAX) a.team_list[q] = c;
                }
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += CL_THREADS) {
        const int h = lab[t];
        const int base = cstart[h], K = cnt[h];
        if (K == 1) continue;
        int rank = 0;
        for (int i = 0; i < K; ++i) rank += (tmp[base + i] < t) ? 1 : 0;
        a.cl_members[base + rank] = t;
    }
    CL_STAMP(5);
    if (a.dbg && tid == 0) a.dbg[7] = E;
    __syncthreads();
    if (a.cl_owner && a.shard_n > 1) {
        // ---- which device solves which ILP (cluster-sharded step): LPT on the column counts --------------------------------------
        // weight of a multi-target cluster = its columns (children of its members); rank by (weight descending, cluster index
        // ascending) by counting, then ONE thread deals them out in that order, each to the least loaded device.  The table is
        // indexed by cluster index, so it does not depend on the (atomic) order of multi_list.
        const int nM = s_edges;
        int* wgt = tmp;                 // [nM] (the member scratch is dead; nM <= Tcap / 2: a multi-target cluster has two members at least)
        int* cix = tmp + a.Tcap / 2;    // [nM]
        int* ord = tlabel;              // [nM] cluster list position by rank (the union-find parents are dead as well)
        for (int i = tid; i < nM; i += CL_THREADS) {
            const int c = a.multi_list[i];
            int w = 0;
            for (int k = a.cl_ptr[c]; k < a.cl_ptr[c + 1]; ++k) { const int t = a.cl_members[k]; w += a.tcend[t] - a.tchild[t]; }
            wgt[i] = w; cix[i] = c;
        }
        __syncthreads();
        for (int i = tid; i < nM; i += CL_THREADS) {
            const int w = wgt[i], c = cix[i];
            int rank = 0;
            for (int j = 0; j < nM; ++j) { const int wj = wgt[j], cj = cix[j]; rank += (wj > w || (wj == w && cj < c)) ? 1 : 0; }
            ord[rank] = i;
        }
        __syncthreads();
        if (tid == 0) {
            long long load[16];
            const int ns = a.shard_n < 16 ? a.shard_n : 16;
            for (int r = 0; r < ns; ++r) load[r] = 0;
            for (int q = 0; q < nM; ++q) {
                const int i = ord[q];
                int best = 0;
                for (int r = 1; r < ns; ++r) if (load[r] < load[best]) best = r;
                a.cl_owner[cix[i]] = best;
                load[best] += wgt[i];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.counts[0] = nC;
        a.counts[1] = s_edges;
        a.counts[2] = s_changed;
        if (a.team_list) a.counts[5] = s_team < TEAM_MAX ? s_team : TEAM_MAX;
    }
#undef __syncthreads
}

__global__ __launch_bounds__(CL_THREADS) void cluster_kernel(const ClusterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cluster_body<false>(a, smem);
}
__global__ __launch_bounds__(CL_THREADS) void cluster_big_kernel(const ClusterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cluster_body<true>(a, smem);
}
// Streaming API path: step 7 of the scan (the M-of-N initiator, tracker.py:264-278) needs nothing of steps 2-6 -- only which
// measurements the grow kernel gated -- so it runs as a SECOND workgroup of this launch, next to the clustering, instead of behind
// the ILPs in post_scan_kernel (10 us of the critical path of every streamed scan).  No hand-off inside the kernel: what it gives
// birth to is admitted by post_scan_kernel, two launches later.
__global__ __launch_bounds__(CL_THREADS) void cluster_init_kernel(const ClusterArgs a, const InitArgs in, const int32_t* sticky_overflow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x == 1) {
        if ((a.status && a.status->overflow) || (sticky_overflow && *sticky_overflow)) return;      // void scan: nothing is initiated
        initiator_body<false>(in);      // (a scan with messages for the initiator runs it behind the scan: mht_forest_scan)
        return;
    }
    cluster_body<false>(a, smem);
}
// a group of sectors per launch: blockIdx.y = sector, its argument block is read from HBM (two variants by scan parity)
__global__ __launch_bounds__(CL_THREADS) void cluster_batch_kernel(const PBatch av) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ClusterArgs a;
    load_args(a, static_cast<const ClusterArgs*>(av.p[blockIdx.y]));
    cluster_body<false>(a, smem);
}

// ---- tables beyond LDS: the union-find of the forest's grow launch, as two ordinary launches (r4) -------------------------------------
// cluster_big_kernel keeps the one-workgroup algorithm and moves its tables to HBM: correct at any size, but its phases become
// milliseconds of L2 atomics from ONE workgroup.  The stateless seam uses what the forest uses instead (mht_uf.h): a thread per word of
// the association bitsets exchanges the owner word of every measurement node in it and links the targets it meets there -- any number of
// workgroups, no edge records (so neither the 16 + 16 bits of one nor the 65 536-node limit apply) --, then a thread per target follows
// its parents to the smallest member of its component = the label.
__global__ __launch_bounds__(256) void uf_seam_hook_kernel(const unsigned long long* assoc, long long n_words, int AW, unsigned long long* owner, unsigned long long* parent) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long bits = assoc[i];
        if (!bits) continue;
        const int t = (int)(i / AW), w = (int)(i % AW);
        const unsigned long long mine = (1ull << 32) | (unsigned)t;
        while (bits) {
            const int b = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const unsigned long long old = atomicMax(&owner[(size_t)w * 64 + b], mine);
            if ((unsigned)(old >> 32) == 1u && (int)(unsigned)old != t) uf_link(parent, 1u, t, (int)(unsigned)old);
        }
    }
}
__global__ __launch_bounds__(256) void uf_seam_label_kernel(const unsigned long long* parent, int T, int32_t* label) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        int r = t;
        for (;;) {
            const unsigned long long w = parent[r];
            if ((unsigned)(w >> 32) != 1u) break;
            r = (int)(0xffffffffu - (unsigned)w);
        }
        label[t] = r;
    }
}

constexpr size_t CL_LDS_BUDGET = 150 * 1024;
// LDS carve for a given table size: the pending-pair list gets a quarter of what the tables leave (at most CL_PEND_MAX), the
// edge list the rest (at most CL_ELDS_MAX); elds < Tcap = does not fit
static void cluster_carve(int Tcap, int n_mnodes, int& elds, int& pcap) {
    const size_t mslots = n_mnodes > 2 * Tcap ? n_mnodes : 2 * Tcap;
    const size_t tables = ((size_t)4 * Tcap + mslots) * 4;
    elds = 0; pcap = 0;
    if (tables >= CL_LDS_BUDGET) return;
    const size_t room = (CL_LDS_BUDGET - tables) / 4;
    pcap = (int)(room / 4 < (size_t)CL_PEND_MAX ? room / 4 : (size_t)CL_PEND_MAX);
    elds = (int)(room - pcap < (size_t)CL_ELDS_MAX ? room - pcap : (size_t)CL_ELDS_MAX);
}
int cluster_elds(int Tcap, int n_mnodes) {
    int elds, pcap;
    cluster_carve(Tcap, n_mnodes, elds, pcap);
    return elds;
}
// the tables fit LDS (cluster_kernel), or they go to HBM scratch (cluster_big_kernel: gtab of cluster_big_ints() ints)
bool cluster_fits_lds(int Tcap, int n_mnodes) { const int e = cluster_elds(Tcap, n_mnodes); return e >= Tcap && e >= 1024; }
size_t cluster_big_ints(int Tcap, int n_mnodes) { return (size_t)5 * Tcap + (size_t)(n_mnodes > 2 * Tcap ? n_mnodes : 2 * Tcap); }
size_t cluster_lds_bytes(int Tcap, int n_mnodes) {
    const size_t mslots = n_mnodes > 2 * Tcap ? n_mnodes : 2 * Tcap;
    int elds, pcap;
    cluster_carve(Tcap, n_mnodes, elds, pcap);
    return ((size_t)4 * Tcap + mslots + (size_t)elds + (size_t)pcap) * 4;
}

int launch_cluster(mht_ctx* ctx, const ClusterArgs& a_in, const InitArgs* init, const int32_t* sticky_overflow) {
    ClusterArgs a = a_in;
    size_t& attr_bytes = ctx->lds_attr_cluster;
    if (a.n_mnodes > 65536 || a.Tcap > 65536) {
        set_error("cluster: %d targets / %d measurement nodes exceed the 16 + 16 bits of an edge record", a.Tcap, a.n_mnodes);
        return MHT_E_CAPACITY;
    }
    if (!cluster_fits_lds(a.Tcap, a.n_mnodes)) {      // tables in HBM scratch
        if (!a.gtab) {
            set_error("cluster: Tcap=%d and %d measurement nodes do not fit the clustering kernel's LDS budget (%zu KiB) and no HBM table was provided", a.Tcap,
                      a.n_mnodes, CL_LDS_BUDGET / 1024);
            return MHT_E_CAPACITY;
        }
        a.elds = CL_ELDS_MAX; a.pcap = CL_PEND_MAX;
        const size_t lds_big = (size_t)(a.elds + a.pcap) * 4;
        static size_t attr_big = 0;
        if (lds_big > 48 * 1024 && lds_big > attr_big) {
            MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
            attr_big = lds_big;
        }
        // (the initiator, if one was handed in, does not ride along: the caller runs it behind the scan, mht_forest.hip)
        hipLaunchKernelGGL(cluster_big_kernel, dim3(1), dim3(CL_THREADS), lds_big, ctx->stream, a);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
    cluster_carve(a.Tcap, a.n_mnodes, a.elds, a.pcap);
    const size_t lds = cluster_lds_bytes(a.Tcap, a.n_mnodes);
    if (lds > 48 * 1024 && lds > attr_bytes) {
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_init_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    if (init) hipLaunchKernelGGL(cluster_init_kernel, dim3(2), dim3(CL_THREADS), lds, ctx->stream, a, *init, sticky_overflow);
    else hipLaunchKernelGGL(cluster_kernel, dim3(1), dim3(CL_THREADS), lds, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

// the LDS carve (elds, pcap) of a forest's argument block, as launch_cluster sets it
void cluster_prepare(ClusterArgs& a) {
    if (cluster_fits_lds(a.Tcap, a.n_mnodes)) cluster_carve(a.Tcap, a.n_mnodes, a.elds, a.pcap);
    else { a.elds = CL_ELDS_MAX; a.pcap = CL_PEND_MAX; }
}

int launch_cluster_batch(mht_ctx* ctx, const PBatch& av, int n_sectors, int Tcap, int n_mnodes) {
    const size_t lds = cluster_lds_bytes(Tcap, n_mnodes);
    static size_t attr_bytes = 0;
    if (lds > 48 * 1024 && lds > attr_bytes) {
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    hipLaunchKernelGGL(cluster_batch_kernel, dim3(1, n_sectors), dim3(CL_THREADS), lds, ctx->stream, av);
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
    if (!cluster_fits_lds(T, words * 64)) {      // tables beyond LDS: the device-wide union-find (any size)
        const size_t nn = (size_t)words * 64;
        int rc = ctx->counts.ensure((nn + (size_t)T) * 8);
        if (rc) return rc;
        unsigned long long* owner = static_cast<unsigned long long*>(ctx->counts.ptr);
        unsigned long long* parent = owner + nn;
        MHT_HIP_CHECK(hipMemsetAsync(owner, 0, (nn + (size_t)T) * 8, ctx->stream));
        const long long n_words = (long long)T * words;
        long long g = (n_words + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(uf_seam_hook_kernel, dim3((unsigned)g), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned long long*>(assoc), n_words, (int)words, owner, parent);
        MHT_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(uf_seam_label_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, ctx->stream, static_cast<const unsigned long long*>(parent), (int)T, label);
        MHT_HIP_CHECK(hipGetLastError());
        MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return MHT_OK;
    }
    size_t ecap = (size_t)T * words * 64;
    if (ecap > (1u << 22)) ecap = 1u << 22;
    const bool big = !cluster_fits_lds(T, words * 64);
    const size_t ints = 2 * ecap + 6 * (size_t)T + 32 + (big ? cluster_big_ints(T, words * 64) : 0);
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
    if (big) a.gtab = q + 6 * T + 32;
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
