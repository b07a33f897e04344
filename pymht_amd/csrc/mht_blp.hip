// Global-hypothesis selection: per cluster the 0-1 ILP
//      min  sum_h f_h tau_h   s.t.  A1 tau <= 1 (a measurement in at most one selected leaf),
//                                   A2 tau  = 1 (exactly one leaf per target),  tau binary
// (reference: Tracker._solveOptimumAssociation / _solveBLP_OR_TOOLS, pymht/tracker.py:979-1027, :1155-1217, which
// builds dense A1/A2 with recursive Python + list.index and hands them to OR-Tools CBC), and for a target that is
// alone in its cluster Target._selectBestHypothesis (pymht/pyTarget.py:446-459).
//
// One workgroup per multi-target cluster.  Columns are the new leaf hypotheses (children) of the cluster's
// targets in DFS order; the rows a column touches are the measurement nodes on its root->leaf path (path[d][h]).
// Solver (all on the GPU):
//   1. Lagrangian relaxation of A1 with prices u >= 0 on measurement nodes; per target the minimiser of the
//      reduced cost f_h + sum u.  If the minimisers are conflict-free and every priced node is used exactly once
//      (complementary slackness) the primal cost equals the dual bound: CERTIFIED optimal.  Projected subgradient
//      steps (Polyak step length, upper bound from a greedy dive) move the prices otherwise.
//   2. If the certificate is not reached: depth-first branch and bound over the targets (hot ones first) with the
//      Lagrangian bound (valid for any u >= 0); the top levels re-optimise the prices of their residual problem and solve a
//      node outright when its minimisers certify.  Exact up to 1e-12 relative: BRANCHED.
// The reference's LP relaxation is integral in >99 % of instances, so step 2 is rare; it keeps
// the selection exact without any host fallback.
//
// Storage: a cluster whose columns / rows / members fit (<= 3072 columns, <= 1024 measurement nodes, <= 256 targets)
// is copied into LDS once (costs, rows as dense local ids, prices, usage counters, marks) and every dual step runs
// out of LDS; larger clusters run the same code on HBM scratch (L2 resident).
#include "mht_kernels.h"
#include "mht_commit.h"
#include <stdlib.h>

namespace mht {

#ifndef MHT_BLP_THREADS
#define MHT_BLP_THREADS 256
#endif
constexpr int BLP_THREADS = MHT_BLP_THREADS;
constexpr int FUSED_K = 8;           // clusters of up to this many targets: a wavefront per target for minimisers / usage / regrets
constexpr double DINF = 1.0e300;
constexpr int BIG_MAXH = 2048, BIG_MAXR = 1024, BIG_MAXK = 256;      // default LDS tier: columns, rows, targets of a cluster solved out of LDS
// (member tables hold cap_k + 4 entries: (cap_k + 4) * 4 and * 8 are multiples of 16 bytes when cap_k is a multiple of 4)

struct Red {
    double d[BLP_THREADS / 64];
    double q[BLP_THREADS / 64][4];
    int i[BLP_THREADS / 64];
};

// wave64 sum with DPP row shifts / row broadcasts (a handful of VALU ops instead of 12 LDS-crossbar permutes);
// fixed summation order -> deterministic.  Result is valid in every lane.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0x111, 0xf>(v);      // row_shr:1
    v += dpp_f64<0x112, 0xf>(v);      // row_shr:2
    v += dpp_f64<0x114, 0xf>(v);      // row_shr:4
    v += dpp_f64<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds the row sum
    v += dpp_f64<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ double block_sum(double v, Red* r) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) r->d[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BLP_THREADS / 64; ++w) s += r->d[w];
    return s;
}
__device__ __forceinline__ int block_or(int v, Red* r) {
    v = __any(v) ? 1 : 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) r->i[threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < BLP_THREADS / 64; ++w) s |= r->i[w];
    return s;
}
// minimum of a value over a 16-lane row (result in lane 15 of the row) / the wavefront (result in every lane): three VALU
// ops per step where the (value, index) pair needs a dozen
template <int CTRL, int ROW_MASK> __device__ __forceinline__ void dpp_min_value(double& v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int nlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int nhi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    v = fmin(v, __hiloint2double(nhi, nlo));
}
__device__ __forceinline__ double row16_min_value(double v) {
    dpp_min_value<0x111, 0xf>(v);
    dpp_min_value<0x112, 0xf>(v);
    dpp_min_value<0x114, 0xf>(v);
    dpp_min_value<0x118, 0xf>(v);
    return v;
}
__device__ __forceinline__ double wave_min_value(double v) {
    v = row16_min_value(v);
    dpp_min_value<0x142, 0xa>(v);      // row_bcast:15
    dpp_min_value<0x143, 0xc>(v);      // row_bcast:31 -> lane 63 holds the minimum
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// lexicographic (value, index) minimum over the wavefront with DPP (index -1 = none); result valid in every lane
template <int CTRL, int ROW_MASK> __device__ __forceinline__ void dpp_min_pair(double& v, int& i) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int nlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int nhi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const int ni = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false);
    const double nv = __hiloint2double(nhi, nlo);
    if (ni >= 0 && (i < 0 || nv < v || (nv == v && ni < i))) { v = nv; i = ni; }
}
__device__ __forceinline__ void row16_min_pair(double& v, int& i) {      // lane 15 of every 16-lane row gets the row minimum
    dpp_min_pair<0x111, 0xf>(v, i);
    dpp_min_pair<0x112, 0xf>(v, i);
    dpp_min_pair<0x114, 0xf>(v, i);
    dpp_min_pair<0x118, 0xf>(v, i);
}
__device__ __forceinline__ void wave_min_pair(double& v, int& i) {
    row16_min_pair(v, i);
    dpp_min_pair<0x142, 0xa>(v, i);      // row_bcast:15
    dpp_min_pair<0x143, 0xc>(v, i);      // row_bcast:31 -> lane 63 holds the minimum
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63), lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    v = __hiloint2double(hi, lo);
    i = __builtin_amdgcn_readlane(i, 63);
}
__device__ __forceinline__ void block_min_pair(double& v, int& i, Red* r) {
    wave_min_pair(v, i);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { r->d[threadIdx.x >> 6] = v; r->i[threadIdx.x >> 6] = i; }
    __syncthreads();
    v = r->d[0];
    i = r->i[0];
#pragma unroll
    for (int w = 1; w < BLP_THREADS / 64; ++w) {
        const double ov = r->d[w];
        const int oi = r->i[w];
        if (oi >= 0 && (i < 0 || ov < v || (ov == v && oi < i))) { v = ov; i = oi; }
    }
}

// ---- storage policies -----------------------------------------------------------------------------------------
// Columns are addressed by a policy-local index; rows by a policy-local id in [0, nrows()).
#ifndef MHT_GS_CK
#define MHT_GS_CK 1
#endif
struct GStore {      // HBM: column = global child index, row = global measurement-node id, row set = LDS bitset
    const BlpArgs* a; const int32_t* mem; const unsigned long long* uw; int UW, PD; size_t cap;
    double* pu; int32_t* pusage; int32_t* pmark;      // prices / usage / marks by measurement node: the workgroup's own copy (a team member's, or the shared one)
    int32_t *best_h, *ub_sel, *ch, *lix; double *best_rc, *cst, *uus, *lrc, *rest, *mn;
    // (r5) ck: the members' column ranges are cbL[k] .. ceL[k] (LDS, filled by solve_cluster for clusters of <= cap_k targets) instead of two
    // dependent global look-ups (member -> target -> range) in front of every member's sweep.
    // (Tried and backed out: the rows' prices / usage counters / marks in the LDS solver's row tables, addressed by the node's dense rank in
    // the row bitset -- the rank costs two LDS reads and a 64-bit popcount per look-up and the tables become generic pointers (flat
    // accesses): G20 131 -> 155 ms, profiles/r05_ilp_tail.txt.)
    bool ck; const int32_t *cbL, *ceL;
    __device__ __forceinline__ int row(int m) const { return m; }
    __device__ __forceinline__ int col_begin(int k) const { return (MHT_GS_CK && ck) ? cbL[k] : a->tchild[mem[k]]; }
    __device__ __forceinline__ int col_end(int k) const { return (MHT_GS_CK && ck) ? ceL[k] : a->tcend[mem[k]]; }
    __device__ __forceinline__ double cost(int h) const { return a->cost[h]; }
    __device__ __forceinline__ int ent(int d, int h) const { return a->pds ? a->path[(size_t)h * a->pds + d] : a->path[(size_t)d * cap + h]; }
    __device__ __forceinline__ double& u(int m) const { return pu[row(m)]; }
    __device__ __forceinline__ int32_t& usage(int m) const { return pusage[row(m)]; }
    __device__ __forceinline__ int32_t& mark(int m) const { return pmark[row(m)]; }
    __device__ __forceinline__ int to_global(int h) const { return h; }
    // The rows' prices / usage counters live in HBM scratch: every row a thread visits is a round trip to the L2 of its own (the sums that
    // are formed over them keep the loads from overlapping).  A thread per WORD of the bitset walks up to 64 rows one after the other -- the
    // 188 rows of G20 sit in three words: three threads, 64 round trips each, four times per subgradient step (2/3 of its 90 us).  Up to
    // FOR_ROWS_FLAT words a thread per BIT: at most UW / 4 looks at the bitset (LDS) and ~nR / 256 rows per thread.
    static constexpr int FOR_ROWS_FLAT = 256;
    template <typename F> __device__ __forceinline__ void for_rows(F f) const {
        if (UW <= FOR_ROWS_FLAT) {
            for (int m = threadIdx.x; m < UW * 64; m += BLP_THREADS)
                if ((uw[m >> 6] >> (m & 63)) & 1ull) f(m);
            return;
        }
        for (int w = threadIdx.x; w < UW; w += BLP_THREADS) {
            unsigned long long bits = uw[w];
            while (bits) {
                const int m = w * 64 + __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                f(m);
            }
        }
    }
};
struct EnumEnt;
struct LStore {      // LDS: column = dense local index, row = dense local id
    double* costL; unsigned short* entL; double* uL; int32_t* usageL; int32_t* markL; int32_t* colb; int32_t* gbase;
    int32_t* gcolL;            // [cap_h] global column of every LDS column (reduced clusters); `reduced` says whether it is in use
    bool reduced;
    double* rcL; unsigned short* membL; unsigned long long* minkey;   // per-column reduced cost / member, per-member minimum key
    unsigned short* ordL;      // enumerate_small: column of every ranked entry
    unsigned short* enumL;     // enumerate_small: search state of the wavefronts (ENUM_LDS bytes)
    EnumEnt* xL;               // enumerate_small: [cap_h] reduced cost + signature of every column
    int nH, nR, PD, K;
    int32_t *best_h, *ub_sel, *ch, *lix; double *best_rc, *cst, *uus, *lrc, *rest, *mn;
    __device__ __forceinline__ int col_begin(int k) const { return colb[k]; }
    __device__ __forceinline__ int col_end(int k) const { return colb[k + 1]; }
    __device__ __forceinline__ double cost(int h) const { return costL[h]; }
    __device__ __forceinline__ int ent(int d, int h) const { const int e = entL[h * 8 + d]; return e == nR ? -1 : e; }
    __device__ __forceinline__ double& u(int m) const { return uL[m]; }
    __device__ __forceinline__ int32_t& usage(int m) const { return usageL[m]; }
    __device__ __forceinline__ int32_t& mark(int m) const { return markL[m]; }
    __device__ __forceinline__ int to_global(int h) const {       // member k with colb[k] <= h
        if (reduced) return gcolL[h];
        int lo = 0, hi = K;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (colb[mid] <= h) lo = mid; else hi = mid; }
        return gbase[lo] + (h - colb[lo]);
    }
    template <typename F> __device__ __forceinline__ void for_rows(F f) const {
        for (int m = threadIdx.x; m < nR; m += BLP_THREADS) f(m);
    }
};

// ---- column operations, generic (HBM policy) and specialised (LDS policy: one 16-byte record per column holds its
//      <= 8 rows as dense ids, "no row" = the dummy row nR whose price / mark / usage are never non-zero, so all
//      eight look-ups are unconditional and independent: two LDS round trips instead of 2*PD dependent ones) ----------
// The rows of a column with at most 8 path levels, fetched as ONE batch of independent loads (a forest record is two 16-byte
// pieces; the stateless layout is a strided gather), so that the prices / marks behind them are a second batch: two dependent
// round trips per column instead of 2 * PD -- a sweep over the 18 k columns of a giant cluster is nothing but these look-ups.
#ifndef MHT_ROWS8
#define MHT_ROWS8 1
#endif
__device__ __forceinline__ void rows8(const GStore& s, int h, int (&e)[8]) {
    if (s.a->pds == 8) {
        const int4* p = reinterpret_cast<const int4*>(s.a->path + (size_t)h * 8);
        const int4 q0 = p[0], q1 = p[1];
        e[0] = q0.x; e[1] = q0.y; e[2] = q0.z; e[3] = q0.w; e[4] = q1.x; e[5] = q1.y; e[6] = q1.z; e[7] = q1.w;
    } else {
#pragma unroll
        for (int d = 0; d < 8; ++d) e[d] = (d < s.PD) ? s.ent(d, h) : -1;
    }
}
__device__ __forceinline__ double reduced_cost(const GStore& s, int h) {
    double rc = s.cost(h);
    if (s.PD <= 8 && MHT_ROWS8) {
        int e[8];
        rows8(s, h, e);
        double uv[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) uv[d] = s.u(e[d] >= 0 ? e[d] : 0);
#pragma unroll
        for (int d = 0; d < 8; ++d) rc = (e[d] >= 0) ? rc + uv[d] : rc;      // (same order of additions as the loop below)
        return rc;
    }
    for (int d = 0; d < s.PD; ++d) {
        const int e = s.ent(d, h);
        if (e >= 0) rc += s.u(e);
    }
    return rc;
}
__device__ __forceinline__ bool compatible(const GStore& s, int h) {
    if (s.PD <= 8 && MHT_ROWS8) {
        int e[8];
        rows8(s, h, e);
        int bad = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) bad |= (e[d] >= 0) ? s.mark(e[d] >= 0 ? e[d] : 0) : 0;
        return bad == 0;
    }
    for (int d = 0; d < s.PD; ++d) {
        const int e = s.ent(d, h);
        if (e >= 0 && s.mark(e)) return false;
    }
    return true;
}
__device__ __forceinline__ void set_marks(const GStore& s, int h, int value) {
    if ((int)threadIdx.x < s.PD) {
        const int e = s.ent(threadIdx.x, h);
        if (e >= 0) s.mark(e) = value;
    }
    __threadfence_block();
    __syncthreads();
}
__device__ __forceinline__ double priced_part(const GStore& s, int h) {      // this thread's share of sum u over column h
    if ((int)threadIdx.x < s.PD) {
        const int e = s.ent(threadIdx.x, h);
        if (e >= 0) return s.u(e);
    }
    return 0.0;
}
__device__ __forceinline__ void add_usage(const GStore& s, int k, int d) {
    const int e = s.ent(d, s.best_h[k]);
    if (e >= 0) atomicAdd(&s.usage(e), 1);
}

struct Rows8 { unsigned short e[8]; };
__device__ __forceinline__ Rows8 rows_of(const LStore& s, int h) {
    const uint4 v = reinterpret_cast<const uint4*>(s.entL)[h];
    Rows8 r;
    r.e[0] = v.x & 0xffff; r.e[1] = v.x >> 16; r.e[2] = v.y & 0xffff; r.e[3] = v.y >> 16;
    r.e[4] = v.z & 0xffff; r.e[5] = v.z >> 16; r.e[6] = v.w & 0xffff; r.e[7] = v.w >> 16;
    return r;
}
__device__ __forceinline__ double reduced_cost(const LStore& s, int h) {
    const Rows8 r = rows_of(s, h);
    const double c = s.costL[h];
    const double u0 = s.uL[r.e[0]], u1 = s.uL[r.e[1]], u2 = s.uL[r.e[2]], u3 = s.uL[r.e[3]];
    const double u4 = s.uL[r.e[4]], u5 = s.uL[r.e[5]], u6 = s.uL[r.e[6]], u7 = s.uL[r.e[7]];
    return ((((((((c + u0) + u1) + u2) + u3) + u4) + u5) + u6) + u7);
}
__device__ __forceinline__ bool compatible(const LStore& s, int h) {
    const Rows8 r = rows_of(s, h);
    const int m = s.markL[r.e[0]] | s.markL[r.e[1]] | s.markL[r.e[2]] | s.markL[r.e[3]] | s.markL[r.e[4]] | s.markL[r.e[5]] |
                  s.markL[r.e[6]] | s.markL[r.e[7]];
    return m == 0;
}
__device__ __forceinline__ void set_marks(const LStore& s, int h, int value) {
    if (threadIdx.x < 8) {
        const int e = s.entL[h * 8 + threadIdx.x];
        if (e != s.nR) s.markL[e] = value;
    }
    __syncthreads();
}
__device__ __forceinline__ double priced_part(const LStore& s, int h) {
    return threadIdx.x < 8 ? s.uL[s.entL[h * 8 + threadIdx.x]] : 0.0;
}
__device__ __forceinline__ void add_usage(const LStore& s, int k, int d) {
    const int e = s.entL[s.best_h[k] * 8 + d];
    if (e != s.nR) atomicAdd(&s.usageL[e], 1);
}
template <typename S> __device__ __forceinline__ double priced_sum(const S& s, int h, Red* r) {
    return block_sum(priced_part(s, h), r);
}

// One wavefront's sweep over the columns [hb, he) of a member on HBM scratch: this lane's (reduced cost, lowest index) minimum among the
// columns compatible with the marks (COMPAT) -- SWEEP_U columns in flight per lane.  A column is two dependent round trips to the L2 (its
// rows, then their prices / marks); a member of a giant cluster has a few hundred columns, so a lane that takes them one at a time pays
// those round trips three to five times per member and a sweep over 44 members (11 per wavefront) is ~80 us of nothing but latency -- the
// time of every subgradient step and of every node of the branch and bound on HBM scratch (G20: 0.4 ms per node).  The ILP workgroups
// are one per CU (LDS), i.e. one wavefront per SIMD: the registers the batch needs are there for the taking.  The columns of a lane are
// still visited in ascending order and compared with strict <, every reduced cost is summed in the same order: same minimisers, bit for bit.
// (SWEEP_U measured on G20 with the next member's rows in flight, profiles/r05_ilp_tail.txt: 1: 111 ms, 2: 111, 3: 119, 4: 118, 8: 170 -- once the members are
// pipelined the batch size buys nothing and its registers cost: the batch lives twice, this member's and the next one's)
#ifndef MHT_SWEEP_U
#define MHT_SWEEP_U 2
#endif
constexpr int SWEEP_U = MHT_SWEEP_U;
struct RowBatch { int e[SWEEP_U][8]; double c[SWEEP_U]; };
// rows and costs of the columns h0, h0 + 64, ... of a member [hb, he) -- loads only, nothing waits for them here
__device__ __forceinline__ void rows_load(const GStore& s, int hb, int h0, int he, RowBatch& b) {
#pragma unroll
    for (int q = 0; q < SWEEP_U; ++q) {
        const int h = h0 + 64 * q, hc = h < he ? h : hb;      // (clamped: every member has a column)
        rows8(s, hc, b.e[q]);
        b.c[q] = s.cost(hc);
    }
}
// ... their prices (and marks), the reduced costs, this lane's running minimum
template <bool COMPAT> __device__ __forceinline__ void rows_finish(const GStore& s, const RowBatch& b, int h0, int he, double& bv, int& bi) {
    double uv[SWEEP_U][8];
    int mk[SWEEP_U], mv[SWEEP_U][8];
    // (levels d >= PD hold no row in any column -- a uniform test: no look-up is issued for them; within PD every look-up is unconditional,
    // node 0 standing in for "no row": the sweep is bound by the number of gathers it issues)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        if (d < s.PD) {
#pragma unroll
            for (int q = 0; q < SWEEP_U; ++q) {
                const int m = b.e[q][d] >= 0 ? b.e[q][d] : 0;
                uv[q][d] = s.pu[m];
                if (COMPAT) mv[q][d] = s.pmark[m];
            }
        } else {
#pragma unroll
            for (int q = 0; q < SWEEP_U; ++q) { uv[q][d] = 0.0; mv[q][d] = 0; }
        }
    }
#pragma unroll
    for (int q = 0; q < SWEEP_U; ++q) {
        mk[q] = 0;
        if (COMPAT) {
#pragma unroll
            for (int d = 0; d < 8; ++d) mk[q] |= (b.e[q][d] >= 0) ? mv[q][d] : 0;
        }
    }
#pragma unroll
    for (int q = 0; q < SWEEP_U; ++q) {
        const int h = h0 + 64 * q;
        double rc = b.c[q];
#pragma unroll
        for (int d = 0; d < 8; ++d) rc = (b.e[q][d] >= 0) ? rc + uv[q][d] : rc;      // (reduced_cost()'s order of additions)
        if (h < he && mk[q] == 0 && (bi < 0 || rc < bv)) { bv = rc; bi = h; }
    }
}
// The wavefront's members one after the other: member(i) -> k, out(i, value, column) with the (reduced cost, lowest index) minimum of member
// i among its columns (COMPAT: those compatible with the marks), uniform over the wavefront.  The rows of member i + 1 are on their way
// while member i's prices are: one round trip per member instead of two.
template <bool COMPAT, typename M, typename O> __device__ __forceinline__ void sweep_wave(const GStore& s, int n, int lane, M member, O out) {
    if (!(s.PD <= 8 && MHT_ROWS8)) {
        for (int i = 0; i < n; ++i) {
            const int k = member(i);
            double bv = DINF;
            int bi = -1;
            for (int h = s.col_begin(k) + lane; h < s.col_end(k); h += 64) {
                if (COMPAT && !compatible(s, h)) continue;
                const double rc = reduced_cost(s, h);
                if (bi < 0 || rc < bv) { bv = rc; bi = h; }
            }
            wave_min_pair(bv, bi);
            out(i, bv, bi);
        }
        return;
    }
    RowBatch cur, nxt;
    int hb = 0, he = 0, nhb = 0, nhe = 0;
    if (n > 0) {
        const int k = member(0);
        hb = s.col_begin(k); he = s.col_end(k);
        rows_load(s, hb, hb + lane, he, cur);
    }
    for (int i = 0; i < n; ++i) {
        if (i + 1 < n) {
            const int k1 = member(i + 1);
            nhb = s.col_begin(k1); nhe = s.col_end(k1);
            rows_load(s, nhb, nhb + lane, nhe, nxt);
        }
        double bv = DINF;
        int bi = -1;
        rows_finish<COMPAT>(s, cur, hb + lane, he, bv, bi);
        for (int h0 = hb + lane + 64 * SWEEP_U; h0 < he; h0 += 64 * SWEEP_U) {      // a member of more than 64 * SWEEP_U columns
            RowBatch t;
            rows_load(s, hb, h0, he, t);
            rows_finish<COMPAT>(s, t, h0, he, bv, bi);
        }
        wave_min_pair(bv, bi);
        out(i, bv, bi);
        cur = nxt; hb = nhb; he = nhe;
    }
}
template <bool COMPAT, typename M, typename O> __device__ __forceinline__ void sweep_wave(const LStore& s, int n, int lane, M member, O out) {
    for (int i = 0; i < n; ++i) {
        const int k = member(i);
        double bv = DINF;
        int bi = -1;
        for (int h = s.col_begin(k) + lane; h < s.col_end(k); h += 64) {
            if (COMPAT && !compatible(s, h)) continue;
            const double rc = reduced_cost(s, h);
            if (bi < 0 || rc < bv) { bv = rc; bi = h; }
        }
        wave_min_pair(bv, bi);
        out(i, bv, bi);
    }
}

// per target the minimiser of the reduced cost (lowest column wins ties) -> best_h[k], best_rc[k]
__device__ __forceinline__ void compute_minimisers(const GStore& s, int K, Red* r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = BLP_THREADS / 64;
    sweep_wave<false>(s, K > wave ? (K - wave + NW - 1) / NW : 0, lane, [&](int i) { return wave + NW * i; },
                      [&](int i, double bv, int bi) { if (lane == 0) { s.best_h[wave + NW * i] = bi; s.best_rc[wave + NW * i] = bv; } });
    __threadfence_block();
    __syncthreads();
}
__device__ __forceinline__ void compute_minimisers(const LStore& s, int K, Red* r) {
    if (K <= FUSED_K) {
        // small cluster: one wavefront per target (two in turn from five targets on), reduced costs computed in the same sweep (kept
        // in rcL for the coordinate step), DPP reduction: a single phase
        const int lane = threadIdx.x & 63;
        for (int k = threadIdx.x >> 6; k < K; k += BLP_THREADS / 64) {
            const int hb = s.colb[k], he = s.colb[k + 1];
            double bv = DINF;
            int bi = -1;
            for (int h = hb + lane; h < he; h += 128) {      // two columns in flight per lane
                const int h1 = h + 64, h1c = h1 < he ? h1 : h;
                const double rc0 = reduced_cost(s, h), rc1 = reduced_cost(s, h1c);
                s.rcL[h] = rc0;
                if (bi < 0 || rc0 < bv) { bv = rc0; bi = h; }
                if (h1 < he) {
                    s.rcL[h1] = rc1;
                    if (rc1 < bv) { bv = rc1; bi = h1; }
                }
            }
            {   // (value, lowest index) minimum: the value by DPP, the index from the lanes that hold it (one lane unless
                // costs tie exactly)
                const double gmin = wave_min_value(bi >= 0 ? bv : DINF);
                unsigned long long mk = __ballot(bi >= 0 && bv == gmin);
                int gi = -1;
                while (mk) {
                    const int src = __ffsll((long long)mk) - 1;
                    const int cand = __builtin_amdgcn_readlane(bi, src);
                    if (gi < 0 || cand < gi) gi = cand;
                    mk &= mk - 1;
                }
                bv = gmin;
                bi = gi;
            }
            if (lane == 0) { s.best_h[k] = bi; s.best_rc[k] = bv; }
            if (lane < 8 && bi >= 0) {                      // usage of the minimiser's rows right here: no phase of its own
                const int e = s.entL[bi * 8 + lane];
                if (e != s.nR) atomicAdd(&s.usageL[e], 1);
            }
        }
        __syncthreads();
        return;
    }
    // pass 1: one thread per column -> reduced cost in LDS; pass 2: 16 lanes per target scan its columns and reduce
    // with four DPP row shifts (no atomics, no LDS-crossbar permutes)
    for (int h = threadIdx.x; h < s.nH; h += BLP_THREADS) s.rcL[h] = reduced_cost(s, h);
    __syncthreads();
    const int row = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    for (int k = row; k < K; k += BLP_THREADS / 16) {
        double bv = DINF;
        int bi = -1;
        const int hb = s.colb[k], he = s.colb[k + 1];
#pragma unroll 2
        for (int h = hb + l16; h < he; h += 16) {
            const double rc = s.rcL[h];
            if (bi < 0 || rc < bv) { bv = rc; bi = h; }
        }
        row16_min_pair(bv, bi);
        if (l16 == 15) { s.best_h[k] = bi; s.best_rc[k] = bv; }
    }
    __syncthreads();
}

// cheapest column of member k among those compatible with the marks (if need_compat) and lexicographically after
// (prc, pix); returns (DINF, -1) if none.  Uniform result in every thread.
__device__ __forceinline__ void argmin_member(const GStore& s, int k, bool need_compat, double prc, int pix, double& bv, int& bi, Red* r) {
    bv = DINF;
    bi = -1;
    for (int h = s.col_begin(k) + threadIdx.x; h < s.col_end(k); h += BLP_THREADS) {
        if (need_compat && !compatible(s, h)) continue;
        const double rc = reduced_cost(s, h);
        if (rc < prc || (rc == prc && h <= pix)) continue;
        if (bi < 0 || rc < bv) { bv = rc; bi = h; }
    }
    block_min_pair(bv, bi, r);
}
__device__ __forceinline__ void argmin_member(const LStore& s, int k, bool need_compat, double prc, int pix, double& bv, int& bi, Red* r) {
    bv = DINF;
    bi = -1;
    for (int h = s.col_begin(k) + threadIdx.x; h < s.col_end(k); h += BLP_THREADS) {
        if (need_compat && !compatible(s, h)) continue;
        const double rc = reduced_cost(s, h);
        if (rc < prc || (rc == prc && h <= pix)) continue;
        if (bi < 0 || rc < bv) { bv = rc; bi = h; }
    }
    block_min_pair(bv, bi, r);
}

// ---- dual coordinate ascent (LDS policy) -------------------------------------------------------------------------
// The conflicts of MHT clusters are mostly local: a few targets whose cheapest leaves want the same measurement node.
// Along the single price u_m the dual function is piecewise linear and its maximiser is known in closed form: with the
// users' regrets r_t = (cheapest column of t avoiding m) - (cheapest column of t) sorted r1 >= r2 >= ..., any increase
// in (r2, r1) makes every user but the highest bidder leave m strictly -- an auction step.  Rows are repriced together
// only when no target takes part in two of them (every target nominates the lowest conflicted row of its minimiser; a
// row is active iff all its users nominated it), so the step is a block-coordinate ascent: the dual bound never
// decreases.  Priced rows nobody uses any more are lowered to just below the cheapest taker.  Nothing here affects
// exactness: the certificate (no conflict, no priced-but-unused row) is what proves optimality, and clusters that are
// not certified after CA_ROUNDS go to the branch and bound.
#ifndef MHT_BB_NODE_STEPS
#define MHT_BB_NODE_STEPS 4
#endif
constexpr int BB_NODE_STEPS = MHT_BB_NODE_STEPS;   // subgradient steps per such node
constexpr int CA_ROUNDS = 16;      // coordinate rounds before the branch and bound takes over
constexpr int CA_ROUNDS_PAIR = 6;  // ... for two-target clusters: their branch and bound is ~5 nodes, cheaper than more rounds
__device__ __forceinline__ bool scratch_is_private(const GStore&) { return false; }
__device__ __forceinline__ bool scratch_is_private(const LStore&) { return true; }
__device__ __forceinline__ bool usage_counted_by_minimisers(const GStore&, int) { return false; }
__device__ __forceinline__ bool usage_counted_by_minimisers(const LStore&, int K) { return K <= FUSED_K; }
// Branch and bound right after the coordinate rounds only where it is cheap and cannot explode: clusters of <= 4 targets.
// Larger clusters go on with subgradient steps (up to max_iter) first: their branch and bound needs good prices (a 68-target
// scenario ran into the node limit with the prices of 16 coordinate rounds).
__device__ __forceinline__ int coordinate_rounds(const GStore&, int) { return 0; }       // giant clusters: subgradient steps only
__device__ __forceinline__ int coordinate_rounds(const LStore&, int K) { return K == 2 ? CA_ROUNDS_PAIR : CA_ROUNDS; }
__device__ __forceinline__ bool bb_after_rounds(const GStore&, int) { return false; }
__device__ __forceinline__ bool bb_after_rounds(const LStore&, int K) { return K <= 4; }
// HBM policy (giant clusters): no coordinate rounds (coordinate_rounds() is 0 there), these are never executed
__device__ __forceinline__ void nominate(const GStore&, int) {}
__device__ __forceinline__ bool coordinate_step(const GStore&, int, bool, bool) { return false; }
// every target nominates the lowest conflicted row of its minimiser (lix = row or -1; markL counts nominations).
// Runs in the same phase as the certificate flags (both only need the usage counters).
__device__ __forceinline__ void nominate(const LStore& s, int K) {
    for (int k = threadIdx.x; k < K; k += BLP_THREADS) {
        const Rows8 e8 = rows_of(s, s.best_h[k]);
        int us[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) us[d] = s.usageL[e8.e[d]];      // dummy row nR: usage 0
        int act = 0x7fffffff;
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (us[d] >= 2 && (int)e8.e[d] < act) act = e8.e[d];
        s.lix[k] = (act == 0x7fffffff) ? -1 : act;
        if (act != 0x7fffffff) atomicAdd(&s.markL[act], 1);
    }
}
// cheapest column of target k that avoids row m, by `G` cooperating lanes (G = 16: DPP row; G = 64: wavefront)
template <int G> __device__ __forceinline__ double regret_of(const LStore& s, int k, int m, bool active, int l) {
    double alt = DINF;      // only the value matters: DINF = no column avoids m
    if (active) {
        const int hb = s.colb[k], he = s.colb[k + 1];
        const unsigned mm = (unsigned)m, m2 = mm | (mm << 16);
        auto has_row = [&](const uint4& v) -> bool {      // does any of the eight 16-bit row ids equal m?
            const unsigned x0 = v.x ^ m2, x1 = v.y ^ m2, x2 = v.z ^ m2, x3 = v.w ^ m2;
            return !(x0 & 0xffffu) || !(x0 >> 16) || !(x1 & 0xffffu) || !(x1 >> 16) ||
                   !(x2 & 0xffffu) || !(x2 >> 16) || !(x3 & 0xffffu) || !(x3 >> 16);
        };
        for (int h = hb + l; h < he; h += 2 * G) {      // two columns in flight per lane
            const int h1 = h + G, h1c = h1 < he ? h1 : h;
            const uint4 v0 = reinterpret_cast<const uint4*>(s.entL)[h], v1 = reinterpret_cast<const uint4*>(s.entL)[h1c];
            const double rc0 = s.rcL[h], rc1 = s.rcL[h1c];
            if (!has_row(v0)) alt = fmin(alt, rc0);
            if (h1 < he && !has_row(v1)) alt = fmin(alt, rc1);
        }
    }
    return (G == 64) ? wave_min_value(alt) : row16_min_value(alt);
}
__device__ __forceinline__ bool coordinate_step(const LStore& s, int K, bool conflict, bool slack) {
    const int tid = threadIdx.x;
    if (conflict) {
        // regrets of the users of active rows (mn[k] = regret, -1 = target not taking part)
        if (K <= FUSED_K) {
            const int lane = tid & 63;
            for (int k = tid >> 6; k < K; k += BLP_THREADS / 64) {
                const int m = s.lix[k];
                const bool active = m >= 0 && s.markL[m] == s.usageL[m];
                const double alt = regret_of<64>(s, k, m, active, lane);
                if (lane == 0) s.mn[k] = active ? alt - s.best_rc[k] : -1.0;
            }
        } else {
            const int row = tid >> 4, l16 = tid & 15;
            for (int k0 = 0; k0 < K; k0 += BLP_THREADS / 16) {
                const int k = k0 + row;
                const int m = (k < K) ? s.lix[k] : -1;
                const bool active = m >= 0 && s.markL[m] == s.usageL[m];
                const double alt = regret_of<16>(s, k, m, active, l16);
                if (l16 == 15 && k < K) s.mn[k] = active ? alt - s.best_rc[k] : -1.0;
            }
        }
        __syncthreads();
        // price increase of every active row, written by its lowest-index user
        for (int k = tid; k < K; k += BLP_THREADS) {
            const int m = s.lix[k];
            if (m < 0 || s.mn[k] < 0.0) continue;
            double r1 = -1.0, r2 = -1.0;
            bool lowest = true;
            for (int j = 0; j < K; ++j) {
                const int mj = s.lix[j];
                const double v = s.mn[j];
                if (mj == m) {
                    if (j < k) lowest = false;
                    if (v > r1) { r2 = r1; r1 = v; }
                    else if (v > r2) r2 = v;
                }
            }
            if (lowest && r2 >= 0.0 && r2 < DINF) s.uL[m] += r2 + 0.5 * fmin(r1 - r2, 1.0);
        }
        if (!slack) {     // nobody reads the usage / nomination counters any more: reset them here, not in a phase of their own
            for (int m = tid; m < s.nR; m += BLP_THREADS) { s.usageL[m] = 0; s.markL[m] = 0; }
            return true;
        }
        if (slack) {      // the slack pass re-uses markL: clear the nomination counters first
            __syncthreads();
            for (int k = tid; k < K; k += BLP_THREADS)
                if (s.lix[k] >= 0) s.markL[s.lix[k]] = 0;
            __syncthreads();
        }
    }
    if (slack) {
        // priced rows without a user: markL[m] <- float bits of the smallest gap (rounded up) any column containing m has to
        // its target's minimum; a target that was repriced above blocks the row for this round (gap 0)
        const unsigned INF_BITS = 0x7f800000u;
        for (int m = tid; m < s.nR; m += BLP_THREADS)
            if (s.uL[m] > 0.0 && s.usageL[m] == 0) s.markL[m] = (int)INF_BITS;
        __syncthreads();
        for (int h = tid; h < s.nH; h += BLP_THREADS) {
            const Rows8 e8 = rows_of(s, h);
            const int k = s.membL[h];
            const bool busy = conflict && s.lix[k] >= 0 && s.mn[k] >= 0.0;
            const double gap = busy ? 0.0 : s.rcL[h] - s.best_rc[k];
            float gf = (float)gap;
            if ((double)gf < gap) gf = __uint_as_float(__float_as_uint(gf) + 1u);      // round up (gap >= 0)
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int e = e8.e[d];
                if (e != s.nR && s.markL[e] != 0) atomicMin(reinterpret_cast<unsigned*>(&s.markL[e]), __float_as_uint(gf));
            }
        }
        __syncthreads();
        for (int m = tid; m < s.nR; m += BLP_THREADS)
            if (s.uL[m] > 0.0 && s.usageL[m] == 0) {
                const unsigned b = (unsigned)s.markL[m];
                const double g = (b == INF_BITS) ? DINF : (double)__uint_as_float(b);
                s.uL[m] = fmax(0.0, s.uL[m] - (g * (1.0 + 9.5367431640625e-7) + 1e-9));
                s.markL[m] = 0;
            }
        __syncthreads();
    }
    return false;
}

// Two-target clusters (the most common kind) whose minimisers collide: the optimum straight away, by enumeration -- of the few
// pairs that can be optimal.  Every column gets the bit mask of its (dense) rows.  A pair through the cheapest column of one
// target (with the cheapest compatible column of the other) gives an upper bound UB; a column c of target t can only be in a pair
// of cost <= UB if cost[c] + (cheapest column of the other target) <= UB.  What passes that test (typically a dozen columns per
// target out of a few hundred) is listed, and only list x list is enumerated: min cost[i] + cost[j] over pairs with disjoint masks,
// ties to the lowest (i, j).  Exact: the optimal pair passes the test by construction.  (The plain n0 x n1 enumeration took 28-32 us
// on pairs with 250 + 250 columns and was the slowest ILP of one headline scan in five.)
// Needs <= 64 rows; returns false (nothing touched but rcL / the exact search's scratch) if that does not hold.
__device__ __forceinline__ bool enumerate_pair(const GStore&, Red*) { return false; }
__device__ __forceinline__ bool enumerate_pair(const LStore& s, Red* r) {
    if (s.nR > 64) return false;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = s.colb[0], b1 = s.colb[1], n0 = b1 - b0, n1 = s.colb[2] - b1;
    unsigned long long* mk = reinterpret_cast<unsigned long long*>(s.rcL);      // the reduced costs are dead until the next sweep
    unsigned short* list = s.ordL;                                              // [nH] candidates of target 0, then of target 1 (from n0 on)
    int* cnt = reinterpret_cast<int*>(s.enumL);                                 // [2] list lengths
    for (int h = tid; h < s.nH; h += BLP_THREADS) {
        const Rows8 e = rows_of(s, h);
        unsigned long long m = 0ull;
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if ((int)e.e[d] != s.nR) m |= 1ull << e.e[d];
        mk[h] = m;
    }
    if (tid < 2) cnt[tid] = 0;
    // cheapest column of either target: wavefronts 0 and 1
    if (wave < 2) {
        const int bb = wave ? b1 : b0, nn = wave ? n1 : n0;
        double cv = DINF;
        int ci = -1;
        for (int h = lane; h < nn; h += 64) {
            const double c = s.costL[bb + h];
            if (ci < 0 || c < cv) { cv = c; ci = h; }
        }
        wave_min_pair(cv, ci);
        if (lane == 0) { r->q[0][wave] = cv; r->q[1][wave] = (double)ci; }
    }
    __syncthreads();
    const double min0 = r->q[0][0], min1 = r->q[0][1];
    const int a0 = (int)r->q[1][0], a1 = (int)r->q[1][1];
    // upper bound: (a0, cheapest compatible j) and (cheapest compatible i, a1)
    double ub = DINF;
    {
        const unsigned long long ma0 = mk[b0 + a0], ma1 = mk[b1 + a1];
        for (int j = tid; j < n1; j += BLP_THREADS)
            if ((ma0 & mk[b1 + j]) == 0ull) ub = fmin(ub, min0 + s.costL[b1 + j]);
        for (int i = tid; i < n0; i += BLP_THREADS)
            if ((ma1 & mk[b0 + i]) == 0ull) ub = fmin(ub, s.costL[b0 + i] + min1);
        ub = wave_min_value(ub);
        __syncthreads();
        if (lane == 0) r->d[wave] = ub;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < BLP_THREADS / 64; ++w) ub = fmin(ub, r->d[w]);
    }
    if (ub >= DINF) return false;      // (cannot happen: the missed-detection columns have no rows)
    // candidates (sums are compared exactly as the enumeration forms them: cost[i] + cost[j])
    for (int i = tid; i < n0; i += BLP_THREADS)
        if (s.costL[b0 + i] + min1 <= ub) list[atomicAdd(&cnt[0], 1)] = (unsigned short)i;
    for (int j = tid; j < n1; j += BLP_THREADS)
        if (min0 + s.costL[b1 + j] <= ub) list[n0 + atomicAdd(&cnt[1], 1)] = (unsigned short)j;
    __syncthreads();
    const int m0 = cnt[0], m1 = cnt[1];
    double bv = DINF;
    int bi = -1;
    for (int p = tid; p < m0 * m1; p += BLP_THREADS) {
        const int i = list[p / m1], j = list[n0 + p % m1];
        const double v = s.costL[b0 + i] + s.costL[b1 + j];
        const int idx = i * n1 + j;
        if ((mk[b0 + i] & mk[b1 + j]) == 0ull && (v < bv || (v == bv && idx < bi) || bi < 0)) { bv = v; bi = idx; }
    }
    block_min_pair(bv, bi, r);
    if (bi < 0) return false;
    if (tid == 0) { s.ub_sel[0] = b0 + bi / n1; s.ub_sel[1] = b1 + bi % n1; }
    __threadfence_block();
    __syncthreads();
    return true;
}

// visiting order of the dive: targets by ascending minimal reduced cost (ties by index); identity for the HBM policy
__device__ __forceinline__ int dive_member(const GStore& s, int K, int pos) { return pos; }
__device__ __forceinline__ int dive_member(const LStore& s, int K, int pos) { return s.lix[pos]; }
__device__ __forceinline__ void dive_order(const GStore& s, int K) {}
__device__ __forceinline__ void dive_order(const LStore& s, int K) {      // rank sort of K <= 256 keys, one thread per target
    for (int k = threadIdx.x; k < K; k += BLP_THREADS) {
        const double v = s.best_rc[k];
        int rank = 0;
        for (int j = 0; j < K; ++j) {
            const double w = s.best_rc[j];
            rank += (w < v || (w == v && j < k)) ? 1 : 0;
        }
        s.lix[rank] = k;
    }
    __syncthreads();
}

// (DINF if some member has no column left that is compatible with the earlier picks: cannot happen while every member still has
// its conflict-free miss path, does happen in a cluster cut down to a few columns per member by reduced-cost fixing)
template <typename S> __device__ __forceinline__ double greedy_dive(const S& s, int K, int32_t* out_sel, Red* r) {
    double total = 0.0;
    bool ok = true;
    dive_order(s, K);
    for (int pos = 0; pos < K; ++pos) {
        const int k = dive_member(s, K, pos);
        double bv;
        int bi;
        argmin_member(s, k, true, -DINF, -1, bv, bi, r);      // (uniform result)
        if (threadIdx.x == 0) out_sel[k] = bi;
        if (bi < 0) { ok = false; continue; }
        total += s.cost(bi);
        set_marks(s, bi, 1);
    }
    __threadfence_block();
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const int h = out_sel[k];
        if (h >= 0) set_marks(s, h, 0);
    }
    return ok ? total : DINF;
}

// ---- exact search for small clusters the coordinate rounds did not certify ---------------------------------------------
// The clusters that end up here are a handful of near-duplicate tracks (K = 3..9 targets, a few hundred columns each) whose LP
// relaxation has a duality gap: prices zig-zag, the depth-first branch and bound pays ~10 us per node for re-priced bounds and
// the subgradient steps ~5 us per iteration -- 80..550 us where the ordinary cluster takes 9.  What conflicts can exist is tiny,
// though: only rows (measurement nodes) used by columns of >= 2 DIFFERENT members matter, and there are a dozen or two of those.
//   1. contested rows get dense ids (<= 64 of them, else the routine declines), every column a 64-bit signature;
//   2. columns are compared by REDUCED cost at the prices the coordinate rounds have reached (u >= 0): for any selection
//      sum cost = sum rc - sum_{rows used} u >= sum rc - sum_{all rows} u,  so
//      (rc of the columns chosen so far) + (cheapest rc of every member still open) - sum u  bounds every completion;
//   3. a greedy dive gives a feasible point; a column whose reduced cost alone lifts the root bound above it cannot be part of
//      anything better (reduced-cost fixing) and is dropped -- what survives is a few dozen columns per member, ranked by
//      (reduced cost, index) with one pass over the member's list per survivor;
//   4. depth-first search over the members (shortest list first) in that order, one search per wavefront: the state is
//      wave-uniform, the 64 lanes look at 64 consecutive entries of a ranked list at once ("first entry that neither conflicts
//      nor exceeds the bound" = one 16-byte LDS read per lane and two ballots; near-duplicate tracks conflict in most of their
//      columns).  The entries of the shortest list are dealt out to the wavefronts; the incumbent is one LDS word.
// Exact like the branch and bound: nothing is pruned unless its bound is strictly worse than a feasible point (ties survive: among
// optimal selections the one with the lexicographically lowest column indices in member order wins, whatever the wave timing).
constexpr int ENUM_MAXK = 10;
constexpr int ENUM_AFTER = 8;             // coordinate rounds a small cluster gets before the exact search
constexpr int ENUM_WIDE = 512, ENUM_AFTER_WIDE = 4;      // ... a cluster with more than ENUM_WIDE columns
constexpr int ENUM_BUDGET = 1 << 13;      // search nodes per thread before the routine gives up (the branch and bound takes over)
__device__ __forceinline__ unsigned long long enum_key(double v) {      // monotone map double -> u64
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double enum_val(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ bool enumerate_small(const GStore&, int, Red*, int&, unsigned long long*, double&) { return false; }
constexpr int ENUM_W = BLP_THREADS / 64;
constexpr size_t ENUM_LDS = BLP_THREADS <= 256 ? 768 : 768 * (BLP_THREADS / 256);             // search state of the wavefronts + scalars (see the carve in enumerate_small)
struct alignas(16) EnumEnt { double rc; unsigned long long sig; };      // reduced cost and contested-row signature of a ranked column
template <typename S> __device__ __forceinline__ double greedy_dive(const S& s, int K, int32_t* out_sel, Red* r);
__device__ __forceinline__ bool enumerate_small(const LStore& s, int K, Red* r, int& nodes_out, unsigned long long* stamp, double& ub_known) {
    if (K > ENUM_MAXK || K < 2) return false;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nH = s.nH, nR = s.nR;
    EnumEnt* xL = s.xL;
    // scratch (ENUM_LDS bytes): incumbent, bounds, member order, per-wavefront best cost / best selection / open positions
    unsigned long long* s_ub = reinterpret_cast<unsigned long long*>(s.enumL);
    unsigned long long* s_minrc = s_ub + 1;                             // [ENUM_MAXK] cheapest reduced cost of every member (as keys)
    double* s_rest = reinterpret_cast<double*>(s_minrc + ENUM_MAXK);    // [ENUM_MAXK + 1] ... summed over the levels still open
    double* s_bestc = s_rest + ENUM_MAXK + 1;                           // [ENUM_W]
    int* s_order = reinterpret_cast<int*>(s_bestc + ENUM_W);            // [ENUM_MAXK]
    int* s_len = s_order + ENUM_MAXK;                                   // [ENUM_MAXK] surviving columns of every member
    int* s_cnt = s_len + ENUM_MAXK;
    int* s_abort = s_cnt + 1;
    int* s_nodes = s_abort + 1;                                         // [ENUM_W]
    int* s_pos = s_nodes + ENUM_W;                                      // [ENUM_MAXK][ENUM_W] positions of the open levels
    int* s_best = s_pos + ENUM_MAXK * ENUM_W;                           // [ENUM_MAXK][ENUM_W] best selection of the wavefront
    static_assert(8 + ENUM_MAXK * 8 + (ENUM_MAXK + 1 + ENUM_W) * 8 + (2 * ENUM_MAXK + 2 + ENUM_W + 2 * ENUM_MAXK * ENUM_W) * 4 <= ENUM_LDS, "enumerate_small scratch");
    // 0. a feasible point (the dive leaves the marks at zero)
    double ub0 = greedy_dive(s, K, s.ch, r);
    if (ub_known < ub0) ub0 = ub_known;      // (the caller's incumbent in s.ub_sel[]: it passes every test below and is found again)
    stamp[1] = wall_clock64();
    // 1. which members use a row (the usage / nomination counters are zero between iterations: borrowed, zeroed again below);
    //    reduced costs at the current prices (the last sweep's values predate the last price step)
    if (tid == 0) { *s_cnt = 0; *s_abort = 0; *s_ub = enum_key(ub0); }
    if (tid < ENUM_W) { s_bestc[tid] = DINF; s_nodes[tid] = 0; }
    if (tid < ENUM_MAXK) { s_minrc[tid] = enum_key(DINF); s_len[tid] = 0; }
    __syncthreads();
    for (int h = tid; h < nH; h += BLP_THREADS) {
        const Rows8 e = rows_of(s, h);
        const int k = s.membL[h];
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (e.e[d] != nR) atomicOr(&s.markL[e.e[d]], 1 << k);
        const double rc = reduced_cost(s, h);
        s.rcL[h] = rc;
        atomicMin(&s_minrc[k], enum_key(rc));
    }
    double up = 0.0;
    for (int m = tid; m < nR; m += BLP_THREADS) up += s.uL[m];
    const double usum = block_sum(up, r);      // (its barriers also publish the row masks and the reduced costs)
    for (int m = tid; m < nR; m += BLP_THREADS) s.usageL[m] = (__popc(s.markL[m]) >= 2) ? atomicAdd(s_cnt, 1) : -1;
    __syncthreads();
    bool ok = *s_cnt <= 64;      // (a dive that ran into a dead end gives ub0 = DINF: nothing is fixed then, the search finds its own incumbent)
    stamp[2] = wall_clock64();
    auto slack = [](double ub) { return ub + 1e-9 * fmax(1.0, fabs(ub)); };
    if (ok) {
        // 2. reduced-cost fixing: the survivors of every member, in any order ...
        double root = -usum;
        for (int k = 0; k < K; ++k) root += enum_val(s_minrc[k]);
        const double lim0 = slack(ub0);
        for (int h = tid; h < nH; h += BLP_THREADS) {
            const int k = s.membL[h];
            const double c = s.rcL[h];
            if (root + (c - enum_val(s_minrc[k])) > lim0) continue;      // cannot be part of anything as good as the dive's selection
            EnumEnt en; en.rc = c; en.sig = (unsigned long long)h;
            xL[s.colb[k] + atomicAdd(&s_len[k], 1)] = en;
        }
        __syncthreads();
        // ... ranked by (reduced cost, index): a bitonic network per member, one wavefront each (comparators all ascending, so
        // the entries beyond the list's end act as +infinity without being there)
        for (int k = wave; k < K; k += ENUM_W) {
            const int hb = s.colb[k], n = s_len[k];
            int N = 1;
            while (N < n) N <<= 1;
            for (int size = 2; size <= N; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = lane; t < (N >> 1); t += 64) {
                        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));      // t-th index with the stride bit clear
                        const int hi = (stride == (size >> 1)) ? (lo ^ (size - 1)) : (lo | stride);
                        const int a = lo < hi ? lo : hi, b = lo < hi ? hi : lo;
                        if (b < n) {
                            const EnumEnt ea = xL[hb + a], eb = xL[hb + b];
                            if (eb.rc < ea.rc || (eb.rc == ea.rc && eb.sig < ea.sig)) { xL[hb + a] = eb; xL[hb + b] = ea; }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
        }
        __syncthreads();
        // the column of every ranked entry moves to ordL, its place is taken by the column's signature
        for (int k = 0; k < K; ++k)
            for (int i2 = s.colb[k] + tid; i2 < s.colb[k] + s_len[k]; i2 += BLP_THREADS) {
                const int h = (int)xL[i2].sig;
                const Rows8 e = rows_of(s, h);
                unsigned long long sg = 0ull;
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    if (e.e[d] != nR) { const int cc = s.usageL[e.e[d]]; if (cc >= 0) sg |= 1ull << cc; }
                s.ordL[i2] = (unsigned short)h;
                xL[i2].sig = sg;
            }
        __syncthreads();
        if (tid == 0) {      // members by surviving columns (fewest first); cheapest reduced costs of the levels still open
            for (int k = 0; k < K; ++k) s_order[k] = k;
            for (int i = 1; i < K; ++i) {
                const int v = s_order[i], nv = s_len[v];
                int j = i - 1;
                while (j >= 0 && s_len[s_order[j]] > nv) { s_order[j + 1] = s_order[j]; --j; }
                s_order[j + 1] = v;
            }
            s_rest[K] = 0.0;
            for (int i = K - 1; i >= 0; --i) s_rest[i] = s_rest[i + 1] + enum_val(s_minrc[s_order[i]]);
        }
        __syncthreads();
    }
    stamp[3] = wall_clock64();
    if (ok) {
        // 3. the search.  A level's state is its position in the member's list; the mask and the sums above it are rebuilt from
        //    the positions when the search comes back up (same order of additions: the same floating-point sums as on the way down).
        const int k0 = s_order[0], b0 = s.colb[k0], n0 = s_len[k0];
        double best = DINF;      // (wave-uniform)
        int nodes = 0;
        auto record = [&](double tot, int last_level) {      // a complete selection (positions in s_pos): better than this wavefront's best?
            bool better = tot < best;
            if (!better && tot == best) {      // lexicographically lower column indices in member order
                int d = 0;
                for (int q = 0; q <= last_level && d == 0; ++q) d = (int)s.ordL[s_pos[q * ENUM_W + wave]] - s_best[q * ENUM_W + wave];
                better = d < 0;
            }
            if (better) {
                best = tot;
                if (lane == 0) {
                    for (int q = 0; q <= last_level; ++q) s_best[q * ENUM_W + wave] = s.ordL[s_pos[q * ENUM_W + wave]];
                    atomicMin(s_ub, enum_key(tot));
                }
            }
        };
        // the entries of the shortest list are dealt out to the wavefronts (the cheapest ones first: they carry the large subtrees)
        for (int i0 = wave; i0 < n0 && !*s_abort; i0 += ENUM_W) {
            const EnumEnt f0 = xL[b0 + i0];
            if (f0.rc + s_rest[1] - usum > slack(enum_val(*s_ub))) break;      // ranked: no later entry of this list can do better
            const double tot0 = s.costL[s.ordL[b0 + i0]];
            ++nodes;
            if (lane == 0) s_pos[0 * ENUM_W + wave] = b0 + i0;
            if (K == 1) { record(tot0, 0); continue; }
            int L = 1;
            unsigned long long mask = f0.sig;
            double acc = tot0, arc = f0.rc;
            int i = s.colb[s_order[1]];
            while (true) {
                const int k = s_order[L], he = s.colb[k] + s_len[k];
                const double lim = slack(enum_val(*s_ub)), tail = s_rest[L + 1] - usum;
                // first entry at or after i that neither conflicts nor exceeds the bound; the bound is monotone along the list
                int found = -1;
                double frc = 0.0;
                unsigned long long fsig = 0ull;
                while (i < he) {
                    const int idx = i + lane;
                    const bool valid = idx < he;
                    const EnumEnt en = xL[valid ? idx : he - 1];
                    const bool over = !valid || (arc + en.rc + tail > lim);
                    const bool feas = !over && !(en.sig & mask);
                    const unsigned long long mo = __ballot(over);
                    unsigned long long mf = __ballot(feas);
                    if (mo) mf &= (1ull << (__ffsll((long long)mo) - 1)) - 1ull;
                    if (mf) {
                        const int fl = __ffsll((long long)mf) - 1;
                        found = i + fl;
                        frc = __shfl(en.rc, fl);
                        fsig = __shfl(en.sig, fl);
                        break;
                    }
                    if (mo) break;
                    i += 64;
                }
                if (found >= 0) {
                    if (++nodes > ENUM_BUDGET) { if (lane == 0) *s_abort = 1; break; }
                    const double tot = acc + s.costL[s.ordL[found]];
                    if (lane == 0) s_pos[L * ENUM_W + wave] = found;
                    if (L == K - 1) {
                        record(tot, L);
                        i = found + 1;      // (the list is ranked by reduced cost, not by cost: a later entry can still be cheaper)
                        continue;
                    }
                    mask |= fsig;          // one level down
                    acc = tot;
                    arc = arc + frc;
                    ++L;
                    i = s.colb[s_order[L]];
                    continue;
                }
                // back up one level: its next entry; mask and sums rebuilt from the positions above it
                --L;
                if (L < 1) break;
                i = s_pos[L * ENUM_W + wave] + 1;
                mask = f0.sig;
                acc = tot0;
                arc = f0.rc;
                for (int q = 1; q < L; ++q) {
                    const int pq = s_pos[q * ENUM_W + wave];
                    const EnumEnt en = xL[pq];
                    mask |= en.sig;
                    acc = acc + s.costL[s.ordL[pq]];
                    arc = arc + en.rc;
                }
            }
        }
        if (lane == 0) { s_bestc[wave] = best; s_nodes[wave] = nodes; }
    }
    __syncthreads();
    ok = ok && !*s_abort;
    // 4. the winner: lowest cost, then lexicographically lowest columns in member order
    if (ok) {
        int w = -1;
        for (int q = 0; q < ENUM_W; ++q) {
            if (!(s_bestc[q] < DINF)) continue;
            bool better = w < 0 || s_bestc[q] < s_bestc[w];
            if (!better && s_bestc[q] == s_bestc[w]) {
                int d = 0;
                for (int l = 0; l < K && d == 0; ++l) d = s_best[l * ENUM_W + q] - s_best[l * ENUM_W + w];
                better = d < 0;
            }
            if (better) w = q;
        }
        ok = w >= 0;
        if (ok) {
            for (int l = tid; l < K; l += BLP_THREADS) s.ub_sel[s_order[l]] = s_best[l * ENUM_W + w];
            int tn = 0;
            for (int q = 0; q < ENUM_W; ++q) tn += s_nodes[q];
            nodes_out = tn;
            ub_known = s_bestc[w];
        }
    }
    for (int m = tid; m < nR; m += BLP_THREADS) { s.usageL[m] = 0; s.markL[m] = 0; }
    __threadfence_block();
    __syncthreads();
    return ok;
}

// ---- giant clusters: reduced-cost fixing, then the LDS solver -----------------------------------------------------------------
// A cluster too large for LDS (thousands of columns: dense clutter, long windows) runs its dual phase on HBM scratch, ~5 us per
// sweep -- bearable -- but its branch and bound pays that per node, 0.2..5 s where the scan period is seconds.  Once the dual phase
// has prices u and a feasible point UB, though, a column whose reduced cost alone lifts the root bound
//     sum_k min_h rc_h - sum_m u_m
// above UB cannot be part of anything better (for any selection sum cost >= sum rc - sum u), and what survives is a small
// fraction of the columns.  If it fits the LDS tables (columns, rows, members), the cluster is rebuilt from the survivors and
// solved there from scratch -- coordinate rounds, exact search or branch and bound at LDS speed -- which is exact: the survivors
// contain every selection of cost <= UB, the incumbent itself included.
constexpr int MHT_BLP_REDUCE = -2;      // solve_core(GStore): not solved, thresholds of the survivors in s.mn[], counts in s.lix[]
__device__ __forceinline__ bool reducible(const BlpArgs&, const LStore&, int, double, Red*) { return false; }
__device__ __forceinline__ bool reducible(const BlpArgs& a, const GStore& s, int K, double UB, Red* r) {
    if (a.force_hbm || a.no_reduce || K > a.cap_k || a.PD > 8 || !(UB < DINF)) return false;
    const int tid = threadIdx.x;
    compute_minimisers(s, K, r);      // best_rc[] at the prices the dual phase ended with
    double up = 0.0, rs = 0.0;
    s.for_rows([&](int m) { up += s.u(m); });
    for (int k = tid; k < K; k += BLP_THREADS) rs += s.best_rc[k];
    const double root = block_sum(rs, r) - block_sum(up, r);
    const double lim = UB + 1e-9 * fmax(1.0, fabs(UB));
    int total = 0;
    for (int k = 0; k < K; ++k) {
        const double thr = lim - root + s.best_rc[k];
        double c = 0.0;
        for (int h = s.col_begin(k) + tid; h < s.col_end(k); h += BLP_THREADS) c += (reduced_cost(s, h) <= thr) ? 1.0 : 0.0;
        const int ck = (int)block_sum(c, r);
        if (tid == 0) { s.mn[k] = thr; s.lix[k] = ck; }
        total += ck;
        if (total > a.cap_h) return false;      // (uniform)
    }
    // distinct rows of the survivors (the marks are zero between uses)
    for (int k = 0; k < K; ++k) {
        const double thr = lim - root + s.best_rc[k];
        for (int h = s.col_begin(k) + tid; h < s.col_end(k); h += BLP_THREADS)
            if (reduced_cost(s, h) <= thr)
                for (int d = 0; d < s.PD; ++d) { const int e = s.ent(d, h); if (e >= 0) s.mark(e) = 1; }
    }
    __threadfence_block();
    __syncthreads();
    double nr = 0.0;
    s.for_rows([&](int m) { nr += s.mark(m) ? 1.0 : 0.0; s.mark(m) = 0; });
    const int rows = (int)block_sum(nr, r);
    __threadfence_block();
    __syncthreads();
    return rows < a.cap_r - 1;
}

// Solves one cluster; on return ub_sel[k] holds the chosen (policy-local) column of member k.
// `ub` in: cost of a feasible selection already sitting in s.ub_sel[] (DINF: none); out: cost of the selection returned there.
// A member of a team (see mht_kernels.h: TEAM_*): q of W, the team's shared incumbent word.  W = 1: a cluster searched by its own workgroup only.
struct Team {
    int q, W; unsigned long long* gub;
    // a team that spans the devices of a cluster-sharded step (BlpArgs::shard_team): the subtrees are dealt out over Wg = W x devices members,
    // this one is number qg; the incumbent word and the members' files stay per device, the devices' best selections meet in the exchange
    int qg, Wg;
    __device__ Team(int q_, int W_, unsigned long long* g_) : q(q_), W(W_), gub(g_), qg(q_), Wg(W_) {}
    __device__ Team(int q_, int W_, unsigned long long* g_, int qg_, int Wg_) : q(q_), W(W_), gub(g_), qg(qg_), Wg(Wg_) {}
};
#ifndef MHT_TEAM_LEVEL
#define MHT_TEAM_LEVEL 5
#endif
constexpr int TEAM_LEVEL = MHT_TEAM_LEVEL;      // the search is dealt out at this level (0-based): the members all walk the levels above it (a few dozen nodes),
                                                // below it each descends only into its own subtrees.  G9's 147 ms instance, 32 members: level 1 80 ms (two members hold half of the nodes),
                                                // 3: 53, 4: 46, 5: 35, 6: 64 ms (the shared levels grow).
// On HBM scratch a node costs ~0.25 ms instead of ~5 us and every node above the deal-out level is walked by ALL members: G20 (44 targets,
// 7 831 columns, not reducible), 32 members: level 2: 521 ms (9.1 k nodes in all, badly balanced), 3: 232 ms (11.4 k), 5: 865 ms (74.7 k),
// 8: 2 146 ms (195 k); one workgroup: 3 740 ms (14.9 k).
#ifndef MHT_TEAM_LEVEL_HBM
#define MHT_TEAM_LEVEL_HBM 3
#endif
__device__ __forceinline__ constexpr int team_level(const LStore&) { return TEAM_LEVEL; }
__device__ __forceinline__ constexpr int team_level(const GStore&) { return MHT_TEAM_LEVEL_HBM; }
// (Tried in round 5 and dropped: subtrees CLAIMED through a table -- first member to arrive takes it -- instead of dealt out by hash.  Exact and
// balanced, but the members all start on the same few subtrees in sequence and good incumbents arrive later: G20 14.1 k nodes instead of
// 11.4 k, 122 ms against 118, profiles/r05_ilp_tail.txt.)
__device__ __forceinline__ unsigned team_hash(int c0, int c1) {      // (level-0 column, level-1 column) -> member: price-independent
    unsigned h = (unsigned)c0 * 0x9E3779B1u ^ ((unsigned)c1 * 0x85EBCA6Bu + 0x7F4A7C15u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}
template <typename S> __device__ __forceinline__ void solve_core(const BlpArgs& a, const S& s, int K, Red* r, int& status, int& iters, int& nodes, unsigned long long* stamp,
                                                                 double& ub, const Team tm = Team(0, 1, nullptr)) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double& UB = ub;
    double best_LB = -DINF, theta = 1.0, utot = 0.0;
    int stall = 0;
    status = 0; iters = 0; nodes = 0;
    // dual iterations: the LDS policy does its coordinate-ascent rounds and then goes straight to the branch and bound (the
    // ascent is monotone, so its prices give a tight Lagrangian bound; on the rare clusters it does not certify -- two
    // near-duplicate tracks sharing a measurement in every scan of the window zig-zag geometrically -- subgradient steps took
    // up to ~100 more iterations where the branch and bound needs ~10 nodes); the HBM policy (large clusters, where a branch and
    // bound could explode) continues with subgradient steps up to max_iter before it branches.
    const int ca_rounds = coordinate_rounds(s, K);
    const int ca_end = a.max_iter < ca_rounds ? a.max_iter : ca_rounds;        // rounds [0, ca_end) are coordinate rounds
    const int it_cap = bb_after_rounds(s, K) ? ca_end : a.max_iter;
    // (a round of a cluster with many columns costs ~10 us instead of ~4.4: the exact search is worth its set-up earlier there)
    const int enum_after = (s.col_end(K - 1) - s.col_begin(0) > ENUM_WIDE) ? ENUM_AFTER_WIDE : ENUM_AFTER;
    const int enum_at = (ca_end > 0 && K >= 3 && K <= ENUM_MAXK) ? (ca_end < enum_after ? ca_end : enum_after) : -1;
    for (int it = 0; it <= it_cap; ++it) {
        iters = it;
        if (it == enum_at && !a.no_enum) {      // small cluster the first rounds did not certify: exact search
            const unsigned long long e0 = wall_clock64();
            const bool solved = enumerate_small(s, K, r, nodes, stamp, UB);
            stamp[5] = wall_clock64() - e0;
            if (solved) { status = MHT_BLP_BRANCHED; return; }
        }
        if (it == ca_end && ca_end > 0 && !bb_after_rounds(s, K)) {
            // a larger cluster that the coordinate rounds did not certify: the subgradient steps start from zero prices, as
            // they always did (from the rounds' prices they converged worse: a 29-target / 18 k-column cluster then ran
            // its branch and bound into the node limit)
            s.for_rows([&](int m) { s.u(m) = 0.0; });
            __threadfence_block();
            __syncthreads();
        }
        // a cluster on HBM scratch that has a feasible point: do few enough columns survive reduced-cost fixing at these prices to
        // go on in LDS?  (asked a few times: every sweep over thousands of columns in HBM costs ~1 ms)
        if ((it == 24 || it == 64 || it == 128) && UB < DINF && reducible(a, s, K, UB, r)) { status = MHT_BLP_REDUCE; return; }
        // A: per target the minimiser of the reduced cost (lowest column index wins ties)
        compute_minimisers(s, K, r);
        if (it == 0) stamp[1] = wall_clock64();
        // B: how often each measurement node is used by the minimisers
        if (!usage_counted_by_minimisers(s, K)) {
            for (int idx = tid; idx < K * s.PD; idx += BLP_THREADS) {
                const int k = idx / s.PD, d = idx - k * s.PD;
                add_usage(s, k, d);
            }
            __threadfence_block();
            __syncthreads();
        }
        if (it == 0) stamp[2] = wall_clock64();
        // C: certificate flags; the subgradient rounds also need the dual value and the subgradient norm
        const bool coord = it < ca_end;
        double nrm = 0.0, usum = 0.0, LB = 0.0;
        int conflict = 0, slack = 0;
        if (coord) {
            // coordinate round: only "is some row over-used" / "is some priced row unused"; nominations ride along (they
            // need the usage counters only).  r->i was last read several barriers ago: one barrier suffices.
            s.for_rows([&](int m) {
                const int us = s.usage(m);
                conflict |= (us >= 2);
                slack |= (s.u(m) > 0.0 && us == 0);
            });
            nominate(s, K);
            const int f = (__any(conflict) ? 1 : 0) | (__any(slack) ? 2 : 0);
            if (lane == 0) r->i[wave] = f;
            __threadfence_block();
            __syncthreads();
            int ff = 0;
#pragma unroll
            for (int w = 0; w < BLP_THREADS / 64; ++w) ff |= r->i[w];
            conflict = ff & 1;
            slack = (ff >> 1) & 1;
            if (it == 0) stamp[3] = wall_clock64();
            if (!conflict) {        // the minimisers are a feasible selection (and optimal if no priced row is unused)
                bool take = true;
                if (UB < DINF) {      // (only a cluster rebuilt from an HBM phase arrives with an incumbent: keep the better of the two)
                    double sc = 0.0;
                    for (int k = tid; k < K; k += BLP_THREADS) sc += s.cost(s.best_h[k]);
                    sc = block_sum(sc, r);
                    take = sc < UB;
                    if (take) UB = sc;
                }
                if (take)
                    for (int k = tid; k < K; k += BLP_THREADS) s.ub_sel[k] = s.best_h[k];
                __threadfence_block();
                __syncthreads();
            }
        } else {
            s.for_rows([&](int m) {
                const int us = s.usage(m);
                const double um = s.u(m);
                double g = (double)(us - 1);
                if (um <= 0.0 && g < 0.0) g = 0.0;
                nrm += g * g;
                usum += um;
                conflict |= (us >= 2);
                slack |= (um > 0.0 && us == 0);
            });
            double src = 0.0, sc = 0.0;
            for (int k = tid; k < K; k += BLP_THREADS) { src += s.best_rc[k]; sc += s.cost(s.best_h[k]); }
            {   // one fused block reduction for the four sums and the two flags
                const double v0 = wave_sum(nrm), v1 = wave_sum(usum), v2 = wave_sum(src), v3 = wave_sum(sc);
                int f = (__any(conflict) ? 1 : 0) | (__any(slack) ? 2 : 0);
                __syncthreads();
                if (lane == 0) { r->q[wave][0] = v0; r->q[wave][1] = v1; r->q[wave][2] = v2; r->q[wave][3] = v3; r->i[wave] = f; }
                __syncthreads();
                nrm = 0.0; utot = 0.0; src = 0.0; sc = 0.0; f = 0;
#pragma unroll
                for (int w = 0; w < BLP_THREADS / 64; ++w) {
                    nrm += r->q[w][0]; utot += r->q[w][1]; src += r->q[w][2]; sc += r->q[w][3]; f |= r->i[w];
                }
                conflict = f & 1;
                slack = (f >> 1) & 1;
            }
            if (it == 0) stamp[3] = wall_clock64();
            LB = src - utot;
            if (!conflict && sc < UB) {
                UB = sc;
                for (int k = tid; k < K; k += BLP_THREADS) s.ub_sel[k] = s.best_h[k];
                __threadfence_block();
                __syncthreads();
            }
        }
        bool done = false;
        if (!conflict && !slack) { status = MHT_BLP_CERTIFIED; done = true; }
        // a pair that one coordinate round did not settle (near-duplicate tracks: the prices zig-zag for several rounds and may
        // end in the branch and bound): exact by enumeration, optimal like a certificate.  (Pairs that settle in one round are
        // cheaper that way: the enumeration costs about one round.)
        if (!done && coord && it == 1 && K == 2 && enumerate_pair(s, r)) {
            status = MHT_BLP_CERTIFIED;
            break;
        }
        bool counters_reset = false;
        if (!done && coord) counters_reset = coordinate_step(s, K, conflict != 0, slack != 0);
        if (!done && !coord) {
            if (LB > best_LB + 1e-12) { best_LB = LB; stall = 0; }
            else if (++stall >= 10) { theta *= 0.5; stall = 0; }
            if (UB >= DINF || (conflict && (it % 4) == 0)) {
                const double g = greedy_dive(s, K, s.ch, r);
                if (g < UB) {
                    UB = g;
                    for (int k = tid; k < K; k += BLP_THREADS) s.ub_sel[k] = s.ch[k];
                    __threadfence_block();
                    __syncthreads();
                }
            }
            if (UB - best_LB <= 1e-12 * fmax(1.0, fabs(UB))) { status = MHT_BLP_CERTIFIED; done = true; }   // zero duality gap
            if (it == it_cap || nrm == 0.0) done = true;
            if (theta < 1.0 / 16.0) done = true;      // the dual bound has stalled through four step halvings: no certificate is
                                                      // coming (duality gap), hand over to the branch and bound now
        }
        // a certified cluster in LDS is finished here: its tables die with it (HBM scratch is shared and must be left clean)
        if (done && status != 0 && scratch_is_private(s)) break;
        // projected subgradient step on the prices (skipped when done); usage counters go back to zero either way
        const double step = (done || coord) ? 0.0 : theta * fmax(UB - LB, 1e-6) / nrm;
        if (!counters_reset)
            s.for_rows([&](int m) {
                if (!done && !coord) {
                    const double um = s.u(m);
                    double g = (double)(s.usage(m) - 1);
                    if (um <= 0.0 && g < 0.0) g = 0.0;
                    s.u(m) = fmax(0.0, um + step * g);
                }
                s.usage(m) = 0;
                if (coord) s.mark(m) = 0;      // nomination counters
            });
        __threadfence_block();
        __syncthreads();
        if (done) break;
    }
#ifdef MHT_BLP_TRACE
    if (tid == 0 && tm.q == 0) printf("[blp] dual phase: %d iterations, status %d, at %.2f ms\n", iters, status, 1e-5 * (double)(wall_clock64() - stamp[0]));
#endif
    if (status != 0) return;
    // ---- depth-first branch and bound -------------------------------------------------------------------------------
    // Positions 0..K-1 of the search are the cluster's targets in "hot first" order (ord[]): targets whose minimiser touches
    // an over-used or a priced row come first, the decisions that matter sit at the top of the tree.  At the first
    // BB_RE_LEVELS levels a node does not just evaluate the Lagrangian bound with the prices it inherits: it takes a few
    // subgradient steps on ITS residual problem (remaining targets, rows not blocked by the fixed columns) and checks the
    // certificate there -- conflict-free minimisers that use every priced free row are an optimal completion, the node is
    // solved without descending.  A static bound needs ~10^6 nodes on a 34-target cluster with an LP gap of 0.6 spread over 5
    // targets; this needs ~50 (HiGHS, on the host: branch on the fractional targets, re-optimise at every node -- same idea).
    // Any prices >= 0 give a valid bound, so exactness does not depend on the steps; the candidates of a level are enumerated
    // in (reduced cost, index) order under the prices of that node, which are restored (snapshot in HBM) whenever the search
    // returns to the level.
    if (UB >= DINF) {      // (before best_rc is re-used below: the dive orders the targets by it)
        UB = greedy_dive(s, K, s.ch, r);
        for (int k = tid; k < K; k += BLP_THREADS) s.ub_sel[k] = s.ch[k];
        __threadfence_block();
        __syncthreads();
    }
    if (reducible(a, s, K, UB, r)) { status = MHT_BLP_REDUCE; return; }      // (HBM policy only) solve_cluster rebuilds it in LDS
    int32_t* ord = reinterpret_cast<int32_t*>(s.best_rc);       // position -> member (best_rc is dead from here on)
    int32_t* bh = s.best_h;                                      // position -> minimiser column of the current node
    {   // hot-first order: minimisers and usage under the final prices
        compute_minimisers(s, K, r);
        if (!usage_counted_by_minimisers(s, K)) {
            for (int idx = tid; idx < K * s.PD; idx += BLP_THREADS) add_usage(s, idx / s.PD, idx % s.PD);
            __threadfence_block();
            __syncthreads();
        }
        for (int k = tid; k < K; k += BLP_THREADS) {
            int hot = 0;
            const int h = s.best_h[k];
            for (int d = 0; d < s.PD; ++d) {
                const int e = s.ent(d, h);
                if (e >= 0 && (s.usage(e) >= 2 || s.u(e) > 0.0)) ++hot;
            }
            s.lix[k] = hot;
        }
        __threadfence_block();
        __syncthreads();
        for (int k = tid; k < K; k += BLP_THREADS) {           // rank sort: more hot rows first, ties by member index
            const int hk = s.lix[k];
            int rank = 0;
            for (int j = 0; j < K; ++j) { const int hj = s.lix[j]; rank += (hj > hk || (hj == hk && j < k)) ? 1 : 0; }
            ord[rank] = k;
        }
        s.for_rows([&](int m) { s.usage(m) = 0; });
        __threadfence_block();
        __syncthreads();
    }
    // snapshot slot for the re-optimised prices (a small shared pool: branching clusters are rare); none free or none
    // configured -> plain static-bound search
    int slot = -1;
    if (a.bb_snap && a.bb_busy) {
        if (tid == 0) {
            int got = -1;
            for (int i = 0; i < BB_SLOTS && got < 0; ++i)
                if (atomicCAS(&a.bb_busy[i], 0, 1) == 0) got = i;
            r->i[0] = got;
        }
        __syncthreads();
        slot = r->i[0];
        __syncthreads();
    }
    double* snap = slot >= 0 ? a.bb_snap + (size_t)slot * BB_RE_LEVELS * a.bb_snap_rows : nullptr;
    if (tid == 0) s.cst[0] = 0.0;
    __threadfence_block();
    __syncthreads();
    int level = 0;
    bool enter = true;
#ifdef MHT_BLP_TRACE
    int tr_eval = 0, tr_upper = 0;
    unsigned long long tr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_q = 0, tr_n = wall_clock64();
#define TR_MARK(i) do { const unsigned long long _n = wall_clock64(); tr_t[i] += _n - tr_q; tr_q = _n; } while (0)
#else
#define TR_MARK(i) do {} while (0)
#endif
    status = MHT_BLP_BRANCHED;
    // team search: `own` = cost of the selection in THIS member's ub_sel[]; UB = the pruning bound = min(own, best value any member has
    // published).  The root node is processed with the member's own (deterministic) incumbent, so that every member sees the same
    // prices and the same level-0 candidates; from level 1 on the shared value prunes.
    double own = UB;
    const bool team = tm.Wg > 1 && tm.gub != nullptr;
    if (team && tid == 0 && UB < DINF) atomicMin(tm.gub, enum_key(UB));
    while (true) {
        if (team && enter && level >= 1) {
            __syncthreads();
            if (tid == 0) r->q[0][0] = enum_val(__hip_atomic_load(tm.gub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __syncthreads();
            const double g = r->q[0][0];
            __syncthreads();
            if (g < UB) UB = g;
        }
        const double eps = 1e-12 * fmax(1.0, fabs(UB));
        if (enter) {
            if (++nodes > a.node_limit) { status = MHT_BLP_NODE_LIMIT; break; }
            // wall-clock budget of the cluster (mht_forest_set_blp_time_limit): looked at every 64 nodes, decided by the whole
            // workgroup (the threads read the clock at slightly different times)
            if (a.time_limit > 0 && (nodes & 63) == 0 && block_or(wall_clock64() - stamp[0] > (unsigned long long)a.time_limit ? 1 : 0, r)) {
                status = MHT_BLP_NODE_LIMIT;
                break;
            }
            if (level == K) {
                if (s.cst[K] < UB - eps) {
                    UB = own = s.cst[K];
                    if (team && tid == 0) atomicMin(tm.gub, enum_key(UB));
                    for (int p = tid; p < K; p += BLP_THREADS) s.ub_sel[ord[p]] = s.ch[p];
                    __threadfence_block();
                    __syncthreads();
                }
                --level;
                set_marks(s, s.ch[level], 0);
                enter = false;
                continue;
            }
            const int rounds = (snap && level < BB_RE_LEVELS) ? BB_NODE_STEPS : 0;
#ifdef MHT_BLP_TRACE
            if (team && level <= team_level(s)) ++tr_upper;
#endif
            bool pruned = false;
            double rs_all = 0.0, usumU = 0.0, best_lb = -DINF;
            int best_rd = -1;
            bool final_eval = false;          // one more evaluation under the restored best prices, no step after it
            double* sl = snap ? snap + (size_t)(level < BB_RE_LEVELS ? level : 0) * a.bb_snap_rows : nullptr;
            // (lane i holds the member of the wavefront's i-th position: ord[] is in HBM for a giant cluster, a round trip in front of every
            // member's sweep otherwise)
            int my_k = 0;
            { const int p = level + wave + (BLP_THREADS / 64) * lane; if (p < K) my_k = ord[p]; }
            for (int rd = 0; rd <= rounds + 1; ++rd) {
#ifdef MHT_BLP_TRACE
                ++tr_eval;
                { const unsigned long long _n = wall_clock64(); tr_t[6] += _n - tr_n; tr_q = _n; }      // [6]: outside the evaluations
#endif
                // minimisers of the remaining targets among the columns compatible with the fixed ones
                {
                    constexpr int NW = BLP_THREADS / 64;
                    const int p0 = level + wave;
                    sweep_wave<true>(s, K > p0 ? (K - p0 + NW - 1) / NW : 0, lane, [&](int i) { return i < 64 ? __shfl(my_k, i) : ord[p0 + NW * i]; },
                                     [&](int i, double bv, int bi) { if (lane == 0) { s.mn[p0 + NW * i] = (bi < 0) ? DINF : bv; bh[p0 + NW * i] = bi; } });
                }
                __threadfence_block();
                __syncthreads();
                TR_MARK(0);
                double rs = 0.0, cs = 0.0;
                int dead = 0;
                for (int p = level + tid; p < K; p += BLP_THREADS) {
                    const double v = s.mn[p];
                    if (v >= DINF) dead = 1;
                    else { rs += v; cs += s.cost(bh[p]); }
                }
                dead = block_or(dead, r);
                TR_MARK(1);
                if (dead) { pruned = true; break; }
                int conflict = 0, slack = 0;
                double nrm = 0.0, us = 0.0;
                if (rounds > 0) {          // usage of the free rows by these minimisers
                    for (int idx = tid; idx < (K - level) * s.PD; idx += BLP_THREADS) {
                        const int p = level + idx / s.PD, d = idx % s.PD;
                        const int e = s.ent(d, bh[p]);
                        if (e >= 0) atomicAdd(&s.usage(e), 1);
                    }
                    __threadfence_block();
                    __syncthreads();
                }
                TR_MARK(2);
                s.for_rows([&](int m) {
                    if (s.mark(m)) return;                      // blocked by a fixed column: not part of the residual problem
                    const double um = s.u(m);
                    us += um;
                    if (rounds > 0) {
                        const int ug = s.usage(m);
                        double g = (double)(ug - 1);
                        if (um <= 0.0 && g < 0.0) g = 0.0;
                        nrm += g * g;
                        conflict |= (ug >= 2);
                        slack |= (um > 0.0 && ug == 0);
                    }
                });
                // one fused block reduction for the four sums and the two flags (each value summed exactly as block_sum does it)
                double csum = 0.0;
                {
                    const double v0 = wave_sum(rs), v1 = wave_sum(us), v2 = wave_sum(cs), v3 = wave_sum(nrm);
                    int f = (__any(conflict) ? 1 : 0) | (__any(slack) ? 2 : 0);
                    __syncthreads();
                    if (lane == 0) { r->q[wave][0] = v0; r->q[wave][1] = v1; r->q[wave][2] = v2; r->q[wave][3] = v3; r->i[wave] = f; }
                    __syncthreads();
                    rs_all = 0.0; usumU = 0.0; nrm = 0.0; f = 0;
#pragma unroll
                    for (int w = 0; w < BLP_THREADS / 64; ++w) {
                        rs_all += r->q[w][0]; usumU += r->q[w][1]; csum += r->q[w][2]; nrm += r->q[w][3]; f |= r->i[w];
                    }
                    conflict = f & 1;
                    slack = (f >> 1) & 1;
                }
                const double lb = s.cst[level] + rs_all - usumU;
                TR_MARK(3);
                bool stop = false;
                if (rounds > 0 && !final_eval && lb > best_lb) {       // keep the best prices of the node (a Polyak step with a loose
                    best_lb = lb;                                      // upper bound can overshoot badly)
                    best_rd = rd;
                    s.for_rows([&](int m) { sl[m] = s.u(m); });
                }
                if (rounds > 0) {
                    if (!conflict) {                            // the minimisers complete the fixed columns feasibly
                        const double cand = s.cst[level] + csum;
                        if (cand < UB - eps) {
                            UB = own = cand;
                            if (team && tid == 0) atomicMin(tm.gub, enum_key(UB));
                            for (int p = tid; p < K; p += BLP_THREADS) s.ub_sel[ord[p]] = (p < level) ? s.ch[p] : bh[p];
                            __threadfence_block();
                            __syncthreads();
                        }
                        if (!slack) { pruned = true; stop = true; }      // ... and optimally: the node is solved
                    }
                }
                if (!stop && lb >= UB - 1e-12 * fmax(1.0, fabs(UB))) { pruned = true; stop = true; }
                TR_MARK(4);
                const bool step_on = !stop && !final_eval && rd < rounds && nrm > 0.0;
                // no further step: if the last evaluated prices are not the node's best, go back to those and evaluate once more
                const bool redo = !stop && !final_eval && !step_on && rounds > 0 && best_rd != rd;
                if (rounds > 0) {          // subgradient step on the free rows / restore of the best prices; counters back to zero
                    const double step = step_on ? fmax(UB - lb, 1e-6) / nrm : 0.0;
                    s.for_rows([&](int m) {
                        if (redo) {
                            s.u(m) = sl[m];
                        } else if (!s.mark(m) && step_on) {
                            const double um = s.u(m);
                            double g = (double)(s.usage(m) - 1);
                            if (um <= 0.0 && g < 0.0) g = 0.0;
                            s.u(m) = fmax(0.0, um + step * g);
                        }
                        s.usage(m) = 0;
                    });
                    __threadfence_block();
                    __syncthreads();
                }
                TR_MARK(5);
#ifdef MHT_BLP_TRACE
                tr_n = wall_clock64();
#endif
                if (redo) { final_eval = true; continue; }
                if (stop || !step_on) break;
            }
            if (pruned) {
                if (level == 0) break;
                --level;
                set_marks(s, s.ch[level], 0);
                enter = false;
                continue;
            }
            // (sl holds the node's prices -- the best of its rounds -- for the returns from its children)
            if (tid == 0) { s.rest[level] = rs_all - s.mn[level]; s.uus[level] = usumU; s.lrc[level] = -DINF; s.lix[level] = -1; }
            __threadfence_block();
            __syncthreads();
            enter = false;
        } else if (snap && level + 1 < BB_RE_LEVELS) {
            // back from a child that re-optimised the prices: restore this node's
            const double* slr = snap + (size_t)level * a.bb_snap_rows;
            s.for_rows([&](int m) { s.u(m) = slr[m]; });
            __threadfence_block();
            __syncthreads();
        }
        // next candidate of the target at this level in increasing (reduced cost, index) order under the node's prices
        double bv;
        int bi;
        argmin_member(s, ord[level], true, s.lrc[level], s.lix[level], bv, bi, r);
        if (bi < 0 || s.cst[level] + bv + s.rest[level] - s.uus[level] >= UB - eps) {
            if (level == 0) break;
            --level;
            set_marks(s, s.ch[level], 0);
            continue;
        }
        bool foreign = false;
        if (team && level == team_level(s)) {      // the columns fixed at levels 0 .. team_level name the subtree: whose is it?
            unsigned hsh = 0x9E3779B9u;
            for (int l = 0; l < team_level(s); ++l) hsh = team_hash((int)hsh, s.to_global(s.ch[l]));
            hsh = team_hash((int)hsh, s.to_global(bi));
            foreign = hsh % (unsigned)tm.Wg != (unsigned)tm.qg;
        }
        if (foreign) {
            // another member's subtree: step over the candidate (the enumeration state moves on, nothing is fixed)
            __syncthreads();      // (every thread has read ch[0] / lrc / lix of this level)
            if (tid == 0) { s.lrc[level] = bv; s.lix[level] = bi; }
            __threadfence_block();
            __syncthreads();
            continue;
        }
        if (tid == 0) {
            s.lrc[level] = bv;
            s.lix[level] = bi;
            s.ch[level] = bi;
            s.cst[level + 1] = s.cst[level] + s.cost(bi);
        }
        set_marks(s, bi, 1);
        ++level;
        enter = true;
    }
#ifdef MHT_BLP_TRACE
    if (tid == 0 && team) printf("[blp] member %d of %d: %d nodes (%d at or above the deal-out level), %d evaluations, left at %.2f ms; us per evaluation: sweep %.1f sums %.1f usage %.1f rows %.1f keep %.1f step %.1f outside %.1f\n", tm.q, tm.W, nodes, tr_upper, tr_eval,
                                 1e-5 * (double)(wall_clock64() - stamp[0]), 1e-2 * tr_t[0] / tr_eval, 1e-2 * tr_t[1] / tr_eval, 1e-2 * tr_t[2] / tr_eval, 1e-2 * tr_t[3] / tr_eval,
                                 1e-2 * tr_t[4] / tr_eval, 1e-2 * tr_t[5] / tr_eval, 1e-2 * tr_t[6] / tr_eval);
#endif
    ub = own;      // (the cost of what ub_sel[] holds: a team member may have pruned with a better value found elsewhere)
    for (int l = 0; l < level; ++l) set_marks(s, s.ch[l], 0);     // leave no marks behind
    if (slot >= 0 && tid == 0) atomicExch(&a.bb_busy[slot], 0);
}

// track termination test and N-scan prune depth for target t whose selected leaf is child s (forest mode)
// ---- forest epilogue: track termination, N-scan prune decision, new root, report record, surviving leaf range -------
// (tracker.py:891-916, pyTarget.py:343-356).  It is latency-, not throughput-bound (a handful of dependent look-ups per
// target), so it is arranged in as few dependent load levels as possible: what depends only on the target is fetched
// before the selection is known (TgtPre), the layers of the ring are addressed arithmetically (no pointer table), and
// survival is decided with ONE ancestor-table look-up per child (child descends from the new root <=> its ancestor at the
// root's depth IS the new root) instead of comparing path prefixes.
struct TgtPre { int j, rscan, rnode, id, lab, cb, ce; double rootc; uint8_t rootf; };
__device__ __forceinline__ TgtPre load_target(const BlpArgs& a, int t) {
    TgtPre p;
    const int dg = a.t_depth[t] + 1, w = a.t_window[t] & 0xff;      // (bits 8..: mht_kernels.h WIN_REBUILT_*)
    p.j = dg > w ? dg - w : 0;             // layers the root advances (pyTarget.pruneDepth)
    p.rscan = a.t_root_scan[t]; p.rnode = a.t_root_node[t]; p.id = a.t_id[t]; p.lab = a.t_label[t];
    p.cb = a.tchild[t]; p.ce = a.tcend[t];
    p.rootc = a.t_root_cnllr[t]; p.rootf = a.t_root_f32[t];
    return p;
}
template <typename T> __device__ __forceinline__ const T* ring_ptr(const T* base0, size_t stride, int k) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base0) + (size_t)k * stride);
}
constexpr int KEY_DEAD = -1, KEY_ALL = -2;
// Returns the survival key of the target: KEY_DEAD (terminated), KEY_ALL (alive, root stays) or the node index of the
// new root (children whose ancestor table holds it at level j-1 survive).
// (publication for an overlapping grow launch, BlpArgs::rec0: the new root's cumulative score is written through here, the record that
// names it valid follows from whoever knows the surviving leaf range: blp_publish)
__device__ __forceinline__ void blp_publish(const BlpArgs& a, int t, int key, int j, int rf, int count, int first) {
    if (!a.rec0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the root score's store has left)
    __hip_atomic_store(&a.rec0[t], tgt_rec(a.pub_scan, key != -1, rf, j, count, count > 0 ? first : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int finish_target(const BlpArgs& a, int t, int s, const TgtPre& p, bool store, int* rf_out = nullptr) {
    const int kc = a.kc;
    const double cn = a.cnllr[s];
    const uint8_t fl = a.flags[s];
    const int smeas = ring_ptr(a.ring0.meas, a.ring_stride, kc)[s];
    double sx[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) sx[k] = a.x[(size_t)k * a.cap + s];
    const double sx0 = sx[0], sx1 = sx[1];
    const int j = p.j;
    const int anc_l = a.apath[(size_t)s * a.pds + (j > 0 ? j - 1 : 0)];      // unconditional (clamped level): one batch with the loads above
    const int anc = (j > 0) ? anc_l : -1;
    const bool f32score = (fl & F_SCORE_F32) && p.rootf;
    // getScore() (pyTarget.py:124) with NumPy scalar promotion: float32 - float32 stays float32
    const double score = f32score ? (double)((float)cn - (float)p.rootc) : cn - p.rootc;
    int status = 0;
    if (isfinite(a.radar_range)) {
        const double dx = sx0 - a.radar_x, dy = sx1 - a.radar_y;
        if (sqrt(dx * dx + dy * dy) > a.radar_range) status = 1;                          // tracker.py:895
    }
    if (!status) {
        const double per = f32score ? (double)((float)score / (float)(a.Nwin + 1)) : score / (double)(a.Nwin + 1);
        if (per > a.score_limit) status = 2;                                                 // tracker.py:902
        else if (cn > a.cnllr_limit) status = 3;                                             // tracker.py:908
    }
    int rscan = p.rscan, rnode = p.rnode;
    const bool moved = status == 0 && j > 0;      // new root = the selected leaf's ancestor j levels below the old root
    if (moved) { rscan += j; rnode = anc; }
    // the (new or old) root's record: seven look-ups in one batch, none behind a branch
    const int kr = rscan % a.R;
    const double* rx = ring_ptr(a.ring0.x, a.ring_stride, kr);
    const double rc_l = ring_ptr(a.ring0.cnllr, a.ring_stride, kr)[rnode];
    const uint8_t rf_l = ring_ptr(a.ring0.flags, a.ring_stride, kr)[rnode];
    double rxv[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) rxv[k] = rx[(size_t)k * a.cap + rnode];
    const int rmeas = ring_ptr(a.ring0.meas, a.ring_stride, kr)[rnode];
    const double rc = moved ? rc_l : p.rootc;
    const uint8_t rf = moved ? (uint8_t)((rf_l & F_SCORE_F32) ? 1 : 0) : p.rootf;
    if (store) {
        a.t_alive[t] = status;                 // 0 = alive, else the termination reason
        a.t_jdrop[t] = j;
        a.t_score[t] = score;
        a.w_root_scan[t] = rscan; a.w_root_node[t] = rnode; a.w_root_f32[t] = rf;
        if (a.rec0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(&a.w_root_cnllr[t]), (unsigned long long)__double_as_longlong(rc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.w_root_cnllr[t] = rc;
        mht_target_report& r = a.rec[t];
        r.id = p.id;
        r.status = status;
        r.sel_node = s;
        r.sel_meas = smeas;
        r.root_scan = rscan;
        r.root_node = rnode;
#pragma unroll
        for (int k = 0; k < NX; ++k) r.sel_x[k] = sx[k];
        r.sel_cnllr = cn;
        r.score = score;
        r.root_cnllr = rc;
#pragma unroll
        for (int k = 0; k < NX; ++k) r.root_x[k] = rxv[k];
        r.root_meas = rmeas;
        r.cluster = p.lab;
    }
    if (rf_out) *rf_out = rf;
    return status ? KEY_DEAD : (j > 0 ? anc : KEY_ALL);
}

// N-scan pruning, child side (pyTarget.pruneDepth -> _pruneAllHypothesisExceptThis, pyTarget.py:330-356).  Survivors of a
// target are one contiguous DFS range: only (first, count) are recorded.  One wavefront per target; `va0` is the
// ancestor entry of the lane's child in the first chunk, fetched by the caller before the key was known.
__device__ __forceinline__ int sweep_prefetch(const BlpArgs& a, int j, int cb, int ce, int lane) {
    return (j > 0 && cb + lane < ce) ? a.apath[(size_t)(cb + lane) * a.pds + (j - 1)] : -1;
}
__device__ __forceinline__ void sweep_survivors(const BlpArgs& a, int t, int j, int cb, int ce, int key, int va0, int lane, int rf = -1) {      // rf < 0: the target's root flag is read back
    int count = 0, first = 0x7fffffff;
    if (key == KEY_ALL) {
        count = ce - cb;
        if (count > 0) first = cb;
    } else if (key != KEY_DEAD) {
        for (int c0 = cb; c0 < ce; c0 += 64) {
            const int c = c0 + lane;
            const int v = (c0 == cb) ? va0 : (c < ce ? a.apath[(size_t)c * a.pds + (j - 1)] : -1);
            const unsigned long long m = __ballot(c < ce && v == key);
            if (m && first == 0x7fffffff) first = c0 + __ffsll((long long)m) - 1;
            count += __popcll(m);
        }
    }
    if (lane == 0) {
        a.t_count[t] = count; a.t_firstsurv[t] = first;
        if (a.rec0) blp_publish(a, t, key, j, rf >= 0 ? rf : (int)a.w_root_f32[t], count, first);
    }
}

// surviving leaf ranges of the members of a cluster: one wavefront per target; key[k] = survival key from finish_target
__device__ __forceinline__ void prune_members(const BlpArgs& a, const int32_t* mem, int K, const int32_t* key) {
    if (!a.t_alive) return;
    __threadfence_block();
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int k = threadIdx.x >> 6; k < K; k += BLP_THREADS / 64) {
        const int t = mem[k];
        const int j = a.t_jdrop[t], cb = a.tchild[t], ce = a.tcend[t];
        sweep_survivors(a, t, j, cb, ce, key[k], sweep_prefetch(a, j, cb, ce, lane), lane);
    }
}

// a cluster as the solver sees it: index among all clusters (by smallest member), offset of its (ascending) member list in cl_members,
// size, smallest member (= its label; only filled in when the clusters come from the union-find)
struct ClRef { int c, p0, K, root; };
__device__ __forceinline__ ClRef cl_ref(const BlpArgs& a, int c) { const int p0 = a.cl_ptr[c]; return ClRef{c, p0, a.cl_ptr[c + 1] - p0, -1}; }
// (clusters from the union-find: t_label is written by ONE workgroup of the launch, for the host -- the label travels with the ClRef)
__device__ __forceinline__ TgtPre load_target_cr(const BlpArgs& a, int t, const ClRef& cr) {
    TgtPre p = load_target(a, t);
    if (a.uf_epoch) p.lab = cr.root;
    return p;
}
// my_t: member `threadIdx.x` of the cluster if the caller has it at hand (-1: read from the member list), pre_in: its record if already fetched
__device__ __forceinline__ void solve_cluster(const BlpArgs& a, const ClRef cr, unsigned long long* uw, Red* r, unsigned char* lds,
                                              const Team tm = Team(0, 1, nullptr), const int team_idx = -1, const int my_t = -1, const TgtPre* pre_in = nullptr, const int dbg_bx = -1) {
    const int tid = threadIdx.x;
    const int c = cr.c;
    const int K = cr.K;
    const int32_t* mem = a.cl_members + cr.p0;
    const int slot = cr.p0 + c;
    const int UW = (a.n_mnodes + 63) >> 6;
    const unsigned long long t_begin = wall_clock64();
    const unsigned long long c_begin = clock64();
    TgtPre pre = {};      // per-member data of the prune epilogue, fetched now so that its latency hides behind the solve
    const bool pre_ok = a.t_alive && K <= BLP_THREADS;
    if (pre_ok && tid < K) { if (pre_in && my_t >= 0) pre = *pre_in; else pre = load_target_cr(a, my_t >= 0 ? my_t : mem[tid], cr); }
    // LDS carve: every block below is a multiple of 16 bytes and the dynamic segment starts at offset 0 (the kernel has
    // no static __shared__), so the 16-byte column records stay aligned without integer round trips -- those would
    // make the compiler lose the LDS address space and emit slow flat accesses.
    LStore s;
    unsigned char* q = lds;
    const int L_MAXH = a.cap_h, L_MAXR = a.cap_r, L_MAXK = a.cap_k, L_KPAD = a.cap_k + 4;      // this launch's LDS tier (BlpArgs)
    int* s_wbase = reinterpret_cast<int*>(q); q += (size_t)a.cap_uw * 4;
    int* s_scal = reinterpret_cast<int*>(q); q += 16;
    int& s_nR = s_scal[0];
    int& s_nH = s_scal[1];
    s.entL = reinterpret_cast<unsigned short*>(q); q += (size_t)L_MAXH * 16;
    s.costL = reinterpret_cast<double*>(q); q += (size_t)L_MAXH * 8;
    s.rcL = reinterpret_cast<double*>(q); q += (size_t)L_MAXH * 8;
    s.uL = reinterpret_cast<double*>(q); q += (size_t)L_MAXR * 8;
    s.minkey = reinterpret_cast<unsigned long long*>(q); q += (size_t)L_KPAD * 8;
    s.best_rc = reinterpret_cast<double*>(q); q += (size_t)L_KPAD * 8;
    s.cst = reinterpret_cast<double*>(q); q += (size_t)L_KPAD * 8;
    s.uus = reinterpret_cast<double*>(q); q += (size_t)L_KPAD * 8;
    s.lrc = reinterpret_cast<double*>(q); q += (size_t)L_KPAD * 8;
    s.rest = reinterpret_cast<double*>(q); q += (size_t)L_KPAD * 8;
    s.mn = reinterpret_cast<double*>(q); q += (size_t)L_KPAD * 8;
    s.usageL = reinterpret_cast<int32_t*>(q); q += (size_t)L_MAXR * 4;
    s.markL = reinterpret_cast<int32_t*>(q); q += (size_t)L_MAXR * 4;
    s.colb = reinterpret_cast<int32_t*>(q); q += (size_t)L_KPAD * 4;
    s.gbase = reinterpret_cast<int32_t*>(q); q += (size_t)L_KPAD * 4;
    s.best_h = reinterpret_cast<int32_t*>(q); q += (size_t)L_KPAD * 4;
    s.ub_sel = reinterpret_cast<int32_t*>(q); q += (size_t)L_KPAD * 4;
    s.ch = reinterpret_cast<int32_t*>(q); q += (size_t)L_KPAD * 4;
    s.lix = reinterpret_cast<int32_t*>(q); q += (size_t)L_KPAD * 4;
    s.membL = reinterpret_cast<unsigned short*>(q); q += (size_t)L_MAXH * 2;
    s.ordL = reinterpret_cast<unsigned short*>(q); q += (size_t)L_MAXH * 2;
    s.enumL = reinterpret_cast<unsigned short*>(q); q += ENUM_LDS;
    s.xL = reinterpret_cast<EnumEnt*>(q); q += (size_t)L_MAXH * 16;
    s.gcolL = reinterpret_cast<int32_t*>(q); q += (size_t)L_MAXH * 4;
    s.reduced = false;
    const bool small_k = K <= L_MAXK;
    for (int w = tid; w < UW; w += BLP_THREADS) uw[w] = 0ull;
    // column ranges of the members: one global round trip, then a wave scan
    if (small_k) {
        if (tid < 64) {
            int carry = 0;
            for (int base = 0; base < K; base += 64) {
                const int k = base + tid;
                int gb = 0, n = 0;
                if (k < K) {
                    if (pre_ok && K <= 64) { gb = pre.cb; n = pre.ce - gb; }      // (k = tid: the member's record is at hand)
                    else { const int t = mem[k]; gb = a.tchild[t]; n = a.tcend[t] - gb; }
                }
                int incl = n;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (tid >= o) incl += v;
                }
                if (k < K) { s.colb[k] = carry + incl - n; s.gbase[k] = gb; if (pre_ok && K <= 64) s.lix[k] = pre.j; }      // (lix: the member's prune depth until the solver takes the table -- the copy below prefetches the survivors' sweep with it)
                carry += __shfl(incl, 63);
            }
            if (tid == 0) { s.colb[K] = carry; s_nH = carry; }
        }
    } else if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < K; ++k) acc += a.tcend[mem[k]] - a.tchild[mem[k]];
        s_nH = acc;
    }
    __syncthreads();
    const int nH = s_nH;
    // two-tier launches (groups of sectors): tier 1 = small LDS footprint, several workgroups per CU, takes the clusters that fit
    // it; tier 2 = the default footprint, takes the rest
    if (a.tier == 1 && !(K <= a.t1_k && nH <= a.t1_h)) {      // left for tier 2
        if (tid == 0) a.big_list[atomicAdd(a.big_count, 1)] = c;
        __syncthreads();
        return;
    }
    const bool lds_cols = small_k && nH <= L_MAXH && a.PD <= 8 && !a.force_hbm;
    // the epilogue's survivor test needs ONE ancestor-table entry per child (level j - 1 of its target): fetched with the column, parked in gcolL (only a
    // REDUCED cluster uses that table for something else, and it sweeps its members' ranges generically): one global round trip less behind the solve
    // (+0.5 % over 400 scans, profiles/r06_experiments.txt; the new root's record fetched speculatively through the same entry: measured, -0.2 %, not kept)
    const bool anc_pf = lds_cols && pre_ok && K <= 64 && a.t_alive && a.pds == 8;
    s.nH = nH; s.PD = a.PD; s.K = K;
    // ---- measurement nodes of the cluster (union of the rows of its columns); in the LDS case the columns are
    //      copied in the same sweep: every thread issues the PD+1 loads of a column back to back (one round trip)
    if (lds_cols) {
        // two columns per thread and pass: all 2 x (PD + 1) global loads are issued before the first is consumed, so a
        // cluster of up to 512 columns is copied in ONE global round trip (the loop is latency, not bandwidth, bound)
        for (int h0 = tid; h0 < nH; h0 += 2 * BLP_THREADS) {
            int g[2], ev[2][8], jm[2], an[2];
            double cs[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int h = h0 + q * BLP_THREADS;
                g[q] = -1; jm[q] = 0;
                if (h < nH) {
                    int lo = 0, hi = K;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s.colb[mid] <= h) lo = mid; else hi = mid; }
                    g[q] = s.gbase[lo] + (h - s.colb[lo]);
                    s.membL[h] = (unsigned short)lo;
                    if (anc_pf) jm[q] = s.lix[lo];
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {      // unconditional loads (clamped index), masked afterwards: no branch between them
                const int gq = g[q] >= 0 ? g[q] : 0;
                cs[q] = a.cost[gq];
                an[q] = -1;
                if (anc_pf) an[q] = a.apath[(size_t)gq * 8 + (jm[q] > 0 ? jm[q] - 1 : 0)];
                if (a.pds) {      // forest: one 32-byte record per column (entries beyond PD are -1)
                    const int4* rec = reinterpret_cast<const int4*>(a.path + (size_t)gq * a.pds);
                    const int4 r0 = rec[0], r1 = rec[1];
                    const int rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                    for (int d = 0; d < 8; ++d) ev[q][d] = g[q] >= 0 ? rv[d] : -1;
                } else {
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const int v = (d < a.PD) ? a.path[(size_t)d * a.cap + gq] : -1;
                        ev[q][d] = g[q] >= 0 ? v : -1;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int h = h0 + q * BLP_THREADS;
                if (g[q] < 0) continue;
                s.costL[h] = cs[q];
                if (anc_pf) s.gcolL[h] = an[q];
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    s.entL[h * 8 + d] = (unsigned short)(ev[q][d] < 0 ? 0xffff : ev[q][d]);     // global node id for now
                    if (ev[q][d] >= 0) atomicOr(&uw[ev[q][d] >> 6], 1ull << (ev[q][d] & 63));
                }
            }
        }
    } else {
        for (int k = 0; k < K; ++k) {
            const int t = mem[k];
            for (int d = 0; d < a.PD; ++d)
                for (int h = a.tchild[t] + tid; h < a.tcend[t]; h += BLP_THREADS) {
                    const int e = a.pds ? a.path[(size_t)h * a.pds + d] : a.path[(size_t)d * a.cap + h];
                    if (e >= 0) atomicOr(&uw[e >> 6], 1ull << (e & 63));
                }
        }
    }
    auto row_prefix = [&]() {      // exclusive prefix of the popcounts of the row bitset: dense local row ids
        __syncthreads();
        if (tid < 64) {
            int carry = 0;
            for (int base = 0; base < UW; base += 64) {
                const int w = base + tid;
                const int pc = (w < UW) ? __popcll(uw[w]) : 0;
                int incl = pc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (tid >= o) incl += v;
                }
                if (w < UW) s_wbase[w] = carry + incl - pc;
                carry += __shfl(incl, 63);
            }
            if (tid == 0) s_nR = carry;
        }
        __syncthreads();
    };
    row_prefix();
    int nR = s_nR;
    int status, iters, nodes;
    unsigned long long stamp[6] = {t_begin, 0, 0, 0, 0, 0};      // [0]: start of the cluster (the time limit counts from here)
    const unsigned long long t_setup = wall_clock64();
    bool use_lds = lds_cols && nR < L_MAXR;
    // a team searches a cluster that is solved out of LDS from the start (every member holds its own copy); a cluster on HBM scratch
    // (shared) is its owner's alone
    bool team = tm.Wg > 1 && use_lds;
    // ... and a cluster on HBM scratch when every member has a copy of that scratch to itself (BlpArgs::tm_sm / tm_ss: prices, usage and
    // marks by measurement node, the per-member tables): the members replicate the HBM dual phase as well, reach the same decision about
    // reduced-cost fixing (same prices, same feasible point), rebuild the same LDS problem or -- if the survivors do not fit -- share the
    // branch and bound on HBM.  (Round 3 first handed the owner's reduced problem over to waiting members; a cluster that could not be
    // reduced -- 44 targets, 7 831 columns: 14 900 nodes at 0.25 ms -- stayed with one workgroup for 3.7 s.)
    const bool team_hbm = tm.Wg > 1 && !use_lds && a.tm_sm > 0 && K <= TEAM_SEL;
    if (tm.q > 0 && !team && !team_hbm) return;
    const int32_t* final_sel = nullptr;      // team search: the best member's selection (global columns), read by the last finisher
    int nHl = nH;      // columns of the LDS store (fewer than nH after a reduction)
    double ub_reduced = DINF;
    // every member of a team files what it found (global columns); the LAST one to finish takes the best of all -- value, then the lowest
    // member -- and goes on to the cluster's epilogue (true), the others are done (false).  Nobody waits.
    double best_io = DINF;      // (team_file: the value of the selection the last finisher goes on with)
    auto team_file = [&](const Team& tmm, int tidx, int Kk, auto sel_of, double ubv, int& st_io, int& it_io, int& nd_io, const int32_t*& fsel) -> bool {
        TeamResult* res = a.team_res + (size_t)tidx * TEAM_W;
        TeamResult& me = res[tmm.q];
        for (int k = tid; k < Kk; k += BLP_THREADS) me.sel[k] = sel_of(k);
        if (tid == 0) { me.ub = ubv; me.status = st_io; me.nodes = nd_io; me.iters = it_io; }
        __threadfence();
        __syncthreads();
        if (tid == 0) r->i[0] = atomicAdd(&a.team_state[tidx].done, 1);
        __syncthreads();
        const int before = r->i[0];
        __syncthreads();
        if (before != tmm.W - 1) return false;
        __threadfence();
        if (tid == 0) {
            int bq = 0, any_limit = 0, nsum = 0, itmax = 0;
            double bub = DINF;
            for (int q = 0; q < tmm.W; ++q) {
                const double u = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&res[q].ub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const int st = __hip_atomic_load(&res[q].status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nsum += __hip_atomic_load(&res[q].nodes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int itq = __hip_atomic_load(&res[q].iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                itmax = itq > itmax ? itq : itmax;
                any_limit |= (st == MHT_BLP_NODE_LIMIT);
                if (u < bub) { bub = u; bq = q; }      // (ties: the lowest member)
            }
            r->i[0] = bq; r->i[1] = any_limit; r->i[2] = nsum; r->i[3] = itmax; r->d[0] = bub;
        }
        __syncthreads();
        fsel = res[r->i[0]].sel;
        best_io = r->d[0];
        st_io = r->i[1] ? MHT_BLP_NODE_LIMIT : MHT_BLP_BRANCHED;
        nd_io = r->i[2];
        it_io = r->i[3];
        __syncthreads();
        return true;
    };
    // a team that spans the devices of a cluster-sharded step: this device's best selection goes to its slot of the exchange block (BlpArgs::shard_team:
    // [device][team slot][XT_WORDS] int32, -1 = empty; the value as three non-negative chunks of its order-preserving key so that the exchange's
    // element-wise MAX is a gather), NOT to sel_rel -- the devices' selections of one cluster must not mix; shard_team_resolve_kernel picks the winner
    const bool xteam = tm.Wg != tm.W && a.shard_team != nullptr && team_idx >= 0 && K <= TEAM_SEL;
    auto xteam_out = [&](auto sel_of, double value) {
        int32_t* o = a.shard_team + ((size_t)a.shard_i * TEAM_MAX + team_idx) * XT_WORDS;
        for (int k = tid; k < K; k += BLP_THREADS) { const int h = sel_of(k); a.sel[mem[k]] = h; o[4 + k] = h - a.tchild[mem[k]]; }
        if (tid == 0) {
            const unsigned long long key = enum_key(value);
            o[0] = (int32_t)(key >> 42); o[1] = (int32_t)((key >> 21) & 0x1fffffu); o[2] = (int32_t)(key & 0x1fffffu); o[3] = K;
        }
    };
    if (!use_lds) {
        // ---- HBM scratch: the same solver, generic column access ----------------------------------------------------
        GStore gs;
        gs.a = &a; gs.mem = mem; gs.uw = uw; gs.UW = UW; gs.PD = a.PD; gs.cap = (size_t)a.cap;
        const size_t om = team_hbm ? (size_t)tm.q * a.tm_sm : 0, os = (team_hbm ? (size_t)tm.q * a.tm_ss : 0) + (size_t)slot;      // this member's copies
        gs.pu = a.u + om; gs.pusage = a.usage + om; gs.pmark = a.mark + om;
        gs.ck = small_k;      // (colb / gbase were filled above)
        gs.cbL = s.gbase; gs.ceL = s.lix;
        if (gs.ck) {
            for (int k = tid; k < K; k += BLP_THREADS) s.lix[k] = s.gbase[k] + (s.colb[k + 1] - s.colb[k]);
            __syncthreads();
        }
        gs.best_h = a.best_h + os; gs.best_rc = a.best_rc + os; gs.ub_sel = a.bb_best + os; gs.ch = a.bb_ch + os;
        gs.cst = a.bb_cost + os; gs.uus = a.bb_uused + os; gs.lrc = a.bb_last_rc + os; gs.lix = a.bb_last_idx + os;
        gs.rest = a.bb_rest + os; gs.mn = a.bb_min + os;
        gs.for_rows([&](int m) { gs.u(m) = 0.0; gs.usage(m) = 0; gs.mark(m) = 0; });      // (usage / marks are zero between uses; a member's copy may never have been touched)
        __threadfence_block();
        __syncthreads();
        double ub = DINF;
        solve_core(a, gs, K, r, status, iters, nodes, stamp, ub, team_hbm ? tm : Team(0, 1, nullptr));
        ub_reduced = ub;
#ifdef MHT_BLP_TRACE
        if (tid == 0) printf("[blp] cluster %d K=%d nH=%d: HBM phase status %d iters %d nodes %d, %.2f ms (setup %.2f)\n", c, K, nH, status, iters, nodes,
                             1e-5 * (double)(wall_clock64() - t_begin), 1e-5 * (double)(t_setup - t_begin));
#endif
        if (status == MHT_BLP_REDUCE) {
            // ---- reduced-cost fixing left few enough columns: rebuild the cluster from them in LDS (see reducible()) -------
            // (the rebuild re-uses the LDS tables: the column ranges are taken from global memory from here on)
            gs.ck = false;
            if (tid == 0) {
                int acc = 0;
                for (int k = 0; k < K; ++k) { s.colb[k] = acc; acc += gs.lix[k]; }
                s.colb[K] = acc;
                s_nH = acc;
            }
            for (int w = tid; w < UW; w += BLP_THREADS) uw[w] = 0ull;
            __syncthreads();
            nHl = s_nH;
            const int lane = tid & 63, wave = tid >> 6;
            for (int k = 0; k < K; ++k) {      // survivors of member k, ascending column order
                const double thr = gs.mn[k];
                int run = 0;
                for (int base = gs.col_begin(k); base < gs.col_end(k); base += BLP_THREADS) {
                    const int h = base + tid;
                    const bool keep = h < gs.col_end(k) && reduced_cost(gs, h) <= thr;
                    const unsigned long long bal = __ballot(keep);
                    if (lane == 0) r->i[wave] = __popcll(bal);
                    __syncthreads();
                    int off = run;
                    for (int w = 0; w < wave; ++w) off += r->i[w];
                    int tot = 0;
                    for (int w = 0; w < BLP_THREADS / 64; ++w) tot += r->i[w];
                    if (keep) {
                        const int pos = s.colb[k] + off + __popcll(bal & ((1ull << lane) - 1ull));
                        s.gcolL[pos] = h;
                        if (h == gs.ub_sel[k]) s.ub_sel[k] = pos;      // the incumbent of the HBM phase survives by construction
                        s.membL[pos] = (unsigned short)k;
                        s.costL[pos] = a.cost[h];
                        for (int d = 0; d < 8; ++d) {
                            const int e = d < a.PD ? gs.ent(d, h) : -1;
                            s.entL[pos * 8 + d] = (unsigned short)(e < 0 ? 0xffff : e);      // global node id for now
                            if (e >= 0) atomicOr(&uw[e >> 6], 1ull << (e & 63));
                        }
                    }
                    run += tot;
                    __syncthreads();
                }
            }
            row_prefix();
            nR = s_nR;
            s.reduced = true;
            s.nH = nHl;
            use_lds = true;
            team = team_hbm;      // (every member rebuilt the same problem from its own copy of the HBM phase: the search is shared from here)
        } else {
            // solved on HBM scratch: by this workgroup alone, or -- a team -- every member files what it found and the last one finishes
            if (team_hbm && status == MHT_BLP_CERTIFIED) {
                if (tm.q > 0) return;      // (the dual phase is deterministic: every member holds the same certificate, the owner finishes)
            } else if (team_hbm) {
                if (!team_file(tm, team_idx, K, [&](int k) { return gs.ub_sel[k]; }, ub, status, iters, nodes, final_sel)) return;
            }
            stamp[4] = wall_clock64();
            if (xteam) {
                xteam_out([&](int k) { return final_sel ? __hip_atomic_load(const_cast<int32_t*>(final_sel) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gs.ub_sel[k]; },
                          final_sel ? best_io : ub);
            } else
            for (int k = tid; k < K; k += BLP_THREADS) {
                const int h = final_sel ? __hip_atomic_load(const_cast<int32_t*>(final_sel) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gs.ub_sel[k];
                a.sel[mem[k]] = h;
                if (a.sel_rel) a.sel_rel[mem[k]] = h - a.tchild[mem[k]];
                if (a.t_alive) gs.ch[k] = finish_target(a, mem[k], h, pre_ok ? pre : load_target_cr(a, mem[k], cr), true);
            }
            if (!xteam) prune_members(a, mem, K, gs.ch);
        }
    }
    if (use_lds) {
        // ---- LDS-resident solve: global node ids -> dense local row ids (LDS only) --------------------------------
        s.nR = nR;
        for (int m = tid; m <= nR; m += BLP_THREADS) { s.uL[m] = 0.0; s.usageL[m] = 0; s.markL[m] = 0; }   // incl. dummy row nR
        for (int h = tid; h < nHl; h += BLP_THREADS) {       // one 16-byte record per thread: 8 independent rank look-ups
            const Rows8 g8 = rows_of(s, h);
            unsigned o[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const unsigned e = g8.e[d];
                o[d] = (e == 0xffffu) ? (unsigned)nR
                                      : (unsigned)(s_wbase[e >> 6] + __popcll(uw[e >> 6] & ((1ull << (e & 63)) - 1ull)));
            }
            reinterpret_cast<uint4*>(s.entL)[h] = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
        }
        __syncthreads();
        double ub = s.reduced ? ub_reduced : DINF;      // (a reduced cluster brings the HBM phase's incumbent along, see the rebuild)
#ifdef MHT_BLP_TRACE
        if (tid == 0 && s.reduced) printf("[blp] cluster %d: rebuilt with %d columns %d rows at %.2f ms\n", c, nHl, nR, 1e-5 * (double)(wall_clock64() - t_begin));
#endif
        solve_core(a, s, K, r, status, iters, nodes, stamp, ub, team ? tm : Team(0, 1, nullptr));
#ifdef MHT_BLP_TRACE
        if (tid == 0 && s.reduced) printf("[blp] cluster %d: LDS phase status %d iters %d nodes %d at %.2f ms\n", c, status, iters, nodes, 1e-5 * (double)(wall_clock64() - t_begin));
#endif
        if (team && status == MHT_BLP_CERTIFIED) {
            if (tm.q > 0) return;      // the dual phase is deterministic: every member holds the same certificate, the owner finishes
        } else if (team) {
            if (!team_file(tm, team_idx, K, [&](int k) { return s.to_global(s.ub_sel[k]); }, ub, status, iters, nodes, final_sel)) return;
        }
        stamp[4] = wall_clock64();
        int rf_mine = -1;
        if (xteam) {      // (a sharded step solves only: t_alive is null, the per-target end of the scan follows the exchange)
            xteam_out([&](int k) { return final_sel ? __hip_atomic_load(const_cast<int32_t*>(final_sel) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s.to_global(s.ub_sel[k]); },
                      final_sel ? best_io : ub);
        } else
        for (int k = tid; k < K; k += BLP_THREADS) {
            const int h = final_sel ? __hip_atomic_load(const_cast<int32_t*>(final_sel) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : s.to_global(s.ub_sel[k]);
            a.sel[mem[k]] = h;
            if (a.sel_rel) a.sel_rel[mem[k]] = h - a.tchild[mem[k]];
            if (a.t_alive) {
                const TgtPre q = pre_ok ? pre : load_target_cr(a, mem[k], cr);
                s.ch[k] = finish_target(a, mem[k], h, q, true, &rf_mine);      // (K <= BLP_THREADS in LDS: one member per thread)
                s.ub_sel[k] = q.j;             // the solver's tables are free now: prune depth, survivor count, first survivor
                s.lix[k] = 0;
                s.best_h[k] = 0x7fffffff;
            }
        }
        if (a.t_alive && s.reduced) {      // (the LDS store holds a subset of the children: generic sweep over the members' ranges)
            __syncthreads();
            prune_members(a, mem, K, s.ch);
        } else if (a.t_alive) {
            // surviving leaf ranges: one thread per column (= child), one ancestor look-up each, LDS counters per member
            __syncthreads();
            for (int h = tid; h < nH; h += BLP_THREADS) {
                const int m = s.membL[h];
                const int key = s.ch[m];
                if (key == KEY_DEAD) continue;
                const int g = s.gbase[m] + (h - s.colb[m]);
                const bool sv = key == KEY_ALL || (anc_pf ? s.gcolL[h] : a.apath[(size_t)g * a.pds + (s.ub_sel[m] - 1)]) == key;
                if (sv) { atomicAdd(&s.lix[m], 1); atomicMin(&s.best_h[m], g); }
            }
            __syncthreads();
            for (int k = tid; k < K; k += BLP_THREADS) {
                a.t_count[mem[k]] = s.lix[k];
                a.t_firstsurv[mem[k]] = s.best_h[k];
                blp_publish(a, mem[k], s.ch[k], s.ub_sel[k], rf_mine, s.lix[k], s.best_h[k]);
            }
        }
    }
    if (tid == 0 && a.dbg && dbg_bx >= 0 && dbg_bx < 3900 && tm.q == 0) {
        unsigned long long* g = a.dbg + 32 + (size_t)dbg_bx * 16;
        g[8] = t_begin; g[9] = t_setup; g[10] = stamp[4]; g[11] = wall_clock64();
    }
    if (tid == 0) {
        a.cl_status[c] = status;
        a.cl_iters[c] = iters;
        a.cl_nodes[c] = nodes;
        if (a.cl_time) {
            a.cl_time[8 * c] = (int)(t_setup - t_begin);
            a.cl_time[8 * c + 1] = (int)(wall_clock64() - t_begin);
            for (int q = 1; q <= 4; ++q) a.cl_time[8 * c + 1 + q] = (int)(stamp[q] - t_begin);
            a.cl_time[8 * c + 6] = (int)(clock64() - c_begin);
            a.cl_time[8 * c + 7] = (int)stamp[5];      // ticks spent in enumerate_small
        }
    }
    __threadfence_block();
    __syncthreads();
}

constexpr size_t RED_SLOT = BLP_THREADS <= 512 ? 512 : 1024;      // LDS bytes reserved for the reduction scratch (multiple of 16)
static_assert(sizeof(Red) <= RED_SLOT, "Red must fit its LDS slot");
static size_t blp_lds_bytes(int cap_h, int cap_r, int cap_k, int cap_uw) {
    const size_t kpad = (size_t)cap_k + 4;
    return (size_t)cap_uw * 8 + RED_SLOT + (size_t)cap_uw * 4 + 16 + (size_t)cap_h * 16 + (size_t)cap_h * 8 * 2 + (size_t)cap_r * 8 + 7 * kpad * 8 +
           2 * (size_t)cap_r * 4 + 6 * kpad * 4 + (size_t)cap_h * 2 * 2 + ENUM_LDS + (size_t)cap_h * 16 + (size_t)cap_h * 4;
}

// ---- clusters from the grow launch's union-find (BlpArgs::uf_epoch; mht_kernels.h: FDyn::uf_epoch) -------------------------------------
// What a workgroup keeps of the cluster tables it derived (uf_prologue) once the solver's tables overlay them: its own clusters, the
// teams, the single-target clusters of its wavefronts.  Lives behind the solver's LDS (BLP_UF_PERSIST bytes).
struct UfPersist {
    int nT, nC, nMulti, nSingle, nTeam, my_t, pad[2];
    ClRef own[4];                          // multi-target clusters bx, bx + gx, ... (by rank among the multi-target clusters)
    ClRef team[TEAM_MAX];
    unsigned short single[BLP_THREADS / 64][8];      // wavefront w: targets alone in their cluster, i = gw + j * nw
};
constexpr size_t BLP_UF_PERSIST = (sizeof(UfPersist) + 15) & ~(size_t)15;
constexpr int UF_SPEC = 4;      // parent words per thread fetched together with the target count (1 024 targets in the first round trip)
// LDS of the prologue for T targets (overlaid by the solver afterwards): see the carve below
__host__ __device__ constexpr size_t uf_prologue_bytes(size_t T) { return ((T + 7) & ~(size_t)7) * 18 + 256; }

// Every workgroup of the launch derives the cluster tables for itself, in LDS, from the parents the grow launch left (one global round
// trip, seven barriers): labels = roots = smallest members, member counts, ONE block scan for cluster index / member offset / rank among
// the multi-target, single-target and team-sized clusters, ascending member lists (slot by LDS atomic, rank by counting the smaller
// members -- as cluster_kernel).  Same tables, same order as the clustering kernel's; workgroup `writer` files them in global memory
// for the commit's statistics and the host.  Returns false for a void scan.
// the prologue's one global round trip, issued at the very start of the kernel: status word, target count and -- speculatively, the first
// UF_SPEC x 256 of them -- the parents (the solver's own scalar set-up, ~2 us of argument loads, runs while they are in flight)
struct UfFetch { int s_over, nT; unsigned long long nif; unsigned long long pw[UF_SPEC]; };
// what the prefetch needs of the argument block (the first arguments the kernel loads; everything else follows BEHIND the issue of these
// loads, see blp_uf_kernel)
struct UfHead { const DevStatus* status; const int32_t* nT_dev; const unsigned long long* ni_flag; const unsigned long long* uf_parent; int uf_cap, uf_ovl; };
// Every load is a VECTOR load through an index the compiler cannot see through (zero): a load it knows to be uniform becomes
// global_load + s_waitcnt vmcnt(0) + v_readfirstlane on the spot -- three dependent round trips in front of the parents' one (seen in the ISA
// of the round-4 build).  Like this the seven loads leave together and nothing waits for them before the prologue's first use.
__device__ __forceinline__ UfFetch uf_prefetch(const UfHead& a) {
    UfFetch f;
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    const int32_t* sp = a.status ? &a.status->overflow : a.nT_dev;
    const int so = sp[z];
    f.nT = a.nT_dev[z];
    const unsigned long long nif = (a.uf_ovl ? a.ni_flag : a.uf_parent)[z];
#pragma unroll
    for (int q = 0; q < UF_SPEC; ++q) {
        const int t = (int)threadIdx.x + q * BLP_THREADS;
        f.pw[q] = a.uf_parent[((t < a.uf_cap) ? t : 0) + z];
    }
    f.s_over = a.status ? so : 0;
    f.nif = a.uf_ovl ? nif : 0ull;
    return f;
}
__device__ __forceinline__ bool uf_prologue(const BlpArgs& a, const UfFetch& fe, unsigned char* lds, UfPersist* ps, const int bx, const int gx, int& my_t_out, TgtPre& pre_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define UF_STAMP(k) do { if (a.dbg && tid == 0 && bx < 3900) a.dbg[32 + (size_t)bx * 16 + (k)] = wall_clock64(); } while (0)
    UF_STAMP(0);
    const int s_over = __builtin_amdgcn_readfirstlane(fe.s_over);
    const int nT = __builtin_amdgcn_readfirstlane(fe.nT);
    const unsigned long long nif = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(fe.nif >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)fe.nif);
    // (an overlapping grow launch redid the union-find under the alternative epoch if a target had died in the scan before)
    const unsigned epoch = (a.uf_ovl && (unsigned)nif == (a.uf_epoch >> 1) - 1u && ((nif >> 32) & 1ull)) ? (a.uf_epoch | 1u) : a.uf_epoch;
    unsigned long long pw[UF_SPEC];
#pragma unroll
    for (int q = 0; q < UF_SPEC; ++q) pw[q] = fe.pw[q];
    if (s_over) return false;
    const int Tp = (nT + 7) & ~7;
    unsigned short* par = reinterpret_cast<unsigned short*>(lds);            // [Tp] union-find parent; later: cluster index by head
    unsigned short* lab = par + Tp;                                          // [Tp] root = smallest member = label
    unsigned short* hp = lab + Tp;                                           // [Tp] by head: offset of the member list
    unsigned short* tmp = hp + Tp;                                           // [Tp] members by slot
    unsigned short* ms = tmp + Tp;                                           // [Tp] members ascending (cl_members)
    unsigned short* ml = ms + Tp;                                            // [Tp / 2] heads of the multi-target clusters
    unsigned short* sl = ml + Tp / 2;                                        // [Tp] targets alone in their cluster
    int* cnt = reinterpret_cast<int*>(sl + Tp);                              // [Tp] by head: members | slots handed out << 16
    unsigned long long* s_w = reinterpret_cast<unsigned long long*>(cnt + Tp);   // [4] wave totals of the scan, [4] totals
    int* s_tl = reinterpret_cast<int*>(s_w + 8);                             // [TEAM_MAX] heads of the team-sized clusters
#pragma unroll
    for (int q = 0; q < UF_SPEC; ++q) {
        const int t = tid + q * BLP_THREADS;
        if (t < nT) { par[t] = (unsigned short)(((unsigned)(pw[q] >> 32) == epoch) ? 0xffffffffu - (unsigned)pw[q] : (unsigned)t); cnt[t] = 0; }
    }
    for (int t = tid + UF_SPEC * BLP_THREADS; t < nT; t += BLP_THREADS) {
        const unsigned long long w = a.uf_parent[t];
        par[t] = (unsigned short)(((unsigned)(w >> 32) == epoch) ? 0xffffffffu - (unsigned)w : (unsigned)t);
        cnt[t] = 0;
    }
    __syncthreads();
    UF_STAMP(1);
    for (int t = tid; t < nT; t += BLP_THREADS) {
        int r = t, p = par[r];
        while (p != r) { r = p; p = par[r]; }
        lab[t] = (unsigned short)r;
        atomicAdd(&cnt[r], 1);
    }
    __syncthreads();
    UF_STAMP(2);
    // ONE block scan over the targets in index order (thread = a run of E consecutive targets): heads -> cluster index, their sizes ->
    // member offset, ranks among the multi-target / single-target / team-sized clusters, packed 14 + 13 + 14 + 14 + 9 bits (max_targets 8 192:
    // at most 8 192 clusters, 4 096 of them with two or more targets, 341 with TEAM_MIN_K or more -- no field can wrap)
    const int E = (nT + BLP_THREADS - 1) / BLP_THREADS;
    auto pack_of = [&](int t) -> unsigned long long {
        if (t >= nT || lab[t] != t) return 0ull;
        const unsigned long long K = (unsigned long long)cnt[t];
        return 1ull | ((K >= 2 ? 1ull : 0ull) << 14) | ((K == 1 ? 1ull : 0ull) << 27) | (K << 41) | ((K >= (unsigned long long)TEAM_MIN_K ? 1ull : 0ull) << 55);
    };
    unsigned long long mine = 0ull;
    for (int e = 0; e < E; ++e) mine += pack_of(tid * E + e);
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    UF_STAMP(3);
    unsigned long long run = incl - mine, total = 0ull;
#pragma unroll
    for (int w = 0; w < BLP_THREADS / 64; ++w) { const unsigned long long v = s_w[w]; if (w < wave) run += v; total += v; }
    const int nC = (int)(total & 0x3fffu), nMulti = (int)((total >> 14) & 0x1fffu), nSingle = (int)((total >> 27) & 0x3fffu);
    const int nTeamAll = (int)(total >> 55), nTeam = (a.team_list && nTeamAll > 0) ? (nTeamAll < TEAM_MAX ? nTeamAll : TEAM_MAX) : 0;
    for (int e = 0; e < E; ++e) {
        const int t = tid * E + e;
        const unsigned long long pk = pack_of(t);
        if (pk) {
            const int c = (int)(run & 0x3fffu), mr = (int)((run >> 14) & 0x1fffu), sr = (int)((run >> 27) & 0x3fffu), p0 = (int)((run >> 41) & 0x3fffu), tr = (int)(run >> 55);
            const int K = cnt[t];
            par[t] = (unsigned short)c;       // (the parents are dead: cluster index by head)
            hp[t] = (unsigned short)p0;
            if (K >= 2) ml[mr] = (unsigned short)t; else sl[sr] = (unsigned short)t;
            if (K >= TEAM_MIN_K && tr < TEAM_MAX) s_tl[tr] = t;
            run += pk;
        }
    }
    __syncthreads();
    UF_STAMP(4);
    // Member lists.  A workgroup usually needs ONE: that of its own cluster -- wavefront 0 collects it in ascending order with ballots, no
    // barrier, while the others file the single-target clusters.  All lists at once (a slot by LDS atomic, then the ascending rank by
    // counting the smaller members, as cluster_kernel) only where they are needed: the workgroup that files the tables in global
    // memory, teams (every member needs the team clusters' lists), several clusters per workgroup, a cluster of more than 64 targets.
    const int writer = nMulti < gx - 1 ? nMulti : gx - 1;      // (the first workgroup without an ILP of its own)
    const int own_root = bx < nMulti ? (int)ml[bx] : 0;
    const int own_K = bx < nMulti ? (cnt[own_root] & 0xffff) : 0;
    const bool full = bx == writer || nTeam > 0 || bx + gx < nMulti || own_K > 64;      // (uniform in the workgroup)
    auto ref_of = [&](int h) { return ClRef{(int)par[h], (int)hp[h], cnt[h] & 0xffff, h}; };
    int32_t* gmem = const_cast<int32_t*>(a.cl_members);
    if (tid == 0) { ps->nT = nT; ps->nC = nC; ps->nMulti = nMulti; ps->nSingle = nSingle; ps->nTeam = nTeam; ps->my_t = -1; }
    if (tid >= 64 && tid < 64 + BLP_THREADS / 64 * 8) {
        const int w = (tid - 64) >> 3, j = (tid - 64) & 7;
        const int i = (gx - 1 - bx) * (BLP_THREADS / 64) + w + j * gx * (BLP_THREADS / 64);
        ps->single[w][j] = (i < nSingle) ? sl[i] : (unsigned short)0xffff;
    }
    int my_t = -1;
    if (full) {
        for (int t = tid; t < nT; t += BLP_THREADS) {
            const int h = lab[t];
            const int K = cnt[h] & 0xffff;
            if (K == 1) ms[hp[h]] = (unsigned short)t;
            else tmp[hp[h] + (atomicAdd(&cnt[h], 1 << 16) >> 16)] = (unsigned short)t;
        }
        __syncthreads();
        UF_STAMP(5);
        for (int t = tid; t < nT; t += BLP_THREADS) {
            const int h = lab[t];
            const int K = cnt[h] & 0xffff, base = hp[h];
            if (K == 1) continue;
            int rank = 0;
            for (int i = 0; i < K; ++i) rank += (tmp[base + i] < t) ? 1 : 0;
            ms[base + rank] = (unsigned short)t;
        }
        __syncthreads();
        UF_STAMP(6);
        // the member lists of the workgroup's clusters (and of the teams' clusters: the same values from every member of a team) go to
        // cl_members, where the solver reads them as before
        for (int q = 0; q < 4; ++q) {
            const int i = bx + q * gx;
            if (i >= nMulti) break;
            const ClRef cr = ref_of(ml[i]);
            if (tid == 0) ps->own[q] = cr;
            for (int k = tid; k < cr.K; k += BLP_THREADS) gmem[cr.p0 + k] = ms[cr.p0 + k];
        }
        for (int q = 0; q < nTeam; ++q) {
            const ClRef cr = ref_of(s_tl[q]);
            if (tid == 0) ps->team[q] = cr;
            if (bx >= nMulti) for (int k = tid; k < cr.K; k += BLP_THREADS) gmem[cr.p0 + k] = ms[cr.p0 + k];      // (members of teams; the owner wrote it above)
        }
        if (bx == writer) {
            int32_t* t_label = const_cast<int32_t*>(a.t_label);
            int32_t* cl_ptr = const_cast<int32_t*>(a.cl_ptr);
            int32_t* multi = const_cast<int32_t*>(a.multi_list);
            int32_t* single = const_cast<int32_t*>(a.single_list);
            int32_t* counts = const_cast<int32_t*>(a.counts);
            for (int t = tid; t < nT; t += BLP_THREADS) {
                const int h = lab[t];
                t_label[t] = h;
                if (a.t_cluster) a.t_cluster[t] = par[h];
                gmem[t] = ms[t];
                if (h == t) cl_ptr[par[h]] = hp[h];
            }
            for (int i = tid; i < nMulti; i += BLP_THREADS) multi[i] = par[ml[i]];
            for (int i = tid; i < nSingle; i += BLP_THREADS) single[i] = sl[i];
            if (a.team_list && tid < nTeam) const_cast<int32_t*>(a.team_list)[tid] = par[s_tl[tid]];
            if (tid == 0) {
                cl_ptr[nC] = nT;
                counts[0] = nC; counts[1] = nMulti; counts[2] = nSingle; counts[3] = 0; counts[4] = 0; counts[5] = nTeam;
                // (the counters of the other parity's status word, which the cluster kernel cleared for the scan after this one, are zeroed by
                // that word's commit: the next scan's grow launch may be running already)
            }
            // slots beyond the table: "nobody" for the next scan's grow launch, whose grid is sized by the host's upper bound
            if (a.rec0) for (int t = nT + tid; t < a.pub_ub; t += BLP_THREADS) __hip_atomic_store(&a.rec0[t], tgt_rec(a.pub_scan, 0, 0, 0, 0, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.alloc_reset && tid >= 64 && tid < 64 + FG_REGIONS) a.alloc_reset[(tid - 64) * 32] = 0u;
        }
        if (bx < nMulti && tid < own_K) my_t = ms[hp[own_root] + tid];
    } else if (wave == 0 && bx < nMulti) {
        const int p0 = hp[own_root];
        int n = 0;
        for (int t0 = 0; t0 < nT && n < own_K; t0 += 64) {
            const int t = t0 + lane;
            const bool m = t < nT && lab[t] == own_root;
            const unsigned long long bal = __ballot(m);
            if (m) ms[p0 + n + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)t;
            n += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) ps->own[0] = ref_of(own_root);
        if (lane < own_K) { my_t = ms[p0 + lane]; gmem[p0 + lane] = my_t; }
    }
    // member `tid` of the workgroup's first cluster: its record is on its way while the solver sets itself up
    if (my_t >= 0) { pre_out = load_target(a, my_t); pre_out.lab = own_root; }
    __threadfence_block();
    __syncthreads();
    UF_STAMP(7);
    my_t_out = my_t;
    return true;
}

template <bool UF> __device__ __forceinline__ void blp_singles(const BlpArgs& a, const int bx, const int gx, const int nSingle, const UfPersist* ps = nullptr);
__device__ __forceinline__ void blp_stamp_begin(const BlpArgs& a, int bx);
template <bool UF = false>
__device__ __forceinline__ void blp_body(const BlpArgs& a, unsigned char* lds, const int bx, const int gx, UfPersist* ps = nullptr, const UfFetch* fe = nullptr) {      // workgroup bx of gx
    unsigned long long* uw = reinterpret_cast<unsigned long long*>(lds);           // [cap_uw]
    Red* red = reinterpret_cast<Red*>(lds + (size_t)a.cap_uw * 8);                   // sizeof(Red) padded to RED_SLOT
    if (!UF && a.dbg && threadIdx.x == 0 && bx < 3900) a.dbg[32 + (size_t)bx * 16 + 12] = wall_clock64();
    if (!UF && a.status && a.status->overflow) return;
    // tier 2: what the first launch left (big_list; the single-target clusters went with that launch) -- the same staged loop over that list
    const bool rest = !UF && a.tier == 2;
    const int32_t* work = rest ? a.big_list : a.multi_list;
    int nMulti = UF ? 0 : (rest ? *a.big_count : a.counts[1]), nSingle = UF ? 0 : (rest ? 0 : a.counts[2]);
    int my_t = -1;
    TgtPre my_pre_v = {};
    const TgtPre* my_pre = UF ? &my_pre_v : nullptr;
    // (shard_n > 1: the clusters of one tracker are spread over shard_n devices that hold identical forests -- a multi-target cluster is
    // solved on the device cl_owner names (placed by size by the cluster kernel), a single-target cluster where its target index says
    // so; see blp_epilogue_kernel)
    // teams (mht_kernels.h: TEAM_*): the launch's workgroups without a cluster of their own (block index >= nMulti) are dealt out to
    // the clusters of the team list; member 0 of a team is the workgroup that owns the cluster anyway
    // (a cluster-sharded step with an exchange block for them: the clusters of the team list are searched by every device's team together -- the
    // subtrees dealt out over all members of all devices --, also when a device has no idle workgroup to add to its own member)
    const bool xteams = a.shard_n > 1 && a.shard_team != nullptr;
    bool teams_on = a.team_list && a.tier != 1 && (xteams || (a.shard_n <= 1 && gx > nMulti));
    int nTeam = teams_on ? (UF ? 0 : a.counts[5]) : 0;
    int nIdle = gx - nMulti;
    auto team_W = [&](int ti) { const int w = nIdle > ti ? 1 + (nIdle - ti + nTeam - 1) / nTeam : 1; return w < TEAM_W ? w : TEAM_W; };
    // (ONE call site of the solver: the workgroup's own clusters first, then the single-target clusters, then -- if it has no cluster
    // of its own -- its share of a team's search)
    int own_i = bx, own_q = 0;
    for (int stage = UF ? -1 : 0; stage < 3; ) {
        ClRef cr = ClRef{-1, 0, 0, -1};
        int ti = -1, mt = -1;
        Team tm = Team(0, 1, nullptr);
        if (UF && stage == -1) {
            // clusters from the grow launch's union-find: the tables first (INSIDE the staged loop: what the compiler hoists in front of the
            // loop -- the solver's argument loads and address arithmetic, ~2 us -- then runs while the parents are on their way)
            stage = 0;
            if (!uf_prologue(a, *fe, lds, ps, bx, gx, my_t, my_pre_v)) return;      // (void scan)
            blp_stamp_begin(a, bx);
            if (a.dbg && threadIdx.x == 0 && bx < 3900) a.dbg[32 + (size_t)bx * 16 + 12] = wall_clock64();
            nMulti = ps->nMulti; nSingle = ps->nSingle;
            teams_on = a.team_list && a.tier != 1 && (xteams || (a.shard_n <= 1 && gx > nMulti));
            nTeam = teams_on ? ps->nTeam : 0;
            nIdle = gx - nMulti;
            continue;
        } else if (stage == 0) {
            if (own_i >= nMulti) { stage = 1; continue; }
            if (UF) { cr = ps->own[own_q]; mt = own_q == 0 ? my_t : -1; ++own_q; }
            else cr = cl_ref(a, work[own_i]);
            own_i += gx;
            if (nTeam > 0 && cr.K >= TEAM_MIN_K)
                for (int q = 0; q < nTeam; ++q) if ((UF ? ps->team[q].c : a.team_list[q]) == cr.c) ti = q;
            const bool xt = xteams && ti >= 0 && cr.K <= TEAM_SEL;      // (a file holds TEAM_SEL selections: a larger cluster stays with its owner)
            if (a.shard_n > 1 && !xt && (a.cl_owner ? a.cl_owner[cr.c] : cr.c % a.shard_n) != a.shard_i) continue;
            if (xt) tm = Team(0, team_W(ti), &a.team_state[ti].gub, a.shard_i, team_W(ti) * a.shard_n);      // (member q of device i: number q x devices + i)
            else if (ti >= 0 && team_W(ti) > 1) tm = Team(0, team_W(ti), &a.team_state[ti].gub);
            else ti = -1;
        } else if (stage == 1) {
            stage = 2;
            blp_singles<UF>(a, bx, gx, nSingle, ps);
            continue;
        } else {
            stage = 3;
            if (!(nTeam > 0 && bx >= nMulti)) break;      // a workgroup without a cluster: member of a team
            const int j = bx - nMulti, q = 1 + j / nTeam;
            ti = j % nTeam;
            if (q >= team_W(ti)) break;
            __syncthreads();      // (the wavefronts of this workgroup are done with the single-target clusters)
            cr = UF ? ps->team[ti] : cl_ref(a, a.team_list[ti]);
            tm = (xteams && cr.K <= TEAM_SEL) ? Team(q, team_W(ti), &a.team_state[ti].gub, q * a.shard_n + a.shard_i, team_W(ti) * a.shard_n) : Team(q, team_W(ti), &a.team_state[ti].gub);
        }
        if (a.dbg && threadIdx.x == 0 && bx < 3900 && stage == 0) a.dbg[32 + (size_t)bx * 16 + 13] = wall_clock64();
        solve_cluster(a, cr, uw, red, lds + (size_t)a.cap_uw * 8 + RED_SLOT, tm, ti, mt, mt >= 0 ? my_pre : nullptr, bx);
    }
}

// targets alone in their cluster: one wavefront each, dealt out from the END of the grid (the workgroups without an ILP)
template <bool UF>
__device__ __forceinline__ void blp_singles(const BlpArgs& a, const int bx, const int gx, const int nSingle, const UfPersist* ps) {
    // targets alone in their cluster: min cumulativeNLLR, `<=` => the LAST minimal leaf wins (pyTarget.py:449)
    const int lane = threadIdx.x & 63;
    const int gw = (gx - 1 - bx) * (BLP_THREADS / 64) + (threadIdx.x >> 6);
    int j = 0;
    for (int i = gw; i < nSingle; i += gx * (BLP_THREADS / 64), ++j) {
        const int t = UF ? (int)ps->single[threadIdx.x >> 6][j & 7] : a.single_list[i];
        if (a.shard_n > 1 && t % a.shard_n != a.shard_i) continue;
        TgtPre pre = {};
        int cb, ce;
        if (a.t_alive) { pre = load_target(a, t); if (UF) pre.lab = t; cb = pre.cb; ce = pre.ce; }
        else { cb = a.tchild[t]; ce = a.tcend[t]; }
        double bv = DINF;
        int bi = -1;
        for (int h = cb + lane; h < ce; h += 64) {
            const double v = a.cnllr[h];
            if (a.skip_dead && (a.flags[h] & F_DEAD)) continue;      // (fused away by similar-state pruning)
            if (bi < 0 || v <= bv) { bv = v; bi = h; }
        }
        const int va0 = a.t_alive ? sweep_prefetch(a, pre.j, cb, ce, lane) : -1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (oi >= 0 && (bi < 0 || ov < bv || (ov == bv && oi > bi))) { bv = ov; bi = oi; }
        }
        if (lane == 0) { a.sel[t] = bi; if (a.sel_rel) a.sel_rel[t] = bi - cb; }
        if (a.t_alive) {      // wave-uniform: every lane evaluates the (broadcast) look-ups, lane 0 stores
            int rf = 0;
            const int key = finish_target(a, t, bi, pre, lane == 0, &rf);
            sweep_survivors(a, t, pre.j, cb, ce, key, va0, lane, rf);
        }
    }
}

// ---- light pass (groups of sectors): one WAVEFRONT per multi-target cluster tries the certificate of round 0 ---------------------------
// 96 % of the headline stream's ILPs are certified in the dual phase's first round, at zero prices: every member's cheapest column
// (lowest index among equals) and no measurement node used by two of them -- conflict-free minimisers with nothing priced are optimal.
// That test needs no LDS tables and no workgroup: a wavefront reads the members' costs, the eight rows of the K minimisers (lane =
// (member, level)) and compares them; a certified cluster is finished right here (selection, termination, prune decision, surviving
// leaf ranges -- what blp_singles does for a lone target), everything else goes to the list the full solver works through in a second,
// narrow launch (BlpArgs::big_list).  The 155 KB workgroups, one per CU, then carry the few clusters that need them instead of every
// cluster of every sector.  Results are the full solver's: same minimisers, same tie-break, status CERTIFIED after 0 rounds.
constexpr int LIGHT_MAXK = 8;
__device__ __forceinline__ void blp_light(const BlpArgs& a, const int bx, const int gx, const int nMulti) {
    const int lane = threadIdx.x & 63;
    const int gw = bx * (BLP_THREADS / 64) + (threadIdx.x >> 6), nw = gx * (BLP_THREADS / 64);
    for (int i = gw; i < nMulti; i += nw) {
        const int c = a.multi_list[i];
        const int p0 = a.cl_ptr[c], K = a.cl_ptr[c + 1] - p0;
        if (K > LIGHT_MAXK || a.pds != 8 || a.force_hbm) {
            if (lane == 0) a.big_list[atomicAdd(a.big_count, 1)] = c;
            continue;
        }
        // lane k < K holds member k: its target record first (one round trip for all members) ...
        const int tk = a.cl_members[p0 + (lane < K ? lane : 0)];
        const TgtPre pre = load_target(a, tk);
        // ... then every member's cheapest column: the first 64 columns of all members in ONE batch of loads (a member has ~55), the
        // rest in a loop; (value, lowest index) like compute_minimisers
        double v[LIGHT_MAXK];
        int cbk[LIGHT_MAXK], cek[LIGHT_MAXK];
#pragma unroll
        for (int k = 0; k < LIGHT_MAXK; ++k) {
            cbk[k] = __shfl(pre.cb, k < K ? k : 0);
            cek[k] = __shfl(pre.ce, k < K ? k : 0);
            const int h = cbk[k] + lane;
            v[k] = a.cost[(k < K && h < cek[k]) ? h : cbk[0]];
        }
        int best = -1;
#pragma unroll
        for (int k = 0; k < LIGHT_MAXK; ++k) {
            if (k >= K) break;
            const int h0 = cbk[k] + lane;
            double bv = (h0 < cek[k]) ? v[k] : DINF;
            int bi = (h0 < cek[k]) ? h0 : -1;
            for (int h = h0 + 64; h < cek[k]; h += 64) {
                const double w = a.cost[h];
                if (bi < 0 || w < bv) { bv = w; bi = h; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(bv, o);
                const int oi = __shfl_xor(bi, o);
                if (oi >= 0 && (bi < 0 || ov < bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            if (lane == k) best = bi;
        }
        // rows of the minimisers: lane = member * 8 + level; a node used by two different members = conflict
        const int mk = lane >> 3;
        const int hk = __shfl(best, mk < K ? mk : 0);
        int e = -1;
        if (mk < K && hk >= 0) e = a.path[(size_t)hk * 8 + (lane & 7)];
        // (issued with the row loads: what the sweep of the survivors needs of every member's first 64 children)
        int va[LIGHT_MAXK];
#pragma unroll
        for (int k = 0; k < LIGHT_MAXK; ++k) {
            const int jk = __shfl(pre.j, k < K ? k : 0);
            va[k] = (k < K) ? sweep_prefetch(a, jk, cbk[k], cek[k], lane) : -1;
        }
        bool clash = false;
        for (int q = 0; q < K * 8; ++q) {
            const int eq = __shfl(e, q);
            clash = clash || (eq >= 0 && eq == e && (q >> 3) != mk);
        }
        if (__any(clash) || __any(lane < K && best < 0)) {
            if (lane == 0) a.big_list[atomicAdd(a.big_count, 1)] = c;
            continue;
        }
        // certified: every member is finished like a lone target -- lane k its member's termination / prune decision / report row
        // (finish_target is per-thread code), then the survivor sweeps, a wavefront pass per member
        int key = KEY_DEAD;
        if (lane < K) {
            a.sel[tk] = best;
            if (a.sel_rel) a.sel_rel[tk] = best - pre.cb;
            key = finish_target(a, tk, best, pre, true);
        }
#pragma unroll
        for (int k = 0; k < LIGHT_MAXK; ++k) {
            if (k >= K) break;
            sweep_survivors(a, __shfl(tk, k), __shfl(pre.j, k), cbk[k], cek[k], __shfl(key, k), va[k], lane);
        }
        if (lane == 0) { a.cl_status[c] = MHT_BLP_CERTIFIED; a.cl_iters[c] = 0; a.cl_nodes[c] = 0; }
    }
}

// stage stamps of the scan (DevStatus::t, forest only): [2] = start of the first ILP launch, [4] = end of the last ILP workgroup
__device__ __forceinline__ void blp_stamp_begin(const BlpArgs& a, int bx) {
    if (a.status && bx == 0 && threadIdx.x == 0) {
        DevStatus* st = const_cast<DevStatus*>(a.status);
        if (st->t[2] == 0) st->t[2] = wall_clock64();
    }
}
__device__ __forceinline__ void blp_stamp_end(const BlpArgs& a) {
    if (a.status && threadIdx.x == 0) atomicMax(&const_cast<DevStatus*>(a.status)->t[4], (unsigned long long)wall_clock64());
}

__global__ __launch_bounds__(BLP_THREADS) void blp_kernel(const BlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 3900) a.dbg[32 + (size_t)blockIdx.x * 16] = wall_clock64();
    blp_stamp_begin(a, blockIdx.x);
    blp_body(a, lds, blockIdx.x, gridDim.x);
    blp_stamp_end(a);
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 3900) a.dbg[32 + (size_t)blockIdx.x * 16 + 15] = wall_clock64();
}
// clusters from the grow launch's union-find: every workgroup derives the cluster tables for itself first (uf_prologue); no cluster kernel
__global__ __launch_bounds__(BLP_THREADS) void blp_uf_kernel(const BlpArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // The prologue's round trip leaves FIRST, on the six arguments it needs.  The other ~200 argument words (which the compiler loads in
    // ~30 dependent rounds of s_load + spill to VGPR lanes at the kernel's entry, ~2-3 us) are read through a pointer it cannot see
    // through, i.e. behind the issue of those loads: the two overlap instead of adding up.
    const UfFetch fe = uf_prefetch(UfHead{a_in.status, a_in.nT_dev, a_in.ni_flag, a_in.uf_parent, a_in.uf_cap, a_in.uf_ovl});
    typedef const __attribute__((address_space(4))) BlpArgs* KArgP;
    KArgP kp = (KArgP)__builtin_amdgcn_kernarg_segment_ptr();      // (the argument block is the kernel's only argument: offset 0)
    asm volatile("" : "+s"(kp) : : "memory");
    BlpArgs a;
    __builtin_memcpy(&a, kp, sizeof(BlpArgs));
    UfPersist* ps = reinterpret_cast<UfPersist*>(lds + a.uf_lds_off);
    if (a.status && blockIdx.x == 0 && threadIdx.x == 0) const_cast<DevStatus*>(a.status)->t[1] = wall_clock64();      // stage stamp: clustering starts
    // (this launch is ordered behind the scan's grow launch: whoever reads this word -- the scan's initiator on its own queue -- knows that launch is complete)
    if (a.begun && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.begun, (unsigned long long)a.pub_scan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int gx = (int)gridDim.x, pb = (int)blockIdx.x, bx = pb;
    blp_body<true>(a, lds, bx, gx, ps, &fe);
    if (!fe.s_over) blp_stamp_end(a);
    if (a.blp_done) {
        // the next scan's grow launch may be running: its commit waits until every workgroup of this launch has released what it wrote
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            atomicAdd(a.blp_done, 1ull);
        }
    }
    if (a.dbg && threadIdx.x == 0 && bx < 3900) { a.dbg[32 + (size_t)bx * 16 + 15] = wall_clock64(); a.dbg[32 + (size_t)bx * 16 + 14] = ((unsigned long long)pb << 8) | (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7); }
}
// a group of sectors per launch, argument blocks read from HBM (written once, at group creation).  Workgroups are dealt out
// sector-interleaved in dispatch order (blockIdx.x fastest): the first n * nMulti workgroups to reach the machine are the ones that
// carry ILPs, of ALL sectors -- with sector = blockIdx.y the last sector's ILPs queued behind every other sector's idle workgroups
// (one workgroup per CU at this LDS footprint).
__global__ __launch_bounds__(BLP_THREADS) void blp_batch_kernel(const PBatch av) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    BlpArgs a;
    const int n = gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int sector = lin % n, bx = lin / n;
    load_args(a, static_cast<const BlpArgs*>(av.p[sector]));
    blp_stamp_begin(a, bx);
    blp_body(a, lds, bx, gridDim.x);
    blp_stamp_end(a);
}

// the light pass of a group of sectors (no dynamic LDS: many workgroups per CU), followed by blp_batch_kernel on tier-2 argument blocks
__global__ __launch_bounds__(BLP_THREADS) void blp_light_batch_kernel(const PBatch av) {
    BlpArgs a;
    const int n = gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int sector = lin % n, bx = lin / n;
    load_args(a, static_cast<const BlpArgs*>(av.p[sector]));
    blp_stamp_begin(a, bx);
    if (!(a.status && a.status->overflow)) {
        blp_light(a, bx, gridDim.x, a.counts[1]);
        blp_singles<false>(a, bx, gridDim.x, a.counts[2]);
    }
    blp_stamp_end(a);
}
int launch_blp_light_batch(mht_ctx* ctx, const PBatch& av, int n_sectors, int grid_x) {
    hipLaunchKernelGGL(blp_light_batch_kernel, dim3(grid_x, n_sectors), dim3(BLP_THREADS), 0, ctx->stream, av);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

// Cluster-sharded trackers (several devices hold identical forests and solve disjoint sets of clusters): after the selections have
// been exchanged (sel_rel[t] = selected child relative to the target's block, identical on every device although the blocks may
// sit at different node indices), every device runs the per-target end of the scan for ALL targets: termination test, N-scan
// prune decision, new root, report record, surviving leaf range -- what blp_kernel does behind its own solves (tracker.py:891-916,
// pyTarget.py:343-356).  One wavefront per target.
// Behind the exchange of a cluster-sharded step: every device holds every device's file for the clusters searched by teams across devices
// (BlpArgs::shard_team).  Per slot the smallest value wins, ties go to the lowest device -- the same decision on every device -- and the winner's
// child ordinals become the members' sel_rel entries.
__global__ __launch_bounds__(TEAM_SEL) void shard_team_resolve_kernel(const BlpArgs a, const int shard_n) {
    if (a.status && a.status->overflow) return;
    const int ti = blockIdx.x;
    if (ti >= a.counts[5]) return;
    const ClRef cr = cl_ref(a, a.team_list[ti]);
    int best = -1;
    unsigned long long bk = ~0ull;
    for (int d = 0; d < shard_n; ++d) {      // (uniform)
        const int32_t* o = a.shard_team + ((size_t)d * TEAM_MAX + ti) * XT_WORDS;
        if (o[3] != cr.K) continue;      // (empty: -1)
        const unsigned long long key = ((unsigned long long)(unsigned)o[0] << 42) | ((unsigned long long)(unsigned)o[1] << 21) | (unsigned long long)(unsigned)o[2];
        if (best < 0 || key < bk) { best = d; bk = key; }
    }
    if (best < 0) return;      // (no device filed: the cluster was not searched by a team -- its owner's entries are in sel_rel already)
    const int32_t* o = a.shard_team + ((size_t)best * TEAM_MAX + ti) * XT_WORDS;
    for (int k = threadIdx.x; k < cr.K; k += blockDim.x) a.sel_rel[a.cl_members[cr.p0 + k]] = o[4 + k];
}

int launch_shard_team_resolve(mht_ctx* ctx, const BlpArgs& a, int shard_n) {
    hipLaunchKernelGGL(shard_team_resolve_kernel, dim3(TEAM_MAX), dim3(TEAM_SEL), 0, ctx->stream, a, shard_n);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

__global__ __launch_bounds__(BLP_THREADS) void blp_epilogue_kernel(const BlpArgs a, const int32_t* nT_dev) {
    if (a.status && a.status->overflow) return;
    const int nT = *nT_dev, lane = threadIdx.x & 63;
    for (int t = blockIdx.x * (BLP_THREADS / 64) + (threadIdx.x >> 6); t < nT; t += gridDim.x * (BLP_THREADS / 64)) {
        const TgtPre pre = load_target(a, t);
        const int s = pre.cb + a.sel_rel[t];
        if (lane == 0) a.sel[t] = s;
        const int va0 = sweep_prefetch(a, pre.j, pre.cb, pre.ce, lane);
        const int key = finish_target(a, t, s, pre, lane == 0);
        sweep_survivors(a, t, pre.j, pre.cb, pre.ce, key, va0, lane);
    }
    blp_stamp_end(a);
}

int launch_blp_epilogue(mht_ctx* ctx, const BlpArgs& a, const int32_t* nT_dev, int n_targets_ub) {
    int grid = (n_targets_ub + BLP_THREADS / 64 - 1) / (BLP_THREADS / 64);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(blp_epilogue_kernel, dim3(grid), dim3(BLP_THREADS), 0, ctx->stream, a, nT_dev);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

int launch_blp_batch(mht_ctx* ctx, const PBatch& av, int n_sectors, int grid_x, size_t lds) {
    static size_t attr = 0;
    if (lds > 48 * 1024 && lds > attr) {
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(blp_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(blp_batch_kernel, dim3(grid_x, n_sectors), dim3(BLP_THREADS), lds, ctx->stream, av);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

// the LDS tier of an argument block: tier 0 = the only launch (default footprint), 1 = small footprint first, 2 = default footprint
// for what tier 1 left; returns the dynamic LDS bytes of the launch
static size_t blp_set_tier_bytes(const BlpArgs& a) { return blp_lds_bytes(a.cap_h, a.cap_r, a.cap_k, a.cap_uw); }
size_t blp_set_tier(BlpArgs& a, int tier) {
    const int uw = (((a.n_mnodes + 63) / 64) + 1) & ~1;
    a.tier = tier;
    a.t1_h = 512; a.t1_k = 16;
    if (tier == 1) { a.cap_h = a.t1_h; a.cap_r = 256; a.cap_k = a.t1_k; }
    else { a.cap_h = BIG_MAXH; a.cap_r = BIG_MAXR; a.cap_k = BIG_MAXK; }
    if (tier == 0) {
        static int eh = -1, er = 0, ek = 0;
        if (eh < 0) { const char* e = getenv("MHT_BLP_CAPS"); eh = 0; if (e) sscanf(e, "%d,%d,%d", &eh, &er, &ek); }
        if (eh > 0) { a.cap_h = eh; a.cap_r = er; a.cap_k = ek; }
    }
    a.cap_uw = uw;
    return blp_lds_bytes(a.cap_h, a.cap_r, a.cap_k, a.cap_uw);
}

// can the ILP launch of a forest with this many target slots / measurement nodes derive its clusters itself (union-find prologue)?
bool blp_uf_fits(int Tcap, int n_mnodes) {
    BlpArgs b = {};
    b.n_mnodes = n_mnodes;
    const size_t lds = blp_set_tier(b, 0), pro = uf_prologue_bytes((size_t)Tcap);
    return Tcap <= 8192 && (((lds > pro ? lds : pro) + 15) & ~(size_t)15) + BLP_UF_PERSIST <= 158 * 1024;
}

int launch_blp(mht_ctx* ctx, const BlpArgs& a, int grid) {
    BlpArgs b = a;
    const size_t lds = blp_set_tier(b, 0);
    if (lds > 158 * 1024) {
        set_error("blp: %d measurement nodes do not fit the solver's LDS tables", a.n_mnodes);
        return MHT_E_CAPACITY;
    }
    if (ctx->lds_attr_blp < lds) {
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(blp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->lds_attr_blp = lds;
    }
    if (b.uf_epoch) {      // clusters from the grow launch's union-find
        size_t body = lds > uf_prologue_bytes((size_t)b.uf_cap) ? lds : uf_prologue_bytes((size_t)b.uf_cap);
        body = (body + 15) & ~(size_t)15;
        b.uf_lds_off = (unsigned)body;
        const size_t lds_uf = body + BLP_UF_PERSIST;
        if (lds_uf > 158 * 1024) {
            set_error("blp: the union-find prologue of %d targets does not fit the launch's LDS (%zu bytes)", b.uf_cap, lds_uf);
            return MHT_E_CAPACITY;
        }
        if (ctx->lds_attr_blp_uf < lds_uf) {
            MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(blp_uf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_uf));
            ctx->lds_attr_blp_uf = lds_uf;
        }
        hipLaunchKernelGGL(blp_uf_kernel, dim3(grid), dim3(BLP_THREADS), lds_uf, ctx->stream, b);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
    hipLaunchKernelGGL(blp_kernel, dim3(grid), dim3(BLP_THREADS), lds, ctx->stream, b);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

__global__ void blp_objective_kernel(const int32_t* sel, const double* cost, int nT, const int32_t* st, const int32_t* it,
                                     const int32_t* nd, double* out) {
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0;
    for (int t = 0; t < nT; ++t) s += cost[sel[t]];
    out[0] = s;
    out[1] = (double)st[0];
    out[2] = (double)it[0];
    out[3] = (double)nd[0];
}
}  // namespace mht

using namespace mht;

extern "C" int mht_solve_blp(mht_ctx* ctx, int32_t nHyp, int32_t nT, int32_t nRows, int32_t depth, const int32_t* group_ptr,
                             const int32_t* rows, const double* cost, int32_t max_iter, int32_t node_limit,
                             int32_t* selected, double* objective, int32_t* status, int32_t* iterations, int32_t* nodes) {
    MHT_REQUIRE(ctx && group_ptr && cost && selected, "mht_solve_blp: null argument");
    MHT_REQUIRE(nT >= 1 && nHyp >= nT && nRows >= 0 && depth >= 0 && depth <= MAXPD, "mht_solve_blp: bad sizes");
    MHT_REQUIRE(rows || depth == 0, "mht_solve_blp: rows is null");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    // one cluster holding all targets; a large one is searched by a team of workgroups (mht_kernels.h: TEAM_*), as in the forest: the launch
    // gets TEAM_W - 1 workgroups without a cluster of their own, and the HBM scratch is laid out in TEAM_W copies
    bool seam_team = nT >= TEAM_MIN_K && nT <= TEAM_SEL;
    { const char* e = getenv("MHT_BLP_NO_TEAMS"); if (e && e[0] == '1') seam_team = false; }
    const size_t W1 = seam_team ? (size_t)TEAM_W : 1;
    const size_t S1 = (size_t)2 * nT + 2, nR1 = (size_t)(nRows > 0 ? nRows : 1);
    const size_t S = S1 * W1;
    const size_t nR = nR1 * W1;
    // doubles: u[nR] best_rc bb_cost bb_uused bb_last_rc bb_rest bb_min [S each] out[4] snapshots[BB_SLOTS][BB_RE_LEVELS][snap_rows]
    const size_t snap_rows = nR1 > (size_t)BIG_MAXR ? nR1 : (size_t)BIG_MAXR;
    const size_t n_d = nR + 6 * S + 4 + (size_t)BB_SLOTS * BB_RE_LEVELS * snap_rows;
    // ints: usage[nR] mark[nR] best_h bb_ch bb_best bb_last_idx [S each] cl_ptr[2] members[nT] multi[1] single[1] counts[4] st it nd busy[BB_SLOTS]
    const size_t n_i = 2 * nR + 4 * S + 2 + nT + 2 + 8 + 3 + BB_SLOTS;
    int rc = ctx->hitmask.ensure(n_d * 8 + n_i * 4 + 64);
    if (rc) return rc;
    double* d = static_cast<double*>(ctx->hitmask.ptr);
    int32_t* q = reinterpret_cast<int32_t*>(d + n_d);
    MHT_HIP_CHECK(hipMemsetAsync(ctx->hitmask.ptr, 0, n_d * 8 + n_i * 4, ctx->stream));
    BlpArgs a = {};
    a.u = d; a.best_rc = d + nR; a.bb_cost = a.best_rc + S; a.bb_uused = a.bb_cost + S; a.bb_last_rc = a.bb_uused + S;
    a.bb_rest = a.bb_last_rc + S; a.bb_min = a.bb_rest + S;
    double* out = a.bb_min + S;
    a.bb_snap = out + 4; a.bb_snap_rows = (int)snap_rows;
    a.usage = q; a.mark = q + nR; a.best_h = a.mark + nR; a.bb_ch = a.best_h + S; a.bb_best = a.bb_ch + S; a.bb_last_idx = a.bb_best + S;
    int32_t* cl_ptr = a.bb_last_idx + S;
    int32_t* members = cl_ptr + 2;
    int32_t* multi = members + nT;
    int32_t* single = multi + 1;
    int32_t* counts = single + 1;
    int32_t* st = counts + 8;
    a.cl_ptr = cl_ptr; a.cl_members = members; a.multi_list = multi; a.single_list = single; a.counts = counts;
    a.cl_status = st; a.cl_iters = st + 1; a.cl_nodes = st + 2;
    a.bb_busy = st + 3;
    a.tchild = group_ptr; a.tcend = group_ptr + 1; a.cost = cost; a.cnllr = cost; a.path = rows; a.cap = nHyp; a.PD = depth; a.n_mnodes = (int)nR1;
    if (seam_team) { a.tm_sm = nR1; a.tm_ss = S1; }
    a.sel = selected;
    a.max_iter = max_iter < 0 ? 200 : max_iter;
    { const char* e = getenv("MHT_BLP_NO_ENUM"); a.no_enum = (e && e[0] == '1') ? 1 : 0; }
    { const char* e = getenv("MHT_BLP_NO_REDUCE"); a.no_reduce = (e && e[0] == '1') ? 1 : 0; }
    a.node_limit = node_limit <= 0 ? (1 << 20) : node_limit;
    { const char* e = getenv("MHT_BLP_TIME_LIMIT_US"); a.time_limit = e ? atoll(e) * 100 : 0; }      // testing: wall-clock budget per cluster (10 ns ticks)
    { const char* e = getenv("MHT_BLP_FORCE_HBM"); a.force_hbm = (e && e[0] == '1') ? 1 : 0; }
    if (seam_team) {
        const size_t tb = 64 + sizeof(TeamState) * TEAM_MAX + sizeof(TeamResult) * TEAM_W;
        rc = ctx->counts.ensure(tb);
        if (rc) return rc;
        char* tp = static_cast<char*>(ctx->counts.ptr);
        MHT_HIP_CHECK(hipMemsetAsync(tp, 0, tb, ctx->stream));
        a.team_list = reinterpret_cast<int32_t*>(tp);                                  // [0] = cluster 0 (memset)
        a.team_state = reinterpret_cast<TeamState*>(tp + 64);
        a.team_res = reinterpret_cast<TeamResult*>(tp + 64 + sizeof(TeamState) * TEAM_MAX);
        const unsigned long long inf_key = ~0ull;
        MHT_HIP_CHECK(hipMemcpyAsync(&a.team_state[0].gub, &inf_key, 8, hipMemcpyHostToDevice, ctx->stream));
    }
    int32_t* hbuf = new int32_t[nT + 12];
    hbuf[0] = 0; hbuf[1] = nT;
    for (int t = 0; t < nT; ++t) hbuf[2 + t] = t;
    hbuf[2 + nT] = 0; hbuf[3 + nT] = 0;
    hbuf[4 + nT] = 1; hbuf[5 + nT] = 1; hbuf[6 + nT] = 0; hbuf[7 + nT] = 0;
    hbuf[8 + nT] = 0; hbuf[9 + nT] = seam_team ? 1 : 0; hbuf[10 + nT] = 0; hbuf[11 + nT] = 0;      // counts[4..7]: [5] = clusters in the team list
    hipError_t e = hipMemcpyAsync(cl_ptr, hbuf, (size_t)(nT + 12) * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    delete[] hbuf;
    MHT_HIP_CHECK(e);
    rc = launch_blp(ctx, a, seam_team ? TEAM_W : 1);
    if (rc) return rc;
    hipLaunchKernelGGL(blp_objective_kernel, dim3(1), dim3(64), 0, ctx->stream, selected, cost, nT, st, st + 1, st + 2, out);
    MHT_HIP_CHECK(hipGetLastError());
    double res[4];
    MHT_HIP_CHECK(hipMemcpyAsync(res, out, 32, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (objective) *objective = res[0];
    if (status) *status = (int32_t)res[1];
    if (iterations) *iterations = (int32_t)res[2];
    if (nodes) *nodes = (int32_t)res[3];
    if ((int32_t)res[1] == MHT_BLP_NODE_LIMIT) {
        set_error("mht_solve_blp: node limit %d reached", a.node_limit);
        return MHT_E_LIMIT;
    }
    return MHT_OK;
}
