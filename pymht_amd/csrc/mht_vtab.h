// Covariances by VALUE: one table per forest, shared by all targets and all scans.
//
// The covariance of a hypothesis does not depend on which measurements it was updated with, only on whether there was one:
// P -> P_bar -> S, K -> P_hat never sees z (kalman.py:62, :90-93), and the reference itself shares one P_hat among all hit
// children of a node (pyTarget.py:246).  Targets that start from the same P_0 walk through the same matrices, and the recursion
// converges: on the headline stream 13 k leaves carry ~1 750 distinct covariances.  So a node does not own a covariance, it names
// one:
//   value id   = index of a distinct (covariance, P_d) pair in Pv / pdv, found or inserted through a hash table over the bits;
//   key        = 2 * (value id of the parent) + hit/miss: what a node stores (mht_nodes::cov).  child[key] = the node's own value
//                id, Gk[key] = the gains a leaf with that covariance needs (S^-1, K, ln(lambda_ex sqrt(det 2 pi S)/P_d), the
//                gate half-axes): 64 bytes, one dependent look-up behind the leaf's record, like the per-node column before.
// A transition (parent value, hit/miss) is computed ONCE, by whichever chain lane of fgrow_kernel meets it first; afterwards a
// leaf costs two 4-byte look-ups in the chain workgroups instead of ~1 400 dependent VALU operations and 256 bytes of stores.
// Roots get a key of their own (a pseudo parent id whose miss child is the root's value).
//
// Concurrency (several workgroups, on different XCDs whose L2s are not coherent, may meet the same new value in one launch):
// the hash slots are 64-bit words {tag, id + 1} driven by agent-scope atomics (they execute at the memory side); an inserter
// claims an empty slot (id field all ones), takes an id, writes the value with agent-scope atomic stores, waits for their
// acknowledgement and only then publishes the id in the slot.  A finder that meets a claimed slot re-reads it; one that meets
// a published slot with its tag compares all words of the value through agent-scope loads.  child[] / Gk[] / duplicates of a transition
// computed twice carry identical values: plain stores, consumed after the kernel boundary.  Value ids are handles: which id a
// value gets depends on the timing, nothing else does.
//
// float64 values (AIS forests: the covariances of AIS-updated targets, which the reference carries in float64 -- models/ais.py:4,
// tracker.py:451-487, :859-870; node flag F_COV_F64) live in the SAME table and take TWO consecutive ids: the 16 doubles fill the Pv
// records of id and id + 1, the hash slot's tag carries the kind in its top bit (a float32 value never compares equal to a float64
// one), keys are 2 * id + hit/miss as ever, and the gains of a float64 key k -- 16 doubles: S^-1, K, score constant, gate half-axes --
// fill the rows k and k + 2 (the twin id's row for the same hit/miss).  A pseudo parent of a float64 value takes two ids for the
// same reason.  One table, one id counter, one generation scheme; radar-only forests never see a float64 value.
#pragma once
#include "mht_common.h"
#include "mht_math.h"
#include "mht_la64.h"

namespace mht {

struct VTab {
    unsigned long long* Pv;        // [vcap][VT_PW] the NX * NX floats of a covariance as 64-bit words
    double* pdv;                   // [vcap] P_d of the value (part of its identity: the score constant depends on it)
    float4* Gk;                    // [2 * vcap][GKQ] gains by key
    int32_t* child;                // [2 * vcap] value id by key, -1 = transition not computed yet
    unsigned long long* slots;     // [hmask + 1] hash table
    unsigned* count;               // value ids handed out
    int32_t* overflow;             // sticky error word of the forest (FCounts::overflow)
    int vcap; unsigned hmask;
};

#if defined(__HIPCC__)
constexpr unsigned VT_PENDING = 0xffffffffu;

// value id of (words, pd): found or inserted.  Every lane may call this with its own value (no lane waits inside an iteration for
// another lane of its wavefront: a claim is published within the iteration that made it).  NW = VT_PW: a float32 covariance (one id);
// NW = 2 * VT_PW: a float64 one (two ids, see above).
constexpr unsigned VT_TAG64 = 0x80000000u;
template <int NW, typename VT> __device__ __forceinline__ int vt_find_or_insert_words(const VT& t, const unsigned long long* w, double pd) {
    constexpr bool F64 = NW == 2 * VT_PW;
    static_assert(NW == VT_PW || NW == 2 * VT_PW, "one or two records per value");
    unsigned long long h = 0x9e3779b97f4a7c15ull ^ (unsigned long long)__double_as_longlong(pd);
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        h ^= w[q];
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 32;
    const unsigned tag = F64 ? ((unsigned)(h >> 32) | VT_TAG64) : ((unsigned)(h >> 32) & ~VT_TAG64);
    unsigned pos = (unsigned)h & t.hmask;
    for (int guard = 0; guard < (1 << 24); ++guard) {
        const unsigned long long s = __hip_atomic_load(&t.slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s == 0ull) {
            const unsigned long long claim = ((unsigned long long)tag << 32) | VT_PENDING;
            unsigned long long expected = 0ull;
            if (__hip_atomic_compare_exchange_strong(&t.slots[pos], &expected, claim, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const unsigned id = atomicAdd(t.count, F64 ? 2u : 1u);
                if (id + (F64 ? 1u : 0u) >= (unsigned)t.vcap) {      // table full: the scan is void (MHT_E_CAPACITY), the claim is given back
                    *t.overflow = 1;
                    __hip_atomic_store(&t.slots[pos], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return 0;
                }
#pragma unroll
                for (int q = 0; q < NW; ++q) __hip_atomic_store(&t.Pv[(size_t)id * VT_PW + q], w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(&t.pdv[id]), (unsigned long long)__double_as_longlong(pd), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the value has arrived before its id becomes visible
                __hip_atomic_store(&t.slots[pos], ((unsigned long long)tag << 32) | (unsigned long long)(id + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return (int)id;
            }
            continue;      // somebody else took the slot: look at it again
        }
        if ((unsigned)(s >> 32) == tag) {
            const unsigned idp = (unsigned)s;
            if (idp == VT_PENDING) { __builtin_amdgcn_s_sleep(1); continue; }      // being written: look again
            const unsigned id = idp - 1u;
            bool same = (unsigned long long)__double_as_longlong(pd) ==
                        __hip_atomic_load(reinterpret_cast<unsigned long long*>(&t.pdv[id]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < NW; ++q) same = same && (w[q] == __hip_atomic_load(&t.Pv[(size_t)id * VT_PW + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (same) return (int)id;
        }
        pos = (pos + 1u) & t.hmask;      // another value lives here
    }
    *t.overflow = 1;
    return 0;
}
template <typename VT> __device__ __forceinline__ int vt_find_or_insert(const VT& t, const float* P, double pd) {
    unsigned long long w[VT_PW];
#pragma unroll
    for (int q = 0; q < VT_PW; ++q) w[q] = ((unsigned long long)__float_as_uint(P[2 * q + 1]) << 32) | __float_as_uint(P[2 * q]);
    return vt_find_or_insert_words<VT_PW>(t, w, pd);
}
// a float64 covariance: the first of its two ids
template <typename VT> __device__ __forceinline__ int vt_find_or_insert64(const VT& t, const double* P, double pd) {
    unsigned long long w[2 * VT_PW];
#pragma unroll
    for (int q = 0; q < 2 * VT_PW; ++q) w[q] = (unsigned long long)__double_as_longlong(P[q]);
    return vt_find_or_insert_words<2 * VT_PW>(t, w, pd);
}

// the covariance of value `id` (written in an earlier launch, or by this thread)
template <typename VT> __device__ __forceinline__ void vt_load(const VT& t, int id, float* P) {
    const uint4* p = reinterpret_cast<const uint4*>(t.Pv + (size_t)id * VT_PW);      // (VT_PW is even: 16-byte records)
#pragma unroll
    for (int q = 0; q < NP / 4; ++q) {
        const uint4 v = p[q];
        P[4 * q] = __uint_as_float(v.x); P[4 * q + 1] = __uint_as_float(v.y); P[4 * q + 2] = __uint_as_float(v.z); P[4 * q + 3] = __uint_as_float(v.w);
    }
}

template <typename VT> __device__ __forceinline__ void vt_load64(const VT& t, int id, double* P) {
    const uint4* p = reinterpret_cast<const uint4*>(t.Pv + (size_t)id * VT_PW);
#pragma unroll
    for (int q = 0; q < NP / 2; ++q) {
        const uint4 v = p[q];
        P[2 * q] = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
        P[2 * q + 1] = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
    }
}

// what a leaf with covariance P needs, one row of GKF floats: S^-1 (4), K (NX x 2), score constant, the two gate half-axes, padding
__device__ __forceinline__ void vt_gains(const Model& m, const float* P, double pd, float4* g) {
    CovChain c;
    cov_chain(m, P, c, false);
    float row[GKF];
#pragma unroll
    for (int e = 0; e < GKF; ++e) row[e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) row[e] = c.S_inv[e];
#pragma unroll
    for (int e = 0; e < NK; ++e) row[4 + e] = c.K[e];
    row[4 + NK] = nllr_const(c.S, m.lambda_ex, pd);
    row[5 + NK] = sqrtf((float)m.eta2 * fabsf(c.S[0]));
    row[6 + NK] = sqrtf((float)m.eta2 * fabsf(c.S[3]));
#pragma unroll
    for (int q = 0; q < GKQ; ++q) g[q] = make_float4(row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]);
}
// the fields of a gains row
constexpr int GK_LNC = 4 + NK, GK_RX = 5 + NK, GK_RY = 6 + NK;

// ... and with a float64 covariance: GKF DOUBLES, the same fields (the half-axes only feed the float32 pre-filter), in the rows `key`
// (first half) and `key + 2` (second half) of Gk
__device__ __forceinline__ void vt_gains64(const Model& m, const double* P, double pd, double* row) {
    CovChain64 c;
    cov_chain64(m, P, c, false);
#pragma unroll
    for (int e = 0; e < GKF; ++e) row[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) row[e] = c.S_inv[e];
#pragma unroll
    for (int e = 0; e < NK; ++e) row[4 + e] = c.K[e];
    row[GK_LNC] = nllr_const64(c.S, m.lambda_ex, pd);
    row[GK_RX] = (double)sqrtf((float)m.eta2 * fabsf((float)c.S[0]));
    row[GK_RY] = (double)sqrtf((float)m.eta2 * fabsf((float)c.S[3]));
}
template <typename VT> __device__ __forceinline__ void vt_store_gains64(const VT& t, int key, const double* row) {
    double2* g0 = reinterpret_cast<double2*>(t.Gk + (size_t)key * GKQ);
    double2* g1 = reinterpret_cast<double2*>(t.Gk + (size_t)(key + 2) * GKQ);
#pragma unroll
    for (int q = 0; q < GKQ; ++q) {
        g0[q] = make_double2(row[2 * q], row[2 * q + 1]);
        g1[q] = make_double2(row[GKF / 2 + 2 * q], row[GKF / 2 + 2 * q + 1]);
    }
}
template <typename VT> __device__ __forceinline__ void vt_load_gains64(const VT& t, int key, double* row) {
    const double2* g0 = reinterpret_cast<const double2*>(t.Gk + (size_t)key * GKQ);
    const double2* g1 = reinterpret_cast<const double2*>(t.Gk + (size_t)(key + 2) * GKQ);
#pragma unroll
    for (int q = 0; q < GKQ; ++q) {
        const double2 a = g0[q], b = g1[q];
        row[2 * q] = a.x; row[2 * q + 1] = a.y;
        row[GKF / 2 + 2 * q] = b.x; row[GKF / 2 + 2 * q + 1] = b.y;
    }
}
// a key of its own for a float64 value (a pseudo parent whose miss child it is, as for a root): two ids, so that the rows key and key + 2 are its
template <typename VT> __device__ __forceinline__ int vt_pseudo_key64(const VT& t, int id0, const double* gains_row) {
    const unsigned pid = atomicAdd(t.count, 2u);
    if (pid + 1u >= (unsigned)t.vcap) { *t.overflow = 1; return 0; }
    const int key = 2 * (int)pid;
    vt_store_gains64(t, key, gains_row);
    t.child[key] = id0;
    return key;
}
// the float64 value a float32 one becomes when NumPy promotes its batch (np.array of a list with a float64 member: exact conversion)
template <typename VT> __device__ __forceinline__ int vt_promote(const VT& t, int id32, double pd, double* P64) {
    float P[NP];
    vt_load(t, id32, P);
#pragma unroll
    for (int e = 0; e < NP; ++e) P64[e] = (double)P[e];
    return vt_find_or_insert64(t, P64, pd);
}
#endif

}  // namespace mht
