// Device code of the grow stage (target_part / target_wave / chain_part / fgrow_body): everything the kernels of mht_fgrow.hip are made of.
// See mht_fgrow.hip for the design notes.
#pragma once
#include "mht_kernels.h"
#include "mht_commit.h"
#include "mht_admit.h"
#include "mht_uf.h"
#include <stddef.h>
#include <hip/hip_ext.h>

namespace mht {


struct alignas(16) FLeaf {            // per-leaf results of phase 1 parked in LDS for the later phases (one chunk = FG_CAP leaves)
    double xbar[NX];
    double zhat[2];
    double cn, pd;
    float K[NK];
    float sinv[4];
    float lnc, bx, by, zhx, zhy;
    int src, cid;                     // cid: value id of the leaf's covariance (its children's keys are 2 * cid + hit/miss)
    unsigned char flags, f32state, valid, pad;
};
static_assert(sizeof(FLeaf) % 16 == 0, "FLeaf is copied with 16-byte LDS accesses");
// AIS forest only (fgrow_ais_kernel): the gains of a leaf whose target NumPy has promoted to float64 covariances (mht_vtab.h; models/ais.py:4,
// tracker.py:859-870) -- a second per-leaf record next to FLeaf, so that the kernels of radar-only forests stay byte for byte what they were
struct alignas(16) FLeafX {
    double K[NK];
    double sinv[4];
    double lnc;
    int f64, pad;                     // the gains above are valid (else: FLeaf's float32 ones)
};
static_assert(sizeof(FLeafX) % 16 == 0, "FLeafX is copied with 16-byte LDS accesses");

struct TInfo { int alive, first, cnt, depth, shift; };
// The first poll of a target's record and its depth, issued IN FRONT of the staging of the scan (overlapping launches): the compiler
// makes a load it knows to be uniform a round trip of its own (global_load, s_waitcnt vmcnt(0), v_readfirstlane) and put this one behind the
// staging loop's -- two dependent round trips at the head of the workgroups the launch ends with (seen in the ISA).  Through an index it
// cannot see through they are vector loads nobody waits for until the staging loop's own wait, which covers them.
struct TPre { unsigned long long w; int dep; };
constexpr int FG_MAP = 1024;                          // entries of the child -> leaf table of a chunk (more children: binary search)
constexpr int FG_HWC = 6;                             // constant-turn kernel: words of a leaf's hit mask over the target's candidate list (384 candidates -- with them the workgroup is 53.3 KB at 2 048 measurements, three per CU; more: CtGrow::hw_spill)
constexpr int FG_CONF = 256;                          // constant-turn kernel: entries of the union-find's conflict list in LDS (more are linked straight away)
constexpr int FG_CHAIN_TARGETS = FG_THREADS / 128 > 0 ? FG_THREADS / 128 : 1;     // targets per chain workgroup: wavefront = (target, hit/miss)
typedef const __attribute__((address_space(4))) FGrowArgs* KArgs;      // the kernel's own argument block (constant address space)

// what a workgroup needs to know about target slot t of the table this scan runs on; every index is clamped so that the
// loads go out unconditionally (one round trip), dead or out-of-range slots are masked afterwards
template <bool OVL = true, typename ARGS = void>
__device__ __forceinline__ TInfo target_info(const ARGS& a, const FDyn& d, int t, int nT, int* rf_out = nullptr, const int void_scan = 0, const bool pre = false, const TPre pv = TPre{0ull, 0}) {
    TInfo r;
    const int tc = (t < a.Tcap) ? t : 0;
    if (OVL && d.fused && d.ovl) {
        // the previous scan's ILP launch may still be running: the target's record (mht_kernels.h: TGT_REC_*) is published the moment the
        // target is finished there -- wait for it, for nothing else (slots up to the launch's grid bound all get one)
        // (pre: the first poll and the depth went out in front of the caller's staging loop, target_prefetch)
        const int dep = pre ? __builtin_amdgcn_readfirstlane(pv.dep) : a.p_depth[tc];
        const unsigned tag = (unsigned)d.c_scan & 0xffu;
        unsigned long long w = 0ull;
        bool ok = t < d.n_tgt;
        if (ok) {
            if (pre) w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pv.w >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pv.w);
            else w = __hip_atomic_load(&a.rec0[tc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> TGT_REC_TAG) != tag) {
                if (void_scan) ok = false;      // (a void scan publishes nothing: the caller leaves)
                else {
                    ok = spin_until(&a.rec0[tc], [&](unsigned long long x) { return (unsigned)(x >> TGT_REC_TAG) == tag; }, w);
                    if (!ok && (threadIdx.x & 63) == 0) { a.status->overflow = 2; atomicOr(&a.status->pad[0], 1 << 1); }      // (the wait timed out: the scan is void)
                }
            }
        }
        const int j = (int)((w >> TGT_REC_J) & 15ull);
        r.alive = ok && ((w >> TGT_REC_ALIVE) & 1ull) != 0ull;
        r.first = (int)(w & 0x7ffffffull); r.cnt = (int)((w >> TGT_REC_CNT) & 0x7fffffull); r.depth = dep + 1 - j; r.shift = j;
        if (rf_out) *rf_out = (int)((w >> TGT_REC_RF) & 1ull);
    } else if (d.fused) {      // the previous scan's commit has not run: its per-target results, indexed by old slot
        const int st = a.p_status[tc], cnt = a.p_count[tc], j = a.p_jdrop[tc], first = a.p_firstsurv[tc], dep = a.p_depth[tc];
        r.alive = (t < nT) && st == 0;
        r.first = first; r.cnt = cnt; r.depth = dep + 1 - j; r.shift = j;
    } else {
        const int first = a.t_first[tc], o0 = a.t_leaf_off[tc], o1 = a.t_leaf_off[tc + 1], dep = a.t_depth[tc], sh = a.t_shift[tc];
        r.alive = t < nT;
        r.first = first; r.cnt = o1 - o0; r.depth = dep; r.shift = sh;
    }
    if (!r.alive) r.cnt = 0;
    return r;
}

__device__ __forceinline__ int fg_sortable(float f) {      // monotone map float -> int
    const int i = __float_as_int(f);
    return i >= 0 ? i : (i ^ 0x7fffffff);
}

// per-phase wall-clock stamps (tools/grow_profile.py): compiled in only with -DMHT_GROW_STAMPS
#ifdef MHT_GROW_STAMPS
#define FG_STAMP(k) do { if (d.dbg && (threadIdx.x & 63) == 0 && threadIdx.x < 128 && blockIdx.x < 3900) d.dbg[32 + (size_t)blockIdx.x * 16 + (threadIdx.x >> 6) * 8 + (k)] = wall_clock64(); } while (0)
#define FG_STAMPX(k) do { if (d.dbg && threadIdx.x == 0 && blockIdx.x < 3900) d.dbg[32 + (size_t)blockIdx.x * 16 + 8 + (k)] = wall_clock64(); } while (0)
#else
#define FG_STAMP(k)
#define FG_STAMPX(k)
#endif

// ---- chain workgroups: the gains one scan ahead --------------------------------------------------------------------------
// A node names its covariance by a KEY into the forest's value table (mht_vtab.h): key = 2 * (value id of the parent) + hit/miss,
// child[key] = the node's own value id, Gk[key] = the gains a leaf with that covariance needs.  For every leaf of the previous
// layer and both hit/miss, the children's key 2 * child[leaf key] + h must be resolved before THEY are leaves (next scan):
// wavefront = (target, hit/miss); the lanes first agree on the distinct keys among the target's leaves (ballots), then lane j
// looks the j-th one's transition up.  Nearly always it is known (a 4-byte look-up: ~1 750 distinct covariances serve 13 k
// leaves, and what the recursion has reached once it reaches again); otherwise the lane runs the chain P -> P_bar, S, K, P_hat
// (kalman.py:62, :90-93), finds or inserts the child's covariance by value and writes its gains -- S^-1, K,
// ln(lambda_ex sqrt(det 2 pi S)/P_d), gate half-axes, all from predict(child covariance).
// One transition of the value table: the child (hit: h = 1, miss: h = 0) of covariance value `id` -- its covariance by value and ITS
// gains, filed under key 2 * id + h (see chain_part)
// Covariances are shared by VALUE across all targets (and sectors' targets of one forest): a transition that is new this scan is
// usually met by many wavefronts at once.  The first one claims it (child[key]: -1 -> -2 with one agent-scope compare-and-swap) and
// computes it; the others have nothing to wait for -- nobody reads the entry before the next scan -- and move on.
constexpr int VT_CLAIMED = -2;
template <typename ARGS>
__device__ __forceinline__ void chain_resolve(const ARGS& a, int id, int h, double pd) {
    const int ckey = 2 * id + h;
    if (atomicCAS(&a.vt.child[ckey], -1, VT_CLAIMED) != -1) return;      // known, or somebody else is at it
    float P[NP];
    vt_load(a.vt, id, P);
    Model mdl;          // (uniform registers)
#pragma unroll
    for (int e = 0; e < NP; ++e) { mdl.A[e] = a.model.A[e]; mdl.Q[e] = a.model.Q[e]; }
#pragma unroll
    for (int e = 0; e < NK; ++e) mdl.C[e] = a.model.C[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) mdl.R[e] = a.model.R[e];
    mdl.eta2 = a.model.eta2; mdl.lambda_ex = a.model.lambda_ex;
    float Pc[NP];
    {
        CovChain c;
        cov_chain(mdl, P, c, h != 0);      // (the miss child's covariance is P_bar: no S, K, P_hat needed)
        if (h) {
#pragma unroll
            for (int e = 0; e < NP; ++e) Pc[e] = c.P_hat[e];
        } else {
#pragma unroll
            for (int e = 0; e < NP; ++e) Pc[e] = c.P_bar[e];
        }
    }
    {
        float4 rec[GKQ];
        vt_gains(mdl, Pc, pd, rec);
#pragma unroll
        for (int q = 0; q < GKQ; ++q) a.vt.Gk[(size_t)ckey * GKQ + q] = rec[q];
    }
    a.vt.child[ckey] = vt_find_or_insert(a.vt, Pc, pd);
}

// the same for a float64 value (AIS forests, mht_vtab.h): dgemm chains and dgesv in OpenBLAS' order (mht_la64.h)
template <typename ARGS>
__device__ __forceinline__ void chain_resolve64(const ARGS& a, int id, int h, double pd) {
    const int ckey = 2 * id + h;
    if (atomicCAS(&a.vt.child[ckey], -1, VT_CLAIMED) != -1) return;
    double P[NP];
    vt_load64(a.vt, id, P);
    Model mdl;
#pragma unroll
    for (int e = 0; e < NP; ++e) { mdl.A[e] = a.model.A[e]; mdl.Q[e] = a.model.Q[e]; }
#pragma unroll
    for (int e = 0; e < NK; ++e) mdl.C[e] = a.model.C[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) mdl.R[e] = a.model.R[e];
    mdl.eta2 = a.model.eta2; mdl.lambda_ex = a.model.lambda_ex;
    double Pc[NP];
    {
        CovChain64 c;
        cov_chain64(mdl, P, c, h != 0);
#pragma unroll
        for (int e = 0; e < NP; ++e) Pc[e] = h ? c.P_hat[e] : c.P_bar[e];
    }
    {
        double row[GKF];
        vt_gains64(mdl, Pc, pd, row);
        vt_store_gains64(a.vt, ckey, row);
    }
    a.vt.child[ckey] = vt_find_or_insert64(a.vt, Pc, pd);
}

template <bool OVL = true, int AIS = 0, typename ARGS = void>
__device__ __forceinline__ void chain_part(const ARGS& a, const FDyn& d, int cb, const int t_off = 0, const int born = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nT = born ? a.nT_new[0] : a.nT_dev[0];
    const int po = a.prev_status->overflow, so = *a.sticky_overflow;
    // wavefront = (target, hit/miss), lane = leaf: two targets per workgroup
    const int t = t_off + cb * FG_CHAIN_TARGETS + (wave >> 1), h = wave & 1;
    const TInfo ti = target_info<OVL>(a, d, t, nT, nullptr, po | so);
    if (po || so) return;
    FG_STAMP(0);
    // AIS forest: a target with a float64-covariance leaf is promoted as a whole (np.array of its leaves' P_0, tracker.py:859-862): its
    // float32 leaves continue from their covariance converted to float64 -- the same value the target's workgroup finds or inserts
    bool prom = false;
    if (AIS)
        for (int l0 = 0; l0 < ti.cnt; l0 += 64) {
            const uint8_t f = (l0 + lane < ti.cnt) ? a.flags[ti.first + l0 + lane] : (uint8_t)0;
            prom = prom || __any((f & F_COV_F64) && !(f & F_DEAD));
        }
    for (int l0 = 0; l0 < ti.cnt; l0 += 64) {
        const bool v = l0 + lane < ti.cnt;
        const int src = ti.first + (v ? l0 + lane : 0);
        const int covc = a.cov[src];
        // distinct keys among these leaves: lane j takes the j-th (another chunk of the same target may repeat one: it is then
        // resolved twice, with identical results)
        unsigned long long rem = __ballot(v);
        int mykey = -1, mysrc = 0;
        for (int j = 0; rem; ++j) {
            const int leader = __ffsll((long long)rem) - 1;
            const int cv = __shfl(covc, leader), sv = __shfl(src, leader);
            rem &= ~__ballot(covc == cv);
            if (lane == j) { mykey = cv; mysrc = sv; }
        }
        if (mykey < 0) continue;
        int id = a.vt.child[mykey];                // (set when the leaf was made: by this code one scan ago, or at its birth)
        if (AIS && prom) {
            const double pd = a.pd[mysrc];
            if (!(a.flags[mysrc] & F_COV_F64)) { double P64[NP]; id = vt_promote(a.vt, id, pd, P64); }
            if (a.vt.child[2 * id + h] >= 0) continue;
            chain_resolve64(a, id, h, pd);
            continue;
        }
        const int ckey = 2 * id + h;
        if (a.vt.child[ckey] >= 0) continue;       // the transition is known
        const double pd = a.pd[mysrc];
        chain_resolve(a, id, h, pd);
    }
    FG_STAMP(1);
}
// (Tried and dropped: resolving the transitions inside the target workgroups, behind their emission -- a new transition costs
// ~5 us, some workgroup meets one on almost every scan, and there it lands at the end of the kernel: grow stage 22 -> 26 us; and
// eight targets per chain workgroup -- a wavefront walks its targets one after the other, four dependent round trips each, and
// the chain workgroups became the kernel's tail, 26 us again.  One (target, hit/miss) per wavefront starts with the launch and is
// done at ~6 us, long before the target workgroups.)

// ---- target workgroups ---------------------------------------------------------------------------------------------------
// One child, one ROLE: the four wavefronts of the workgroup all walk the children (lane = child) and each does a quarter of
// the work -- role 0: x[0..1], cumulativeNLLR; 1: x[2..3], P_d, parent; 2: measurement number, covariance column, flags, ILP
// cost, used-measurement byte; 3: path and ancestor records.  (One wavefront doing everything was a ~1500-instruction serial
// stream, 2.9 us; what every role needs -- which hit, z_tilde, NIS, the score -- is recomputed by each.)
// AIS forest: the two records of child c.  A path record has two halves of `half` levels: radar measurement nodes, AIS message nodes;
// the parent's entries from the new root on (level + shift) in both, the child's own rows at level `depth`.
template <int PQ, typename ARGS>
__device__ __forceinline__ void fg_emit_records_ais(const ARGS& a, int l, int c, int depth, int shift, int radar_row, int ais_row, const int* s_pp, const int* s_ap) {
    const int half = a.ais.half;
    const int* pl = s_pp + l * (PQ * 4);
    const int* al = s_ap + l * (PQ * 4);
    int4* po = reinterpret_cast<int4*>(a.out_path + (size_t)c * (PQ * 4));
    int4* ao = reinterpret_cast<int4*>(a.out_apath + (size_t)c * (PQ * 4));
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        int pe[4], ae[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = q * 4 + e;
            const int hb = (i >= half) ? half : 0, lvl = i - hb;
            const bool in_rec = i < 2 * half;
            const int src_p = (in_rec && lvl < depth) ? hb + lvl + shift : 0;
            const int src_a = (i < depth) ? i + shift : 0;
            const int pvv = pl[src_p], avv = al[src_a];
            pe[e] = !in_rec ? -1 : (lvl < depth) ? pvv : ((lvl == depth) ? (hb ? ais_row : radar_row) : -1);
            ae[e] = (i < depth) ? avv : ((i == depth) ? c : -1);
        }
        po[q] = make_int4(pe[0], pe[1], pe[2], pe[3]);
        ao[q] = make_int4(ae[0], ae[1], ae[2], ae[3]);
    }
}

// AIS forest: fused child f of a leaf -- everything comes out of the record forest_ais_kernel left (mht_ais.hip); float64 state and
// score whatever the leaf's chains are (tracker.py:484-500: ais.C and the message are float64)
template <int PQ, typename ARGS>
__device__ __forceinline__ void fg_emit_fused(const ARGS& a, const FDyn& d, const FLeaf& g, int l, int c, const AisRec& r, const int* s_pp, const int* s_ap,
                                              int depth, int shift, double rootc) {
    const size_t cap = a.cap;
#pragma unroll
    for (int i = 0; i < 4; ++i) a.ox[(size_t)i * cap + c] = r.x[i];
    const double cnl = g.cn + r.nllr;
    a.ocnllr[c] = cnl;
    a.opd[c] = g.pd;
    a.oparent[c] = g.src;
    a.omeas[c] = r.radar + 1;          // 0 with an identity = a child without a radar measurement (measurementNumber None)
    a.ocov[c] = r.key;
    a.oflags[c] = F_COV_F64;      // (the key names a float64 value: tracker.py:451-487 carries the fused covariance in float64)
    a.ocost[c] = (cnl - rootc) / (double)a.Nwin;
    a.ais.ommsi[c] = r.mmsi;
    a.ais.ohmmsi[c] = r.mmsi;
    // (no used-measurement byte: tracker.py:333-334 marks the measurements of the pure radar gate only)
    fg_emit_records_ais<PQ>(a, l, c, depth, shift, r.radar >= 0 ? a.cur_slot_base + r.radar : -1, a.cur_slot_base + d.M + r.msg, s_pp, s_ap);
}

template <typename TS, int PQ, int AIS = 0, typename ARGS = void>
__device__ __forceinline__ void fg_emit_child(const ARGS& a, const FDyn& d, int role, const FLeaf& g, int l, int c, int k, int nh, const unsigned long long* hwl,
                                              const float* zx, const float* zy, const int* s_pp, const int* s_ap, int depth, int shift,
                                              double rootc, int root_f32, const unsigned short* cand = nullptr, const float2* zg = nullptr,
                                              const FLeafX* gx = nullptr) {
    // (gx != null && gx->f64 -- AIS forest, a promoted target: float64 gains, TS = double, the children's covariances are float64)
    // (cand != null -- the wavefront-per-target kernel: the hit words index the target's candidate list, and the scan is read from
    // global memory, zg, not from an LDS copy)
    const size_t cap = a.cap;
    const uint8_t fl = g.flags;
    int meas = 0, hit = 0, j = -1;
    if (k > 0) {             // (k-1)-th gated measurement in ascending index (pyTarget.py:242-254)
        int need = k - 1, w = 0;
        unsigned long long bits = hwl[0];
        while (true) {
            const int pc = __popcll(bits);
            if (need < pc) break;
            need -= pc;
            bits = hwl[++w];
        }
        for (int q = 0; q < need; ++q) bits &= bits - 1;
        j = w * 64 + __ffsll((long long)bits) - 1;
        if (cand) j = cand[j];
        meas = j + 1;
        hit = 1;
    }
    FG_STAMPX(2);
    if (AIS) {
        fg_emit_records_ais<PQ>(a, l, c, depth, shift, meas > 0 ? a.cur_slot_base + meas - 1 : -1, -1, s_pp, s_ap);
    } else if (role == 3 || role < 0) {
        // path / ancestor records of the child: the parent's entries from the new root on (d + shift), its own at level `depth`
        const int* pl = s_pp + l * (PQ * 4);
        const int* al = s_ap + l * (PQ * 4);
        int4* po = reinterpret_cast<int4*>(a.out_path + (size_t)c * (PQ * 4));
        int4* ao = reinterpret_cast<int4*>(a.out_apath + (size_t)c * (PQ * 4));
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < PQ) {
                int pe[4], ae[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = q * 4 + e;
                    const int src_i = (d < depth) ? d + shift : 0;
                    const int pvv = pl[src_i], avv = al[src_i];
                    pe[e] = (d < depth) ? pvv : ((d == depth && meas > 0) ? a.cur_slot_base + meas - 1 : -1);
                    ae[e] = (d < depth) ? avv : ((d == depth) ? c : -1);
                }
                po[q] = make_int4(pe[0], pe[1], pe[2], pe[3]);
                ao[q] = make_int4(ae[0], ae[1], ae[2], ae[3]);
            }
        if (role == 3) return;
    }
    FG_STAMPX(3);
    double cnl;
    const bool g64 = AIS && gx && gx->f64;
    uint8_t cfl = g64 ? (uint8_t)F_COV_F64 : (uint8_t)(fl & F_STATE_F32);
    TS zt[2] = {(TS)0, (TS)0};
    if (k == 0) {            // missed-detection child (pyTarget.py:319-328)
        const double inc = (g.pd == a.default_pd) ? a.default_miss_nllr : -log(1.0 - g.pd);
        cnl = g.cn + inc;
    } else {
        float mx, my;
        if (zg) { const float2 v = zg[j]; mx = v.x; my = v.y; } else { mx = zx[j]; my = zy[j]; }
        TS zh[2] = {(TS)g.zhat[0], (TS)g.zhat[1]}, nis;
        if (g64) gate_pair<TS>(zh, gx->sinv, mx, my, (TS)a.model.eta2, zt, nis);
        else gate_pair<TS>(zh, g.sinv, mx, my, (TS)a.model.eta2, zt, nis);
        const TS tinc = (TS)0.5 * nis + (g64 ? (TS)gx->lnc : (TS)g.lnc);           // kalman.py:19
        if (sizeof(TS) == 4 && (fl & F_SCORE_F32)) {          // float32 + float32 stays float32 (NumPy scalar rules)
            cnl = (double)((float)g.cn + (float)tinc);
            cfl |= F_SCORE_F32;
        } else {
            cnl = g.cn + (double)tinc;
        }
    }
    FG_STAMPX(4);
    if (role < 0) {          // (all state components)
        double xo[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xo[i] = g.xbar[i];
        if (k > 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i)      // (one hit: gemv, mht_math.h)
                xo[i] = g64 ? (double)update_component_n<TS>((TS)g.xbar[i], gx->K[i * 2], gx->K[i * 2 + 1], zt, nh == 1)
                            : (double)update_component_n<TS>((TS)g.xbar[i], g.K[i * 2], g.K[i * 2 + 1], zt, nh == 1);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) a.ox[(size_t)i * cap + c] = xo[i];
        a.ocnllr[c] = cnl;
        a.opd[c] = g.pd;
        a.oparent[c] = g.src;
    } else if (role < 2) {          // two state components each
        const int i0 = 2 * role, i1 = 2 * role + 1;
        double x0 = g.xbar[i0], x1 = g.xbar[i1];
        if (k > 0) {
            x0 = (double)update_component_n<TS>((TS)g.xbar[i0], g.K[i0 * 2], g.K[i0 * 2 + 1], zt, nh == 1);
            x1 = (double)update_component_n<TS>((TS)g.xbar[i1], g.K[i1 * 2], g.K[i1 * 2 + 1], zt, nh == 1);
        }
        a.ox[(size_t)i0 * cap + c] = x0;
        a.ox[(size_t)i1 * cap + c] = x1;
        if (role == 0) {
            a.ocnllr[c] = cnl;
        } else {
            a.opd[c] = g.pd;
            a.oparent[c] = g.src;
        }
        return;
    }
    FG_STAMPX(5);
    a.omeas[c] = meas;
    a.ocov[c] = 2 * g.cid + hit;
    a.oflags[c] = cfl;
    if (k > 0) a.used_bytes[j] = 1;
    // getScore()/N (pyTarget.py:124, tracker.py:1127) with NumPy's scalar promotion: float32 - float32 and float32 / int stay
    // float32
    if ((cfl & F_SCORE_F32) && root_f32) a.ocost[c] = (double)(((float)cnl - (float)rootc) / (float)a.Nwin);
    else a.ocost[c] = (cnl - rootc) / (double)a.Nwin;
    FG_STAMPX(6);
}

// the prediction of a target's ONLY leaf, in BLAS gemv order (see target_part)
template <typename ARGS>
__device__ __forceinline__ void fg_single_leaf(const ARGS& a, int src, bool f32state, FLeaf& out) {
    Model mdl;
#pragma unroll
    for (int e = 0; e < NP; ++e) mdl.A[e] = a.model.A[e];
#pragma unroll
    for (int e = 0; e < NK; ++e) mdl.C[e] = a.model.C[e];
    double xd[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) xd[k] = a.x[(size_t)k * a.cap + src];
    double xb[NX], zh[2];
    if (f32state) {
        float xs[NX], xbf[NX], zhf[2];
#pragma unroll
        for (int k = 0; k < NX; ++k) xs[k] = (float)xd[k];
        state_predict_single<float>(mdl, xs, xbf, zhf);
#pragma unroll
        for (int k = 0; k < NX; ++k) xb[k] = (double)xbf[k];
        zh[0] = (double)zhf[0]; zh[1] = (double)zhf[1];
    } else {
        state_predict_single<double>(mdl, xd, xb, zh);
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) out.xbar[k] = xb[k];
    out.zhat[0] = zh[0]; out.zhat[1] = zh[1];
    out.zhx = (float)zh[0]; out.zhy = (float)zh[1];
}

// PQ = 16-byte pieces of a path / ancestor record (2: records of 8 ints, N-scan <= 7; 4: 16 ints) -- a template parameter because
// a leaf's two records sit in registers between their load and their LDS store: 32 registers at PQ = 4, and the kernel is at the
// edge of its budget (128 for four workgroups per CU in the batched launch).
// bslot: the slot whose static block of the node index space the target's children take (its own; a target admitted inside this launch
// takes one behind the slots of the uncommitted table, which the other workgroups of the launch are using).  born = 1: such a target --
// slot t of the COMMITTED table (d.fused = 0 for it), count and root columns of that table
// LEAN: the batched launches (groups of sectors) -- no union-find, no overlap with the previous scan's ILP launch: compiled out (the
// batched kernel sits at 128 registers for four workgroups per CU)
// ---- the end of a target workgroup of an OVERLAPPING launch, as a piece of its own ----------------------------------------------------
// What a target still owes once its children are out is indexed by its COMPACTED index, which only the commit of the previous scan knows
// (FCounts::ni_flag): tchild / tcend and -- when a target died in that scan -- its place in the union-find under the alternative epoch.
// Every target has a workgroup of its own, which simply waits for the word.
// (the whole workgroup calls; `tb` = the target's association bitset in LDS).  Returns false when the wait timed out.
template <typename ARGS>
__device__ __forceinline__ bool target_tail(const ARGS& a, const FDyn& d, int t, int base, int fin_tot, const unsigned long long* tb, int* conf, int conf_cap, int* nconf) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tc = (t < a.Tcap) ? t : 0;
    unsigned long long v;
    const bool ok = spin_until(a.ni_flag, [&](unsigned long long x) { return (unsigned)x == (unsigned)d.c_scan; }, v);
    const bool moved = ok && ((v >> 32) & 1ull);      // some target died: slots and compacted indices differ
    const int pos = !ok ? -1 : (moved ? __hip_atomic_load(&a.new_index[tc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t);
    if (pos < 0) { if (tid == 0) { a.status->overflow = 2; atomicOr(&a.status->pad[0], 1 << 3); } return false; }      // (timed out; a live target always has an index)
    if (tid == 0) { a.tchild[pos] = base; a.tcend[pos] = base + fin_tot; }
    if (moved && d.uf_epoch && wave == FG_THREADS / 64 - 1) {
        const unsigned alt = d.uf_epoch | 1u;      // (epochs are 2 x scan: the alternative one lies between this scan's and the next scan's, the words are updated by atomic max)
        uf_claim(a.uf_owner, a.uf_parent, alt, pos, tb, a.AW, lane, conf, conf_cap, nconf);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < *nconf; i += 64) uf_link(a.uf_parent, alt, pos, conf[i]);
    }
    return true;
}
template <int PQ, int CAP, int AIS = 0, bool LEAN = false, int CT = 0>
__device__ __forceinline__ void target_part(KArgs ap0, const FDyn& d0, int t, unsigned char* smem, const int bslot, const int born = 0) {
    FDyn d = d0;
    if (LEAN) { d.uf_epoch = 0u; d.ovl = 0; d.stamp_end = 0; }
    constexpr int PDS = PQ * 4;
    const auto& a = *ap0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = d.M, W = d.W, Mpad = W * 64, AW = a.AW;
    // LDS carve (every block a multiple of 16 bytes)
    float* zx = reinterpret_cast<float*>(smem);
    float* zy = zx + Mpad;
    FLeaf* lg = reinterpret_cast<FLeaf*>(zy + Mpad);                                        // [CAP]
    int* s_pp = reinterpret_cast<int*>(lg + CAP);                                        // [CAP][pds] path records of the leaves
    int* s_ap = s_pp + PDS * CAP;                                                        // [CAP][pds] ancestor records
    // Hit masks.  [CAP][W] words over the scan's measurements -- except in the constant-turn kernel (long scans: W = 32 at 2 048 measurements
    // is 24.6 KB of a 71 KB workgroup, two per CU): there a leaf's hits are bits over the target's CANDIDATE list (the measurements inside its
    // box, in ascending order: a few dozen), FG_HWC words per leaf; a target with more candidates than that keeps full-width masks in a block
    // of global memory of its slot (CtGrow::hw_spill).
    constexpr bool CMP = CT != 0;
    const int HWS = CMP ? FG_HWC : W;                                                    // words per leaf in LDS
    unsigned long long* hw = reinterpret_cast<unsigned long long*>(s_ap + PDS * CAP);    // [CAP][HWS] hit masks
    unsigned long long* tb = hw + (size_t)CAP * HWS;                                     // [AW] association bitset of the target
    int* s_pref = reinterpret_cast<int*>(tb + AW);                                          // [CAP + 1]
    int* s_misc = s_pref + CAP + 4;                                                      // [32]
    unsigned short* cand = reinterpret_cast<unsigned short*>(s_misc + 32);                  // [Mpad]
    unsigned char* s_map = reinterpret_cast<unsigned char*>(cand + Mpad);                   // [FG_MAP] leaf of the chunk's r-th child
    int* s_conf = reinterpret_cast<int*>(s_map + FG_MAP);                                   // constant-turn kernel: [FG_CONF] the union-find's conflict list (cand lives until the emission there)
    int* s_ais = reinterpret_cast<int*>(s_map + FG_MAP);                                    // AIS forest: [CAP][4] fused children (count, first record), bound identity (never together with s_conf)
    FLeafX* lgx = reinterpret_cast<FLeafX*>(s_ais + 4 * CAP);                              // AIS forest: [CAP] float64 gains of a promoted target's leaves
    int& s_ncand = s_misc[0];
    int& s_base = s_misc[1];
    int& s_ebase = s_misc[2];
    int& s_total = s_misc[3];
    int* s_red = s_misc + 4;          // [4] per-wave partials of the alive prefix
    int* s_boxp = s_misc + 8;         // [2][4] per-wave gate boxes as sortable ints: min x, max x, min y, max y
    int& s_live = s_misc[16];         // live leaves of a target with more than 64 leaf slots (only counted behind similar-state pruning)

    FG_STAMP(0);
    // ---- first round trip: everything that is addressed by the target slot alone -----------------------------------------
    const int nT = born ? a.nT_new[0] : a.nT_dev[0];
    const int po = a.prev_status->overflow, so = *a.sticky_overflow;
    const bool ovl = !LEAN && d.fused && d.ovl;      // (the previous scan's ILP launch may still be running)
    int rf_rec = 0;
    const float2* z2 = reinterpret_cast<const float2*>(d.z);
    if (d.z_tag) {        // (streamed path: the scan's staging kernel ran on another stream and nobody waited for it -- usually long done)
        unsigned long long v;
        if (!spin_until(d.z_flag, [&](unsigned long long x) { return x >= d.z_tag; }, v) && tid == 0) { a.status->overflow = 2; atomicOr(&a.status->pad[0], 1 << 2); }      // (tags only grow: a later scan may have been staged already)
    }
    TPre tpre = {0ull, 0};
    if (ovl) {
        int zo;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zo));
        const int tcp = ((t < a.Tcap) ? t : 0) + zo;
        tpre.w = __hip_atomic_load(&a.rec0[tcp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tpre.dep = a.p_depth[tcp];
    }
    if (ovl)              // (the wait for the target's record comes behind everything that does not depend on it)
        for (int j = tid; j < Mpad; j += FG_THREADS) {
            const float2 v = (j < M) ? z2[j] : make_float2(3.0e38f, 3.0e38f);
            zx[j] = v.x;
            zy[j] = v.y;
        }
    const TInfo ti = target_info<!LEAN>(a, d, t, nT, &rf_rec, po | so, ovl, tpre);
    const int tc = (t < a.Tcap) ? t : 0;
    // (overlapping launch: the root's score was written through in front of the target's record)
    const double rootc = born ? a.b_root_cnllr[tc]
                              : (ovl ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(&a.t_root_cnllr[tc]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                     : a.t_root_cnllr[tc]);
    const int root_f32 = born ? a.b_root_f32[tc] : (ovl ? rf_rec : a.t_root_f32[tc]);
    int acc = 0;
    if (d.fused && !ovl)  // compacted index of this target = alive slots before it (the commit computes the same in workgroup 0)
        for (int i = tid; i < t; i += FG_THREADS) acc += (a.p_status[i] == 0) ? 1 : 0;
    if (!ovl)
        for (int j = tid; j < Mpad; j += FG_THREADS) {
            const float2 v = (j < M) ? z2[j] : make_float2(3.0e38f, 3.0e38f);
            zx[j] = v.x;
            zy[j] = v.y;
        }
    if (po || so) {          // a scan that overflowed its pools voids every scan after it
        if (t == 0 && tid == 0) a.status->overflow = po ? po : 1;
        return;
    }
    if (!ti.alive) return;
    FG_STAMP(1);
    // AIS forest: one LIVE leaf with a float64 covariance -- an AIS-updated node or a descendant of one -- and NumPy promotes the target's
    // whole batch: np.array([node.P_0 ...]) and np.array([node.x_0 ...]) of a list with a float64 member are float64 (tracker.py:859-862), so
    // every leaf's chain runs in float64 from its own (exactly converted) values and every child carries float64 state and covariance
    bool prom = false;
    if (AIS) {
        int anyf = 0;
        for (int i = tid; i < ti.cnt; i += FG_THREADS) { const uint8_t f = a.flags[ti.first + i]; anyf |= ((f & F_COV_F64) && !(f & F_DEAD)) ? 1 : 0; }
        prom = __syncthreads_or(anyf) != 0;
    }
    // AIS forest: tree levels (below the root this scan runs on) the target's association set has been rebuilt from
    int rebuilt_levels = 0;
    if (AIS) {
        const int rl = born ? 0 : (a.ais.t_window[tc] >> WIN_REBUILT_SHIFT) & 0xff;
        rebuilt_levels = (ti.shift > 0 || rl == WIN_REBUILT_ALL) ? (1 << 20) : rl;      // (shift > 0: the root advanced in the scan before -- that commit may still be pending)
    }
    for (int w = tid; w < AW; w += FG_THREADS) tb[w] = 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) s_red[wave] = acc;      // (summed behind the first barrier below: nothing needs the index before the allocation)
    if (tid < 4 && tid >= FG_THREADS / 64) s_red[tid] = 0;
    const int depth0 = ti.depth, shift0 = ti.shift, cnt = ti.cnt, first = ti.first;
    if (d.maybe_dead && cnt > 64) {      // (rare: the live count of a wide target is not in one wavefront's ballot)
        if (tid == 0) s_live = 0;
        __syncthreads();
        int nl = 0;
        for (int i = tid; i < cnt; i += FG_THREADS) nl += (a.flags[first + i] & F_DEAD) ? 0 : 1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nl += __shfl_xor(nl, o);
        if (lane == 0 && nl) atomicAdd(&s_live, nl);
        __syncthreads();
    }
    const int curw = a.cur_slot_base >> 6;      // first word of this scan's measurement nodes in the association bitset

    // ---- the target's children: count, take a block of the node index space, emit -------------------------------------------
    // A chunk = CAP leaves, one per lane of wavefronts 0 and 1.  A target with more leaves runs the chunk loop twice: pass 0
    // only counts, pass 1 emits.
    const bool two_pass = cnt > CAP;
    int total = 0, run = 0, base = 0, fin_tot = 0;
    for (int pass = two_pass ? 0 : 1; pass < 2; ++pass) {
        for (int c0 = 0; c0 < cnt; c0 += CAP) {
            // (the loops exist for targets with more than CAP leaves only.  The argument block is re-read through an opaque
            // pointer in every iteration: otherwise every address and every uniform predicate of the unrolled body is
            // hoisted in front of the loops and held in registers across them -- +100 VGPRs, ~400 spilled SGPRs)
            int depth = __builtin_amdgcn_readfirstlane(depth0), shift = __builtin_amdgcn_readfirstlane(shift0);
            asm volatile("" : "+s"(depth), "+s"(shift));
            KArgs ap = ap0;
            asm volatile("" : "+s"(ap));
            const auto& a = *ap;
            const int n = (cnt - c0 < CAP) ? cnt - c0 : CAP;
            const bool first_emit = (pass == 1 && c0 == 0);
            // ---- phase 1: predict, one leaf per lane of wavefronts 0 and 1; the gains come from the table -----------------------
            if (!CMP) for (int w = tid; w < CAP * W; w += FG_THREADS) hw[w] = 0ull;      // (constant-turn kernel: cleared behind the candidate count, as many words as are used)
            if (tid == 0) s_ncand = 0;
            int last = -1, last2 = -1, nfv = 0, offv = 0;      // (last2, nfv, offv: AIS forest)
            if (wave < 2) {          // (both wavefronts whole: the box reduction below runs over all their lanes)
                const bool keep = tid < CAP;
                FLeaf g;
                const bool in_chunk = tid < n;
                const int src = first + c0 + (in_chunk ? tid : 0);
                // batch A: everything addressed by the leaf; nothing sits behind a branch
                const uint8_t fl = a.flags[src];
                const bool valid = in_chunk && !(fl & F_DEAD);      // (a leaf similar-state pruning took out of the tree: no children)
                if (pass == 1) {
                    const unsigned long long deadm = __ballot(in_chunk && !valid);
                    if (deadm && lane == 0) atomicAdd(&a.status->n_dead, __popcll(deadm));
                }
                const double cn = a.cnllr[src], pd = a.pd[src];
                const int covc = a.cov[src];
                double xd[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) xd[k] = a.x[(size_t)k * a.cap + src];
                // the leaf's path / ancestor records (pds ints each: 2 or 4 x 16 bytes)
                const int4* prec = reinterpret_cast<const int4*>(a.in_path + (size_t)src * PDS);
                const int4* arec = reinterpret_cast<const int4*>(a.in_apath + (size_t)src * PDS);
                int4 pq[PQ], aq[PQ];
#pragma unroll
                for (int q = 0; q < PQ; ++q) { pq[q] = prec[q]; aq[q] = arec[q]; }
                // batch B: the gains of the leaf's covariance column
                float4 gr[GKQ];
#pragma unroll
                for (int q = 0; q < GKQ; ++q) gr[q] = CT ? a.ct.gains[(size_t)src * GKQ + q] : a.vt.Gk[(size_t)covc * GKQ + q];
                int cid = CT ? src : a.vt.child[covc];      // (constant-turn forest: the children's keys are 2 * (leaf node) + hit/miss, mht_kernels.h CtGrow)
                if (AIS) {
                    FLeafX gx;
                    gx.f64 = (prom && in_chunk) ? 1 : 0; gx.pad = 0;
                    double row[GKF];
#pragma unroll
                    for (int e = 0; e < GKF; ++e) row[e] = 0.0;
                    if (gx.f64) {
                        if (fl & F_COV_F64) {
                            vt_load_gains64(a.vt, covc, row);
                        } else {      // a float32 leaf of a promoted target: its covariance converted (found or inserted as a float64 value), its gains from there
                            double P64[NP];
                            cid = vt_promote(a.vt, cid, pd, P64);
                            Model mg;
#pragma unroll
                            for (int e = 0; e < NP; ++e) { mg.A[e] = a.model.A[e]; mg.Q[e] = a.model.Q[e]; }
#pragma unroll
                            for (int e = 0; e < NK; ++e) mg.C[e] = a.model.C[e];
#pragma unroll
                            for (int e = 0; e < 4; ++e) mg.R[e] = a.model.R[e];
                            mg.eta2 = a.model.eta2; mg.lambda_ex = a.model.lambda_ex;
                            vt_gains64(mg, P64, pd, row);
                        }
                        // (the float32 fields of the leaf's record feed the pre-filter box only)
                        float* grw = reinterpret_cast<float*>(gr);
                        grw[GK_RX] = (float)row[GK_RX]; grw[GK_RY] = (float)row[GK_RY];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) gx.sinv[e] = row[e];
#pragma unroll
                    for (int e = 0; e < NK; ++e) gx.K[e] = row[4 + e];
                    gx.lnc = row[GK_LNC];
                    if (keep) lgx[tid] = gx;
                }
                g.valid = valid;
                g.src = src;
                g.flags = fl;
                g.f32state = ((fl & F_STATE_F32) && !(AIS && prom)) ? 1 : 0;
                g.cn = cn;
                g.pd = pd;
                // records parked raw (the root advance `shift` is applied when they are read back); last real measurement on the path
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
                        if (keep) {
                            reinterpret_cast<int4*>(s_pp + tid * PDS)[q] = pq[q];
                            reinterpret_cast<int4*>(s_ap + tid * PDS)[q] = aq[q];
                        }
                        const int pe[4] = {pq[q].x, pq[q].y, pq[q].z, pq[q].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = q * 4 + e;
                            if (!AIS) {
                                if (i >= shift && i < shift + depth && pe[e] >= 0) last = pe[e];
                            } else {      // two halves: the deepest level that has a row contributes its row(s); lvl_* ride in nfv / offv until the loads below
                                const int half = a.ais.half;
                                const int hb = (i >= half) ? half : 0, lvl = i - hb - shift;
                                if (i < 2 * half && lvl >= 0 && lvl < depth && pe[e] >= 0) {
                                    if (hb) { last2 = pe[e]; offv = lvl + 1; } else { last = pe[e]; nfv = lvl + 1; }
                                }
                            }
                        }
                    }
                if (AIS) {
                    if (nfv < offv) last = -1;          // (a deeper AIS-only level: the radar row further up belongs to an ancestor's own all-miss leaf)
                    if (offv < nfv) last2 = -1;
                    // A target whose root has not advanced yet still carries the association set spawnNewNodes built incrementally -- which never
                    // takes a fused child's RADAR measurement (pyTarget.py:292-295: only (scan, mmsi)); the set is rebuilt from the tree, fused
                    // children's radar measurements included, when the root advances (tracker.py:1222-1227, pyTarget.py:414-430; every scan from
                    // the first time on) or similar-state pruning meets the target alone in its cluster (tracker.py:1233-1239).  A fused level the
                    // last rebuild has not seen gives its AIS row only (rebuilt_levels: mht_kernels.h WIN_REBUILT_*).
                    if (nfv == offv && nfv > 0 && nfv > rebuilt_levels) last = -1;
                    nfv = 0; offv = 0;
                    if (valid && d.ais_on) { nfv = a.ais.nf[src]; offv = a.ais.off[src]; }
                    if (keep) { s_ais[tid * 4] = nfv; s_ais[tid * 4 + 1] = offv; s_ais[tid * 4 + 2] = a.ais.hmmsi_in[src]; s_ais[tid * 4 + 3] = 0; }
                }
                if (!valid) { last = -1; last2 = -1; }
                Model mdl;          // (only A and C are used: uniform registers)
#pragma unroll
                for (int e = 0; e < NP; ++e) mdl.A[e] = a.model.A[e];
#pragma unroll
                for (int e = 0; e < NK; ++e) mdl.C[e] = a.model.C[e];
                if (CT) {      // (the leaf's own Phi(T, w): predicted by forest_ct_kernel, in the reference's per-hypothesis order)
#pragma unroll
                    for (int k = 0; k < NX; ++k) g.xbar[k] = a.ct.xbar[(size_t)k * a.cap + src];
                    g.zhat[0] = a.ct.zhat[src]; g.zhat[1] = a.ct.zhat[(size_t)a.cap + src];
                } else if (g.f32state) {
                    float xs[NX], xb[NX], zh[2];
#pragma unroll
                    for (int k = 0; k < NX; ++k) xs[k] = (float)xd[k];
                    state_predict<float>(mdl, xs, xb, zh);
#pragma unroll
                    for (int k = 0; k < NX; ++k) g.xbar[k] = (double)xb[k];
                    g.zhat[0] = (double)zh[0]; g.zhat[1] = (double)zh[1];
                } else {
                    double xb[NX], zh[2];
                    state_predict<double>(mdl, xd, xb, zh);
#pragma unroll
                    for (int k = 0; k < NX; ++k) g.xbar[k] = xb[k];
                    g.zhat[0] = zh[0]; g.zhat[1] = zh[1];
                }
                {   // the gains row: S^-1 (4), K (NX x 2), score constant, gate half-axes (mht_vtab.h::vt_gains)
                    const float* grf = reinterpret_cast<const float*>(gr);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g.sinv[e] = grf[e];
#pragma unroll
                    for (int e = 0; e < NK; ++e) g.K[e] = grf[4 + e];
                    g.lnc = grf[GK_LNC];
                }
                const float zhx = (float)g.zhat[0], zhy = (float)g.zhat[1];
                const float rx = reinterpret_cast<const float*>(gr)[GK_RX], ry = reinterpret_cast<const float*>(gr)[GK_RY];
                // NIS <= eta2  =>  |dz_x| <= sqrt(eta2*S00), |dz_y| <= sqrt(eta2*S11); widened for the float32 rounding of the
                // pre-filter subtraction (coordinates up to ~1e6 m) -- the exact test decides, this only prunes
                const float bx = rx * 1.001f + 1e-6f * (fabsf(zhx) + rx) + 1e-3f;
                const float by = ry * 1.001f + 1e-6f * (fabsf(zhy) + ry) + 1e-3f;
                g.zhx = zhx; g.zhy = zhy; g.bx = bx; g.by = by;
                g.pad = 0;
                g.cid = cid;
                if (keep) lg[tid] = g;
                if (d.maybe_dead && cnt > 1 && cnt <= 64 && wave == 0) {      // live leaves of the target (cnt > 64: counted up front)
                    const int nl = __popcll(__ballot(valid));
                    if (lane == 0) s_live = nl;
                }
                // bounding box of the target's gates (its leaves sit within a few hundred metres of each other): the scan is first
                // cut down to the measurements inside it
                float lox = zhx - bx, hix = zhx + bx, loy = zhy - by, hiy = zhy + by;
                lox -= fabsf(lox) * 2.4e-7f + 1e-30f; hix += fabsf(hix) * 2.4e-7f + 1e-30f;      // outward: a superset of the leaf's own box
                loy -= fabsf(loy) * 2.4e-7f + 1e-30f; hiy += fabsf(hiy) * 2.4e-7f + 1e-30f;
                int b0 = valid ? fg_sortable(lox) : 0x7fffffff, b1 = valid ? fg_sortable(hix) : (int)0x80000000;
                int b2 = valid ? fg_sortable(loy) : 0x7fffffff, b3 = valid ? fg_sortable(hiy) : (int)0x80000000;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    b0 = min(b0, __shfl_xor(b0, o)); b1 = max(b1, __shfl_xor(b1, o));
                    b2 = min(b2, __shfl_xor(b2, o)); b3 = max(b3, __shfl_xor(b3, o));
                }
                if (lane == 0) { s_boxp[wave * 4] = b0; s_boxp[wave * 4 + 1] = b1; s_boxp[wave * 4 + 2] = b2; s_boxp[wave * 4 + 3] = b3; }
            }
            __syncthreads();
            FG_STAMP(2);
            // ONE live leaf in the target: the reference's per-target call hands a (4,4) x (4,1) product to BLAS gemv, whose rows are
            // not FMA chains (mht_math.h::gemv_row; dead leaves -- similar-state pruning, previous scan -- do not count).  Rare (a
            // target's first scan): the leaf's prediction is redone here, outside phase 1's register peak, from a second load of its
            // state; phase 2 (b) reads it behind the next barrier.  (The gate boxes come from the FMA-chain prediction: an ulp away,
            // well inside their widening.)
            if (!CT && __builtin_amdgcn_readfirstlane((cnt == 1 || (d.maybe_dead && cnt > 1 && s_live == 1)) ? 1 : 0) && tid < n) {
                const int src1 = first + c0 + tid;
                const uint8_t fl1 = a.flags[src1];
                if (!(fl1 & F_DEAD)) fg_single_leaf(a, src1, (fl1 & F_STATE_F32) != 0 && !(AIS && prom), lg[tid]);
            }
            // ---- phase 2 (a): measurements inside the target's box -> candidate list (ballot + one LDS atomic per wavefront) -----
            if (last >= 0) atomicOr(&tb[last >> 6], 1ull << (last & 63));      // (the bitset was cleared in front of the barrier)
            if (AIS) {
                if (last2 >= 0) atomicOr(&tb[last2 >> 6], 1ull << (last2 & 63));
                for (int f = 0; f < nfv; ++f) {      // the messages of the leaf's fused children (pyTarget.py:292-295: NOT their radar measurements)
                    const int j = M + a.ais.rec[offv + f].msg;
                    atomicOr(&tb[curw + (j >> 6)], 1ull << (j & 63));
                }
            }
            const int x0 = min(s_boxp[0], s_boxp[4]), x1 = max(s_boxp[1], s_boxp[5]);
            const int y0 = min(s_boxp[2], s_boxp[6]), y1 = max(s_boxp[3], s_boxp[7]);
            if (CMP) {      // the list in ASCENDING order (the hit masks index it, the children of a leaf are emitted in its order): one wavefront, no atomics
                if (wave == 0) {
                    int nc0 = 0;
                    for (int j0 = 0; j0 < Mpad; j0 += 64) {
                        const int j = j0 + lane;
                        const int kx = fg_sortable(zx[j]), ky = fg_sortable(zy[j]);
                        const bool in = (kx >= x0) && (kx <= x1) && (ky >= y0) && (ky <= y1);
                        const unsigned long long bal = __ballot(in);
                        if (in) cand[nc0 + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)j;
                        nc0 += __popcll(bal);
                    }
                    if (lane == 0) s_ncand = nc0;
                }
            } else
            for (int j0 = 0; j0 < Mpad; j0 += FG_THREADS) {
                const int j = j0 + tid;
                bool in = false;
                if (j < Mpad) {
                    const int kx = fg_sortable(zx[j]), ky = fg_sortable(zy[j]);
                    in = (kx >= x0) && (kx <= x1) && (ky >= y0) && (ky <= y1);
                }
                const unsigned long long bal = __ballot(in);
                int wbase = 0;
                if (lane == 0 && bal) wbase = atomicAdd(&s_ncand, __popcll(bal));
                wbase = __shfl(wbase, 0);
                if (in) cand[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)j;
            }
            __syncthreads();
            FG_STAMP(3);
            // ---- phase 2 (b): thread = (leaf, candidate): the leaf's own conservative float32 box, then the exact reference-order NIS
            // (constant-turn kernel: where the leaves' hit words live for this chunk -- LDS, bits = positions in the candidate list, or the slot's
            // block of global memory with full-width masks when the list is longer than FG_HWC words)
            const int nc_all = s_ncand;
            const int nblk = CMP ? (nc_all + 63) >> 6 : W;
            const bool spill = CMP && (nblk > FG_HWC || d.ct_spill);      // (ct_spill: testing, MHT_CT_SPILL=1 -- every target through the global block)
            unsigned long long* hwg = (CMP && spill) ? a.ct.hw_spill + (size_t)bslot * CAP * W : nullptr;
            const int hstride = CMP ? (spill ? W : nblk) : W;
            if (CMP) {
                if (spill) { for (int w = tid; w < n * W; w += FG_THREADS) hwg[w] = 0ull; }
                else { for (int w = tid; w < n * nblk; w += FG_THREADS) hw[w] = 0ull; }
                __syncthreads();
            }
            {
                const int nc = s_ncand;
                const int sh = (n > 1) ? 32 - __clz(n - 1) : 0;      // leaves padded to a power of two
                for (int w = tid; w < (nc << sh); w += FG_THREADS) {
                    const int l = w & ((1 << sh) - 1), ci = w >> sh, j = cand[ci];
                    if (l >= n) continue;
                    const FLeaf& g = lg[l];
                    if (!g.valid) continue;
                    const float mx = zx[j], my = zy[j];
                    if ((fabsf(mx - g.zhx) <= g.bx) && (fabsf(my - g.zhy) <= g.by)) {
                        bool hit;
                        if (g.f32state) {
                            float zh[2] = {(float)g.zhat[0], (float)g.zhat[1]}, zt[2], nis;
                            hit = gate_pair<float>(zh, g.sinv, mx, my, (float)a.model.eta2, zt, nis);
                        } else if (AIS && lgx[l].f64) {
                            double zh[2] = {g.zhat[0], g.zhat[1]}, zt[2], nis;
                            hit = gate_pair<double>(zh, lgx[l].sinv, mx, my, a.model.eta2, zt, nis);
                        } else {
                            double zh[2] = {g.zhat[0], g.zhat[1]}, zt[2], nis;
                            hit = gate_pair<double>(zh, g.sinv, mx, my, a.model.eta2, zt, nis);
                        }
                        if (hit) {
                            if (!CMP) atomicOr(&hw[(size_t)l * W + (j >> 6)], 1ull << (j & 63));
                            else if (spill) atomicOr(&hwg[(size_t)l * W + (j >> 6)], 1ull << (j & 63));
                            else atomicOr(&hw[(size_t)l * nblk + (ci >> 6)], 1ull << (ci & 63));
                            atomicOr(&tb[curw + (j >> 6)], 1ull << (j & 63));
                        }
                    }
                }
            }
            __syncthreads();
            FG_STAMP(4);
            // ---- child counts (1 missed detection + hits per leaf), exclusive prefix, and -- before the first emission -- the
            //      target's block of the node index space and its slice of the edge list: wavefront 0 / wavefront 1
            if (wave == 0) {
                int h0 = 0, h1 = 0;
                const int l1 = (lane + 64 < CAP) ? lane + 64 : lane;      // (second half of the chunk: lanes beyond it re-read their own row)
                if (CMP && spill) { for (int w = 0; w < W; ++w) { h0 += __popcll(hwg[(size_t)lane * W + w]); h1 += __popcll(hwg[(size_t)l1 * W + w]); } }
                else if (CMP) { if (lane < n) for (int w = 0; w < nblk; ++w) h0 += __popcll(hw[(size_t)lane * nblk + w]); if (lane + 64 < n) for (int w = 0; w < nblk; ++w) h1 += __popcll(hw[(size_t)l1 * nblk + w]); }
                else
                for (int w = 0; w < W; ++w) { h0 += __popcll(hw[(size_t)lane * W + w]); h1 += __popcll(hw[(size_t)l1 * W + w]); }
                if (AIS) { h0 += s_ais[lane * 4]; h1 += s_ais[l1 * 4]; }      // (fused children behind the radar children)
                const int m0 = (lane < n && lg[lane].valid) ? 1 + h0 : 0, m1 = (lane + 64 < n && lg[l1].valid) ? 1 + h1 : 0;
                int i0 = m0, i1 = m1;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v0 = __shfl_up(i0, o), v1 = __shfl_up(i1, o);
                    if (lane >= o) { i0 += v0; i1 += v1; }
                }
                const int t0 = __shfl(i0, 63);
                i1 += t0;
                s_pref[lane] = i0 - m0;
                if (lane + 64 < CAP) s_pref[64 + lane] = i1 - m1;
                for (int q = 0, p = i0 - m0; q < m0 && p < FG_MAP; ++q, ++p) s_map[p] = (unsigned char)lane;          // child -> leaf
                for (int q = 0, p = i1 - m1; q < m1 && p < FG_MAP; ++q, ++p) s_map[p] = (unsigned char)(lane + 64);
                const int chunk_total = __shfl(i1, 63);
                if (lane == 63) { s_pref[CAP] = chunk_total; s_total = chunk_total; }
                if (first_emit && lane == 0) {
                    const int tot = two_pass ? total : chunk_total;
                    // the target's block of the node index space: its slot's own static block (no atomic: nothing downstream needs a
                    // dense numbering, the index space is sized for 288 GB of HBM) or, for a target with more children than that, a
                    // piece of this XCD's region of the overflow area (one returning atomic; next region if full)
                    int b = -1;
                    if (tot <= a.block_cap && bslot < a.Tcap) {
                        b = bslot * a.block_cap;
                    } else {
                        int r = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);      // XCC_ID[3:0]
                        for (int tries = 0; tries < FG_REGIONS && b < 0; ++tries) {
                            const unsigned old = atomicAdd(&a.alloc[r * 32], (unsigned)tot);
                            if (old + (unsigned)tot <= (unsigned)a.region_cap) b = a.over_base + r * a.region_cap + (int)old;
                            else r = (r + 1) & (FG_REGIONS - 1);
                        }
                    }
                    if (b < 0) a.status->overflow = 1;      // every region is full: the scan is void (MHT_E_CAPACITY)
                    s_base = b;
                }
            } else if (wave == FG_THREADS / 64 - 1 && first_emit && d.uf_epoch) {
                // no edge list: the target joins the device-wide union-find.  Its nodes' owner words are exchanged here, next to the
                // counts (the answers are back before wavefront 0 is through its prefix); the links go out next to the emission
                // (overlapping launch: the compacted index is not known yet -- hooked under the SLOT, which is the index unless a target died in
                // the previous scan; the end of the workgroup redoes it in that case)
                const int pos = ovl ? t : (d.fused ? (s_red[0] + s_red[1] + s_red[2] + s_red[3]) : t);
                uf_claim(a.uf_owner, a.uf_parent, d.uf_epoch, pos, tb, AW, lane, CMP ? s_conf : reinterpret_cast<int*>(cand), CMP ? FG_CONF : Mpad / 2, &s_misc[17]);
            } else if (wave == 1 && first_emit && !d.uf_epoch) {
                // edges of the clustering graph = set bits of the association bitset (complete here: every leaf's last real
                // measurement and the hits; with two passes the count pass has seen all chunks)
                int ne = 0;
                for (int w = lane; w < AW; w += 64) ne += __popcll(tb[w]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ne += __shfl_xor(ne, o);
                if (lane == 0) {
                    const int seg = blockIdx.x & (EDGE_SEGS - 1);
                    const int e0 = atomicAdd(&a.edge_count[seg], ne);
                    s_ebase = e0;
                    if (e0 + ne > a.edge_cap) a.status->overflow = 1;
                }
            }
            __syncthreads();
            FG_STAMP(5);
            if (pass == 0) {
                total += s_total;
                __syncthreads();
                continue;
            }
            if (first_emit) {
                base = s_base;
                fin_tot = two_pass ? total : s_total;
                // the target's entries of the child tables go out behind the barrier: in front of it the barrier's release waited for
                // the acknowledgement of these global stores (~1 us on the workgroup's critical path)
                if (tid == 0) {
                    const int tot = fin_tot;
                    atomicAdd(&a.status->n_children, tot);
                    if (!ovl) {      // (overlapping launch: the compacted index is not known yet -- at the end of the workgroup)
                        const int pos = d.fused ? (s_red[0] + s_red[1] + s_red[2] + s_red[3]) : t;
                        a.tchild[pos] = base < 0 ? 0 : base;
                        a.tcend[pos] = base < 0 ? 0 : base + tot;
                    }
                }
                if (base < 0) return;
                if (d.uf_epoch) {      // the targets this one shares a node with (uf_claim above): one link each (the last wavefront: emission reaches it last)
                    if (wave == FG_THREADS / 64 - 1) {
                        const int pos = ovl ? t : (d.fused ? (s_red[0] + s_red[1] + s_red[2] + s_red[3]) : t);
                        const int* conf = CMP ? s_conf : reinterpret_cast<const int*>(cand);
                        for (int i = lane; i < s_misc[17]; i += 64) uf_link(a.uf_parent, d.uf_epoch, pos, conf[i]);
                    }
                } else if (wave == 1) {      // edge list: (target << 16 | node) for every set bit (this wavefront's share of phase 4 is the lightest)
                    const int pos = d.fused ? (s_red[0] + s_red[1] + s_red[2] + s_red[3]) : t;
                    const int seg = blockIdx.x & (EDGE_SEGS - 1);
                    int eb = s_ebase;
                    for (int w0 = 0; w0 < AW; w0 += 64) {
                        const int w = w0 + lane;
                        unsigned long long bits = (w < AW) ? tb[w] : 0ull;
                        const int pc = __popcll(bits);
                        int incl = pc;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const int v = __shfl_up(incl, o);
                            if (lane >= o) incl += v;
                        }
                        int my = eb + incl - pc;
                        while (bits) {
                            const int bpos = __ffsll((long long)bits) - 1;
                            bits &= bits - 1;
                            if (my < a.edge_cap) a.edges[(size_t)seg * a.edge_cap + my] = ((unsigned)pos << 16) | (unsigned)(w * 64 + bpos);
                            ++my;
                        }
                        eb += __shfl(incl, 63);
                    }
                }
            }
            // ---- phase 4: lane = child, wavefront = role; children of the chunk at base + run .. --------------------------------------
            FG_STAMP(6);
            // ---- phase 4: one thread per child; children of the chunk at base + run .. ------------------------------------------------
            {
                const int ctot = s_pref[CAP];
                for (int r = tid; r < ctot; r += FG_THREADS) {
                    int l;
                    if (r < FG_MAP) {
                        l = s_map[r];                        // child -> leaf table written with the counts
                    } else {                                 // (more children than the table holds: search the prefix)
                        int lo = 0, hi = CAP;
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (s_pref[mid] <= r) lo = mid; else hi = mid;
                        }
                        l = lo;
                    }
                    const int k = r - s_pref[l], c = base + run + r;
                    const int nh = s_pref[l + 1] - s_pref[l] - 1;      // gated measurements of the leaf (lanes beyond the chunk carry the total)
                    // the leaf's record and the child's path / ancestor sources in ONE batch of wide LDS reads (field-by-field reads
                    // behind the branches below were ~40 dependent LDS round trips per child: 2 us)
                    FG_STAMPX(1);
                    FLeaf g;
                    {
                        const uint4* srcq = reinterpret_cast<const uint4*>(lg + l);
                        uint4* dstq = reinterpret_cast<uint4*>(&g);
#pragma unroll
                        for (int q = 0; q < (int)(sizeof(FLeaf) / 16); ++q) dstq[q] = srcq[q];
                    }
                    if (AIS) {
                        const int nf = s_ais[l * 4], nhr = nh - nf;      // (nh counted the fused children too)
                        if (k > nhr) {
                            fg_emit_fused<PQ>(a, d, g, l, c, a.ais.rec[s_ais[l * 4 + 1] + (k - nhr - 1)], s_pp, s_ap, depth, shift, rootc);
                        } else {
                            a.ais.ommsi[c] = 0;
                            a.ais.ohmmsi[c] = s_ais[l * 4 + 2];
                            FLeafX gx;
                            {
                                const uint4* srcq = reinterpret_cast<const uint4*>(lgx + l);
                                uint4* dstq = reinterpret_cast<uint4*>(&gx);
#pragma unroll
                                for (int q = 0; q < (int)(sizeof(FLeafX) / 16); ++q) dstq[q] = srcq[q];
                            }
                            if (g.f32state) fg_emit_child<float, PQ, 1>(a, d, -1, g, l, c, k, nhr, hw + (size_t)l * W, zx, zy, s_pp, s_ap, depth, shift, rootc, root_f32);
                            else fg_emit_child<double, PQ, 1>(a, d, -1, g, l, c, k, nhr, hw + (size_t)l * W, zx, zy, s_pp, s_ap, depth, shift, rootc, root_f32, nullptr, nullptr, &gx);
                        }
                    } else {
                        // (constant-turn kernel: the leaf's hit words index the candidate list -- unless they were spilled, full width, to global memory)
                        const unsigned long long* hwl = (CMP && spill) ? hwg + (size_t)l * W : hw + (size_t)l * hstride;
                        const unsigned short* cmap = (CMP && !spill) ? cand : nullptr;
                        if (g.f32state) fg_emit_child<float, PQ>(a, d, -1, g, l, c, k, nh, hwl, zx, zy, s_pp, s_ap, depth, shift, rootc, root_f32, cmap);
                        else fg_emit_child<double, PQ>(a, d, -1, g, l, c, k, nh, hwl, zx, zy, s_pp, s_ap, depth, shift, rootc, root_f32, cmap);
                    }
                }
                run += ctot;
            }
            if (c0 + CAP < cnt) __syncthreads();      // the chunk tables are re-used
        }
    }
    FG_STAMP(7);
    if (ovl) {
        // Overlapping launch: the target's compacted index -- tchild / tcend and the union-find are indexed by it -- comes from the commit
        // in workgroup 0 of this launch, which had to wait for the LAST workgroup of the previous scan's ILP launch.  Everything else of
        // the target is done; its entries and its links follow the moment the index is there.
        // (Nearly always the index IS the slot -- no target died in the previous scan, the commit's word says so -- and the target's
        // place in the union-find, taken under the slot next to the emission, stands.  Otherwise: once more, under the compacted index and
        // the scan's alternative epoch, which the ILP launch then reads.)
        __syncthreads();      // (the candidate list's LDS is free: the union-find's list goes there)
        target_tail(a, d, t, base, fin_tot, tb, reinterpret_cast<int*>(cand), Mpad / 2, &s_misc[17]);
    }
}

// ---- wavefront-per-target variant (batched launches: several sectors' targets resident at once) -----------------------------
// A target has ~27 leaves and ~25 gated pairs: a 256-thread workgroup per target (target_part) spends most of its 10 us waiting --
// four dependent round trips and five barriers -- with 44 KB of LDS and 16 wavefront slots held per CU by four targets.  Here ONE
// WAVEFRONT runs a target (four independent targets per 256-thread workgroup, no barrier between them), with ~9 KB of LDS:
//   * lane = leaf (FW_LP per pass): records, gains, prediction, box -- as target_part's phase 1;
//   * the scan is not staged in LDS: the lanes sweep it from global memory (4 KB, L2 resident) against the target's box; the
//     candidates (a handful) go to an LDS list in ascending measurement order;
//   * lane = (leaf, candidate): the exact gate; a leaf's hits are bits over the CANDIDATE list (one 64-bit word per 64 candidates and
//     leaf, not ceil(M / 64) words per leaf);
//   * lane = leaf: child counts, wave prefix; lane = child: emission (fg_emit_child).
// Wave-synchronous: the phases of a wavefront are ordered by the in-order LDS pipeline and a wave-level fence, not by barriers.
// Everything a target produces is what target_part produces for it (same blocks, same child order, same edge records).
#ifndef MHT_FW_LP
#define MHT_FW_LP 64
#endif
constexpr int FW_LP = MHT_FW_LP;      // leaves per pass, one per lane (32: 40 KB per workgroup, three per CU, but a third of the targets needs two passes: slower)
constexpr int FW_MASKW = 64;          // 64-bit hit words of a wavefront: FW_LP leaves x up to 128 candidates in one pass; more candidates: fewer leaves per pass
constexpr int FW_CZ = 64;             // candidates whose coordinates are kept in LDS (the rest are re-read from the scan)
constexpr int FW_MAP = 256;           // child -> leaf table entries (more children: binary search)
struct FWLayout { int lf, pp, ap, mask, tb, cand, candz, pref, map, total; };      // byte offsets inside a wavefront's LDS slice
__host__ __device__ __forceinline__ FWLayout fw_layout(int pds, int AW, int Mpad) {
    FWLayout o;
    int b = 0;
    o.lf = b; b += FW_LP * (int)sizeof(FLeaf);
#ifdef MHT_FW_PARK_RECORDS      // (until the end of round 5: the leaves' path / ancestor records parked in LDS for the emission, 4 KB of a wavefront's 16 KB)
    o.pp = b; b += FW_LP * pds * 4;
    o.ap = b; b += FW_LP * pds * 4;
#else
    o.pp = b; o.ap = b;      // (the emission reads them where phase 1 read them: the leaves of a pass are consecutive nodes, their records consecutive in global memory and in this XCD's L2)
#endif
    o.mask = b; b += FW_MASKW * 8;
    o.tb = b; b += AW * 8;
    o.cand = b; b += (Mpad * 2 + 15) & ~15;
    o.candz = b; b += FW_CZ * 8;
    o.pref = b; b += ((FW_LP + 4) * 4 + 15) & ~15;
    o.map = b; b += FW_MAP;
    o.total = (b + 15) & ~15;
    return o;
}
#ifdef MHT_GROW_STAMPS
#define FW_STAMP(k) do { if (d.dbg && lane == 0 && t < 3900) d.dbg[32 + (size_t)t * 16 + (k)] = wall_clock64(); } while (0)
#else
#define FW_STAMP(k)
#endif
#define FW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <int PQ>
__device__ __forceinline__ void target_wave(KArgs ap0, const FDyn& d, int t, unsigned char* sm) {
    constexpr int PDS = PQ * 4;
    constexpr int LP = FW_LP;
    const auto& a = *ap0;
    const int lane = threadIdx.x & 63;
    const int M = d.M, W = d.W, Mpad = W * 64, AW = a.AW;
    const FWLayout lo = fw_layout(PDS, AW, Mpad);
    FLeaf* lg = reinterpret_cast<FLeaf*>(sm + lo.lf);
#ifdef MHT_FW_PARK_RECORDS
    int* s_pp = reinterpret_cast<int*>(sm + lo.pp);
    int* s_ap = reinterpret_cast<int*>(sm + lo.ap);
#endif
    unsigned long long* mk = reinterpret_cast<unsigned long long*>(sm + lo.mask);
    unsigned long long* tb = reinterpret_cast<unsigned long long*>(sm + lo.tb);
    unsigned short* cand = reinterpret_cast<unsigned short*>(sm + lo.cand);
    float2* candz = reinterpret_cast<float2*>(sm + lo.candz);
    int* s_pref = reinterpret_cast<int*>(sm + lo.pref);
    unsigned char* s_map = reinterpret_cast<unsigned char*>(sm + lo.map);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // ---- first round trip: everything addressed by the target slot alone ------------------------------------------------------------
    FW_STAMP(0);
    const int nT = a.nT_dev[0];
    const int po = a.prev_status->overflow, so = *a.sticky_overflow;
    const TInfo ti = target_info<false>(a, d, t, nT);
    const int tc = (t < a.Tcap) ? t : 0;
    const double rootc = a.t_root_cnllr[tc];
    const int root_f32 = a.t_root_f32[tc];
    int acc = 0;
    if (d.fused)          // compacted index of this target = alive slots before it (eight look-ups per batch: one round trip each)
        for (int i0 = 0; i0 < t; i0 += 512) {
            int v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int i = i0 + q * 64 + lane; v[q] = a.p_status[i < t ? i : 0]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += (i0 + q * 64 + lane < t && v[q] == 0) ? 1 : 0;
        }
    if (po || so) {          // a scan that overflowed its pools voids every scan after it
        if (t == 0 && lane == 0) a.status->overflow = po ? po : 1;
        return;
    }
    if (!ti.alive) return;
    FW_STAMP(1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const int pos = d.fused ? __builtin_amdgcn_readfirstlane(acc) : t;
    for (int w = lane; w < AW; w += 64) tb[w] = 0ull;
    const int depth0 = ti.depth, shift0 = ti.shift;
    const int cnt = __builtin_amdgcn_readfirstlane(ti.cnt), first = __builtin_amdgcn_readfirstlane(ti.first);
    const int curw = a.cur_slot_base >> 6;
    const float2* z2 = reinterpret_cast<const float2*>(d.z);
    // ONE live leaf in the target: gemv order of its prediction (see target_part / mht_math.h::gemv_row)
    bool single = (cnt == 1);
    if (d.maybe_dead && cnt > 1) {
        int nl = 0;
        for (int i = lane; i < cnt; i += 64) nl += (a.flags[first + i] & F_DEAD) ? 0 : 1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nl += __shfl_xor(nl, o);
        single = __builtin_amdgcn_readfirstlane(nl) == 1;
    }

    // The children go to the slot's static block of the node index space, chunk after chunk, as long as they fit (optimistic: no
    // counting pass -- a target of 33..64 leaves takes two chunk iterations, not four).  A target whose children outgrow the block
    // (rare) is counted first (pass 0) and then emitted into a piece of the overflow area.
    int total = 0, run = 0, base = 0, ndead = 0;
    int pass = 1;
    bool counted = false;      // the counting pass has run
    for (; pass < 2; ++pass) {
        int c0 = 0;
        bool redo = false;
        while (c0 < cnt) {
            int depth = __builtin_amdgcn_readfirstlane(depth0), shift = __builtin_amdgcn_readfirstlane(shift0);
            asm volatile("" : "+s"(depth), "+s"(shift));
            KArgs ap = ap0;
            asm volatile("" : "+s"(ap));
            const auto& a = *ap;
            const int n = (cnt - c0 < LP) ? cnt - c0 : LP;
            // ---- phase 1: lane = leaf ----------------------------------------------------------------------------------------------
            int last = -1;
            bool valid;
            {
                const bool in_chunk = lane < n;
                const int src = first + c0 + (in_chunk ? lane : 0);
                const uint8_t fl = a.flags[src];
                valid = in_chunk && !(fl & F_DEAD);
                const double cn = a.cnllr[src], pd = a.pd[src];
                const int covc = a.cov[src];
                double xd[NX];
#pragma unroll
                for (int k = 0; k < NX; ++k) xd[k] = a.x[(size_t)k * a.cap + src];
                const int4* prec = reinterpret_cast<const int4*>(a.in_path + (size_t)src * PDS);
                const int4* arec = reinterpret_cast<const int4*>(a.in_apath + (size_t)src * PDS);
                int4 pq[PQ], aq[PQ];
#pragma unroll
                for (int q = 0; q < PQ; ++q) { pq[q] = prec[q]; aq[q] = arec[q]; }
                float4 gr[GKQ];
#pragma unroll
                for (int q = 0; q < GKQ; ++q) gr[q] = a.vt.Gk[(size_t)covc * GKQ + q];
                const int cid = a.vt.child[covc];
                FLeaf g;
                g.valid = valid; g.src = src; g.flags = fl; g.f32state = (fl & F_STATE_F32) ? 1 : 0; g.cn = cn; g.pd = pd;
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
#ifdef MHT_FW_PARK_RECORDS
                    if (lane < LP) {
                        reinterpret_cast<int4*>(s_pp + lane * PDS)[q] = pq[q];
                        reinterpret_cast<int4*>(s_ap + lane * PDS)[q] = aq[q];
                    }
#endif
                    const int pe[4] = {pq[q].x, pq[q].y, pq[q].z, pq[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = q * 4 + e;
                        if (i >= shift && i < shift + depth && pe[e] >= 0) last = pe[e];
                    }
                }
                if (!valid) last = -1;
                Model mdl;
#pragma unroll
                for (int e = 0; e < NP; ++e) mdl.A[e] = a.model.A[e];
#pragma unroll
                for (int e = 0; e < NK; ++e) mdl.C[e] = a.model.C[e];
                if (g.f32state) {
                    float xs[NX], xb[NX], zh[2];
#pragma unroll
                    for (int k = 0; k < NX; ++k) xs[k] = (float)xd[k];
                    state_predict<float>(mdl, xs, xb, zh);
#pragma unroll
                    for (int k = 0; k < NX; ++k) g.xbar[k] = (double)xb[k];
                    g.zhat[0] = (double)zh[0]; g.zhat[1] = (double)zh[1];
                } else {
                    double xb[NX], zh[2];
                    state_predict<double>(mdl, xd, xb, zh);
#pragma unroll
                    for (int k = 0; k < NX; ++k) g.xbar[k] = xb[k];
                    g.zhat[0] = zh[0]; g.zhat[1] = zh[1];
                }
                const float* grf = reinterpret_cast<const float*>(gr);
#pragma unroll
                for (int e = 0; e < 4; ++e) g.sinv[e] = grf[e];
#pragma unroll
                for (int e = 0; e < NK; ++e) g.K[e] = grf[4 + e];
                g.lnc = grf[GK_LNC];
                const float zhx = (float)g.zhat[0], zhy = (float)g.zhat[1];
                const float rx = grf[GK_RX], ry = grf[GK_RY];
                const float bx = rx * 1.001f + 1e-6f * (fabsf(zhx) + rx) + 1e-3f;      // (the leaf's conservative float32 box: target_part)
                const float by = ry * 1.001f + 1e-6f * (fabsf(zhy) + ry) + 1e-3f;
                g.zhx = zhx; g.zhy = zhy; g.bx = bx; g.by = by; g.pad = 0; g.cid = cid;
                if (lane < LP) lg[lane] = g;
                // bounding box of the chunk's gates
                float lox = zhx - bx, hix = zhx + bx, loy = zhy - by, hiy = zhy + by;
                lox -= fabsf(lox) * 2.4e-7f + 1e-30f; hix += fabsf(hix) * 2.4e-7f + 1e-30f;
                loy -= fabsf(loy) * 2.4e-7f + 1e-30f; hiy += fabsf(hiy) * 2.4e-7f + 1e-30f;
                int b0 = valid ? fg_sortable(lox) : 0x7fffffff, b1 = valid ? fg_sortable(hix) : (int)0x80000000;
                int b2 = valid ? fg_sortable(loy) : 0x7fffffff, b3 = valid ? fg_sortable(hiy) : (int)0x80000000;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    b0 = min(b0, __shfl_xor(b0, o)); b1 = max(b1, __shfl_xor(b1, o));
                    b2 = min(b2, __shfl_xor(b2, o)); b3 = max(b3, __shfl_xor(b3, o));
                }
                if (pass == 1) ndead += __popcll(__ballot(in_chunk && !valid));
                FW_SYNC();
                FW_STAMP(2);
                if (single && valid) fg_single_leaf(a, src, (fl & F_STATE_F32) != 0, lg[lane]);
                if (last >= 0) atomicOr(&tb[last >> 6], 1ull << (last & 63));
                // ---- phase 2 (a): the measurements inside the box, ascending -> candidate list -----------------------------------------
                const int x0 = __builtin_amdgcn_readfirstlane(b0), x1 = __builtin_amdgcn_readfirstlane(b1);
                const int y0 = __builtin_amdgcn_readfirstlane(b2), y1 = __builtin_amdgcn_readfirstlane(b3);
                int ncand = 0;
                for (int jb = 0; jb < M; jb += 512) {      // eight words of the scan per batch of loads (one round trip)
                    float2 zv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int j = jb + q * 64 + lane; zv[q] = z2[j < M ? j : 0]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int j = jb + q * 64 + lane;
                        const int kx = fg_sortable(zv[q].x), ky = fg_sortable(zv[q].y);
                        const bool in = (j < M) && (kx >= x0) && (kx <= x1) && (ky >= y0) && (ky <= y1);
                        const unsigned long long bal = __ballot(in);
                        if (in) {
                            const int ci = ncand + __popcll(bal & lt_mask);
                            cand[ci] = (unsigned short)j;
                            if (ci < FW_CZ) candz[ci] = zv[q];
                        }
                        ncand += __popcll(bal);
                    }
                }
                FW_STAMP(3);
                // leaves of this pass: all n unless their hit words do not fit (a box with more than 256 candidates)
                const int nblk = (ncand + 63) >> 6;
                int n_eff = n;
                if (n * nblk > FW_MASKW) n_eff = FW_MASKW / nblk > 0 ? FW_MASKW / nblk : 1;
                for (int w = lane; w < n_eff * nblk; w += 64) mk[w] = 0ull;
                FW_SYNC();
                // ---- phase 2 (b): lane = (leaf, candidate): the leaf's float32 box, then the exact reference-order NIS -----------------
                {
                    const int sh = (n_eff > 1) ? 32 - __clz(n_eff - 1) : 0;
                    for (int w = lane; w < (ncand << sh); w += 64) {
                        const int l = w & ((1 << sh) - 1), ci = w >> sh;
                        if (l >= n_eff) continue;
                        const FLeaf& gl = lg[l];
                        if (!gl.valid) continue;
                        const int j = cand[ci];
                        const float2 mv = (ci < FW_CZ) ? candz[ci] : z2[j];
                        if ((fabsf(mv.x - gl.zhx) <= gl.bx) && (fabsf(mv.y - gl.zhy) <= gl.by)) {
                            bool hit;
                            if (gl.f32state) {
                                float zh[2] = {(float)gl.zhat[0], (float)gl.zhat[1]}, zt[2], nis;
                                hit = gate_pair<float>(zh, gl.sinv, mv.x, mv.y, (float)a.model.eta2, zt, nis);
                            } else {
                                double zh[2] = {gl.zhat[0], gl.zhat[1]}, zt[2], nis;
                                hit = gate_pair<double>(zh, gl.sinv, mv.x, mv.y, a.model.eta2, zt, nis);
                            }
                            if (hit) {
                                atomicOr(&mk[l * nblk + (ci >> 6)], 1ull << (ci & 63));
                                atomicOr(&tb[curw + (j >> 6)], 1ull << (j & 63));
                            }
                        }
                    }
                }
                FW_SYNC();
                FW_STAMP(4);
                // ---- child counts, wave prefix, block of the node index space, edge slice -------------------------------------------------
                int mine = 0;
                if (lane < n_eff && lg[lane].valid) {
                    mine = 1;
                    for (int b = 0; b < nblk; ++b) mine += __popcll(mk[lane * nblk + b]);
                }
                int incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                const int ctot = __shfl(incl, 63);
                if (lane <= LP && lane < 64) s_pref[lane] = incl - mine;
                if (LP == 64 && lane == 63) s_pref[64] = incl;
                for (int q = 0, p = incl - mine; q < mine && p < FW_MAP; ++q, ++p) s_map[p] = (unsigned char)lane;
                if (pass == 0) {
                    total += ctot;
                    c0 += n_eff;
                    FW_SYNC();
                    continue;
                }
                if (c0 == 0) {      // where the children go
                    if (!counted) {
                        base = t * a.block_cap;
                    } else {
                        int b = -1;
                        if (lane == 0) {
                            if (total <= a.block_cap) {
                                b = t * a.block_cap;
                            } else {
                                int r = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);      // XCC_ID[3:0]
                                for (int tries = 0; tries < FG_REGIONS && b < 0; ++tries) {
                                    const unsigned old = atomicAdd(&a.alloc[r * 32], (unsigned)total);
                                    if (old + (unsigned)total <= (unsigned)a.region_cap) b = a.over_base + r * a.region_cap + (int)old;
                                    else r = (r + 1) & (FG_REGIONS - 1);
                                }
                            }
                            if (b < 0) { a.status->overflow = 1; a.tchild[pos] = 0; a.tcend[pos] = 0; }      // every region is full: the scan is void (MHT_E_CAPACITY)
                        }
                        base = __shfl(b, 0);
                        if (base < 0) return;
                    }
                }
                if (!counted && run + ctot > a.block_cap) {      // the static block is too small for this target: count, then take a piece of the overflow area
                    redo = true;
                    c0 = cnt;
                    continue;
                }
                FW_SYNC();
                FW_STAMP(5);
                // ---- emission: lane = child ---------------------------------------------------------------------------------------------
                const float* zdummy = reinterpret_cast<const float*>(s_pref);      // (the LDS scan copy of target_part: not used, the scan is read through z2)
                for (int r = lane; r < ctot; r += 64) {
                    int l;
                    if (r < FW_MAP) {
                        l = s_map[r];
                    } else {
                        int lo2 = 0, hi2 = n_eff;
                        while (hi2 - lo2 > 1) {
                            const int mid = (lo2 + hi2) >> 1;
                            if (s_pref[mid] <= r) lo2 = mid; else hi2 = mid;
                        }
                        l = lo2;
                    }
                    const int k = r - s_pref[l], c = base + run + r;
                    const int nh = s_pref[l + 1] - s_pref[l] - 1;
                    FLeaf gc;
                    {
                        const uint4* srcq = reinterpret_cast<const uint4*>(lg + l);
                        uint4* dstq = reinterpret_cast<uint4*>(&gc);
#pragma unroll
                        for (int q = 0; q < (int)(sizeof(FLeaf) / 16); ++q) dstq[q] = srcq[q];
                    }
#ifdef MHT_FW_PARK_RECORDS
                    const int* e_pp = s_pp; const int* e_ap = s_ap;
#else
                    const int* e_pp = a.in_path + (size_t)(first + c0) * PDS; const int* e_ap = a.in_apath + (size_t)(first + c0) * PDS;      // (leaf l of the pass = node first + c0 + l)
#endif
                    if (gc.f32state) fg_emit_child<float, PQ>(a, d, -1, gc, l, c, k, nh, mk + l * nblk, zdummy, zdummy, e_pp, e_ap, depth, shift, rootc, root_f32, cand, z2);
                    else fg_emit_child<double, PQ>(a, d, -1, gc, l, c, k, nh, mk + l * nblk, zdummy, zdummy, e_pp, e_ap, depth, shift, rootc, root_f32, cand, z2);
                }
                FW_STAMP(6);
                // The gains one scan ahead (chain_part's job, folded into the target's wavefront here: a launch of many sectors has no idle
                // workgroup slots for separate chain workgroups): are the hit / miss transitions of these leaves' covariances known?  Two
                // 4-byte look-ups per leaf behind the emission (the leaf's value id is in its LDS record), nearly always yes; what
                // nobody has computed yet (~10 per scan in steady state) is resolved here, the distinct ones spread over the lanes.
#ifndef MHT_FW_NOFOLD
                {
                    const bool mine_l = lane < n_eff;
                    const int cidl = mine_l ? lg[lane].cid : 0;
                    const int k0 = a.vt.child[2 * cidl], k1 = a.vt.child[2 * cidl + 1];
                    unsigned long long rem = __ballot(mine_l && (k0 < 0 || k1 < 0));
                    while (rem) {      // (rounds of 32 distinct values: one is the rule)
                        // lane 2 q + h takes transition h of the q-th distinct value that needs one
                        int my_id = -1, my_src = 0, q = 0;
                        while (rem && q < 32) {
                            const int leader = __ffsll((long long)rem) - 1;
                            const int idl = __shfl(cidl, leader);
                            const int kk0 = __shfl(k0, leader), kk1 = __shfl(k1, leader);
                            rem &= ~__ballot(cidl == idl);      // (leaves that share the covariance)
                            if (lane == 2 * q && kk0 < 0) { my_id = idl; my_src = leader; }
                            if (lane == 2 * q + 1 && kk1 < 0) { my_id = idl; my_src = leader; }
                            ++q;
                        }
                        if (my_id >= 0) chain_resolve(a, my_id, lane & 1, lg[my_src].pd);
                    }
                }
#endif
                FW_STAMP(7);
                run += ctot;
                c0 += n_eff;
                FW_SYNC();      // the chunk tables are re-used
            }
        }
        if (redo) { counted = true; pass = -1; run = 0; total = 0; ndead = 0; continue; }
        if (pass == 1) {      // all children are out: the target's entries of the child tables, its edges of the clustering graph
            const auto& a = *ap0;
            int e0 = 0, ne = 0;
            for (int w = lane; w < AW; w += 64) ne += __popcll(tb[w]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ne += __shfl_xor(ne, o);
            const int seg = t & (EDGE_SEGS - 1);
            if (lane == 0) {
                e0 = atomicAdd(&a.edge_count[seg], ne);
                if (e0 + ne > a.edge_cap) a.status->overflow = 1;
                atomicAdd(&a.status->n_children, run);
                if (ndead) atomicAdd(&a.status->n_dead, ndead);
                a.tchild[pos] = base;
                a.tcend[pos] = base + run;
            }
            int eb = __shfl(e0, 0);
            for (int w0 = 0; w0 < AW; w0 += 64) {      // edge list: (target << 16 | node) for every set bit
                const int w = w0 + lane;
                unsigned long long bits = (w < AW) ? tb[w] : 0ull;
                const int pc = __popcll(bits);
                int in2 = pc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(in2, o);
                    if (lane >= o) in2 += v;
                }
                int my = eb + in2 - pc;
                while (bits) {
                    const int bpos = __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    if (my < a.edge_cap) a.edges[(size_t)seg * a.edge_cap + my] = ((unsigned)pos << 16) | (unsigned)(w * 64 + bpos);
                    ++my;
                }
                eb += __shfl(in2, 63);
            }
        }
    }
}

// The PREVIOUS scan's report rides to the host in this launch (drop-in API path): FG_PUB_WGS extra workgroups copy it from its device
// block into pinned, device-mapped host memory (16-byte posted PCIe writes, ~70 KB at the headline size) while the others grow the
// tree -- in the kernel that completed the report the copy sat on the critical path of every scan (~10 us).  The host waits for an
// event recorded behind this launch.
constexpr int FG_PUB_WGS = 8;
static_assert(PUB_DONE_WORDS == FG_PUB_WGS + 1, "one word per pushing workgroup of fgrow_adm_kernel");
// every thread has stored its share of the report into pinned host memory: release it to the host, then one thread posts the workgroup's word
__device__ __forceinline__ void pub_done(const PublishArgs& pub, int w) {
    if (!pub.done) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pub.done + w, pub.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void publish_part(const PublishArgs& p, int w, int n_wgs = FG_PUB_WGS) {
    const ReportHeader* h = reinterpret_cast<const ReportHeader*>(p.src);
    const int n_births = h->n_births, nT = h->n_targets;
    const uint4* s4 = reinterpret_cast<const uint4*>(p.src);
    uint4* d4 = reinterpret_cast<uint4*>(p.dst);
    const int head = (p.birth_off + n_births * (int)sizeof(mht_birth_report) + 15) / 16;      // header + mask + births present
    const int r0 = p.rec_off / 16, rn = (nT * (int)sizeof(mht_target_report) + 15) / 16;
    const int i0 = w * FG_THREADS + threadIdx.x, st = n_wgs * FG_THREADS;
    for (int i = i0; i < head; i += st) d4[i] = s4[i];
    for (int i = i0; i < rn; i += st) d4[r0 + i] = s4[r0 + i];
}

template <int PQ, int CAP, typename CARGS, int AIS = 0, bool LEAN = false, int CT = 0>
__device__ __forceinline__ void fgrow_body(KArgs ap, const CARGS& cm, const FDyn& d, unsigned char* smem, const int bid0 = (int)blockIdx.x) {
    int bid = bid0;
    // stage stamps of this scan (DevStatus::t): the grow stage starts here.  Taken by a workgroup that is not at the edge of its
    // register budget -- the commit workgroup (first of the launch) or, without one, the first chain workgroup (the whole launch is
    // co-resident: it starts within a microsecond of the first target workgroup)
    auto stamp = [&]() {
        if (threadIdx.x == 0) { DevStatus* st = ap->status; st->t[0] = wall_clock64(); st->t[2] = 0; st->t[3] = 0; st->t[4] = 0; }
        // (clusters from the union-find: what the cluster kernel reset for the scan's ILP launch)
        if (d.uf_epoch && ap->uf_team_state && threadIdx.x < TEAM_MAX) { ap->uf_team_state[threadIdx.x].gub = ~0ull; ap->uf_team_state[threadIdx.x].done = 0; }
    };
    if (d.fused) {           // deferred commit of the previous scan: workgroup 0 runs it
        if (bid == 0) { stamp(); commit_body<FG_THREADS>(cm, CommitDyn{d.c_scan, d.c_M, d.c_W, d.ovl ? d.c_wait : 0ull, 0}, reinterpret_cast<int*>(smem)); return; }
        bid -= 1;
    }
    if (bid >= d.n_main) { if (!d.fused && bid == d.n_main) stamp(); chain_part<!LEAN, AIS>(*ap, d, bid - d.n_main); return; }
    if (CAP == 0) {          // wavefront-per-target variant: four targets per workgroup, each wavefront on its own LDS slice
        const int wave = threadIdx.x >> 6;
        const int t = bid * (FG_THREADS / 64) + wave;
        if (!d.fused && bid == 0) stamp();      // (no commit and no chain workgroup in this launch)
        if (t >= d.n_tgt) return;
        target_wave<PQ>(ap, d, t, smem + (size_t)wave * fw_layout(PQ * 4, ap->AW, d.W * 64).total);
    } else {
        if (CT && !d.fused && bid == 0) stamp();      // (no commit and no chain workgroup in a constant-turn launch)
        target_part<PQ, (CAP == 0 ? FG_CAP : CAP), AIS, LEAN, CT>(ap, d, bid, smem, bid, 0);
        if (d.stamp_end && threadIdx.x == 0) atomicMax(&ap->status->t[5], (unsigned long long)wall_clock64());
    }
}

template <int PQ, int CAP, typename CARGS>
__device__ __forceinline__ void fgrow_body_lean(KArgs ap, const CARGS& cm, const FDyn& d, unsigned char* smem) { fgrow_body<PQ, CAP, CARGS, 0, true>(ap, cm, d, smem); }

}  // namespace mht
