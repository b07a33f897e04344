// grow_kernel: the STATELESS gate / update / score seam (mht_gate_scan) -- the L x M fan-out of an explicit list of leaf
// hypotheses against every measurement of a scan and the creation of the child hypotheses, ONE launch per call
// (reference: pymht/tracker.py:804-859 + pymht/utils/kalman.py; children: pyTarget.py:227-258, :319-328).
// The forest does NOT come through here: its grow stage is fgrow_kernel (mht_fgrow.hip, one workgroup per target, gains one
// scan ahead).  This kernel serves callers that hold their own node arrays, and is the piece the golden Kalman vectors pin.
//
// Workgroup = 8 wavefronts = ONE tile of 32 consecutive leaves (tile = blockIdx while the grid is co-resident, ticket-
// numbered beyond that -- see the look-back below).
//   phase 1  lanes 0..31: one leaf per lane, SoA loads (coalesced), predict + precalc in registers (4x4 / 2x2
//            matrices: no MFMA), P_bar / P_hat written to the covariance table of the new layer, everything the
//            children need (x_bar, K, S^-1, z_hat, score constant) parked in LDS.
//   phase 2  the scan (staged in LDS once per workgroup) is cut down to the measurements inside one of <= 4 bounding boxes
//            (one per run of neighbouring leaves); thread = (leaf, candidate): the leaf's own conservative float32
//            box, the exact reference-order NIS only on pairs that pass; hits set bits in the leaf's hit mask (LDS
//            atomicOr).  No (L,M) tensor is ever materialised (the reference builds a 40 MB z_tilde + a 20 MB NIS).
//   phase 3  children of a leaf = 1 (missed detection) + hits.  The dense child index needs the number of
//            children of ALL earlier leaves: every tile publishes its count, every 64th tile a group sum, as
//            {epoch, value} words (one 64-bit agent-scope atomic each: the data is the flag, no fences); a tile's
//            base = group sums of earlier groups + counts of the earlier tiles of its own group: two L2 round trips
//            (<= 512 tiles: all earlier counts directly, one word per thread: one round trip).
//   phase 4  one thread per child: k-th set bit of the hit mask -> measurement, x_hat = x_bar + K z_tilde,
//            NLLR, cumulative score.  Consecutive threads write consecutive children: all SoA stores are coalesced.
// The bound is HBM traffic + launch/dependency latency; the kernel moves
// 200 B/leaf + 48 B/gated pair + 8 B/measurement of algorithmic data.
#include "mht_kernels.h"
#include <stdlib.h>

namespace mht {

struct LeafLds {          // per-leaf results of phase 1/2 parked in LDS for phase 4
    double xbar[4];
    double zhat[2];
    double cn, pd;
    float K[8];
    float sinv[4];
    float lnc, bx, by, zhx, zhy;
    int src, cnt, base;
    unsigned char flags, f32state, valid;
};

__device__ __forceinline__ void box_from_S(const float* S, double eta2, float zhx, float zhy, float& bx, float& by) {
    // NIS <= eta2  =>  |dz_x| <= sqrt(eta2*S00), |dz_y| <= sqrt(eta2*S11); widen for float32 rounding of the
    // pre-filter subtraction (coordinates up to ~1e6 m) -- the exact test decides, this only prunes.
    float rx = sqrtf((float)eta2 * fabsf(S[0])), ry = sqrtf((float)eta2 * fabsf(S[3]));
    bx = rx * 1.001f + 1e-6f * (fabsf(zhx) + rx) + 1e-3f;
    by = ry * 1.001f + 1e-6f * (fabsf(zhy) + ry) + 1e-3f;
}

template <typename TS>
__device__ __forceinline__ void phase1_leaf(const GateArgs& a, int i, const double* xd, const float* P, LeafLds& g) {
    TS xs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[k] = (TS)xd[k];
    Predicted<TS> p;
    predict_precalc<TS>(a.model, xs, P, p, a.L == 1);      // (one leaf in the call: NumPy's gemv order, mht_math.h::gemv_row)
#pragma unroll
    for (int k = 0; k < 4; ++k) g.xbar[k] = (double)p.x_bar[k];
    g.zhat[0] = (double)p.z_hat[0];
    g.zhat[1] = (double)p.z_hat[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) g.K[e] = p.K[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) g.sinv[e] = p.S_inv[e];
    g.lnc = nllr_const(p.S, a.model.lambda_ex, g.pd);
    g.zhx = (float)g.zhat[0];
    g.zhy = (float)g.zhat[1];
    box_from_S(p.S, a.model.eta2, g.zhx, g.zhy, g.bx, g.by);
    if (2 * i + 1 < a.capc_out) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            a.oP[(size_t)e * a.capc_out + 2 * i] = p.P_bar[e];
            a.oP[(size_t)e * a.capc_out + 2 * i + 1] = p.P_hat[e];
        }
    } else {
        a.status->overflow = 1;
    }
}

// monotone map float -> int (for LDS atomicMin/atomicMax on float keys)
__device__ __forceinline__ int sortable(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : (i ^ 0x7fffffff);
}

// tile state word: [63:32] epoch (the scan number, all 32 bits: a 24-bit tag would repeat after 16.7 M scans -- a quarter of an
// hour at replay speed -- and a stale word of a tile index that was last used that long ago would read as ready), [31:0] value
__device__ __forceinline__ unsigned long long pack_state(unsigned epoch, unsigned flag, unsigned value) {
    (void)flag;
    return ((unsigned long long)epoch << 32) | value;
}

template <typename TS>
__device__ __forceinline__ void emit_child(const GateArgs& a, const LeafLds& g, int i, int c, int k,
                                           const unsigned long long* hw, const float* zx, const float* zy) {
    const size_t cap = a.cap_out;
    const uint8_t fl = g.flags;
    int meas = 0, covcol = 2 * i, j = -1;
    if (k > 0) {             // (k-1)-th gated measurement in ascending index (pyTarget.py:242-254)
        int need = k - 1, w = 0;
        unsigned long long bits = hw[0];
        while (true) {
            const int pc = __popcll(bits);
            if (need < pc) break;
            need -= pc;
            bits = hw[++w];
        }
        for (int q = 0; q < need; ++q) bits &= bits - 1;
        j = w * 64 + __ffsll((long long)bits) - 1;
        meas = j + 1;
        covcol = 2 * i + 1;
    }
    double cnl, inc;
    uint8_t cfl = (uint8_t)(fl & F_STATE_F32);
    if (k == 0) {            // missed-detection child (pyTarget.py:319-328)
#pragma unroll
        for (int q = 0; q < 4; ++q) a.ox[(size_t)q * cap + c] = g.xbar[q];
        inc = (g.pd == a.default_pd) ? a.default_miss_nllr : -log(1.0 - g.pd);
        cnl = g.cn + inc;
    } else {
        const float2 m = make_float2(zx[j], zy[j]);
        TS zh[2] = {(TS)g.zhat[0], (TS)g.zhat[1]}, xb[4] = {(TS)g.xbar[0], (TS)g.xbar[1], (TS)g.xbar[2], (TS)g.xbar[3]};
        TS zt[2], nis, xh[4];
        gate_pair<TS>(zh, g.sinv, m.x, m.y, (TS)a.model.eta2, zt, nis);
        update_state<TS>(xb, g.K, zt, xh, g.cnt == 1);      // (one gated measurement: matrix x column = gemv)
#pragma unroll
        for (int q = 0; q < 4; ++q) a.ox[(size_t)q * cap + c] = (double)xh[q];
        const TS tinc = (TS)0.5 * nis + (TS)g.lnc;           // kalman.py:19
        inc = (double)tinc;
        if (sizeof(TS) == 4 && (fl & F_SCORE_F32)) {          // float32 + float32 stays float32 (NumPy scalar rules)
            cnl = (double)((float)g.cn + (float)tinc);
            cfl |= F_SCORE_F32;
        } else {
            cnl = g.cn + inc;
        }
    }
    a.ocnllr[c] = cnl;
    a.opd[c] = g.pd;
    a.oparent[c] = g.src;
    a.omeas[c] = meas;
    a.ocov[c] = covcol;
    a.oflags[c] = cfl;
    if (a.nllr) a.nllr[c] = inc;
}

// per-phase wall-clock stamps of every tile (tools/grow_profile.py): compiled in only with -DMHT_GROW_STAMPS, they cost
// SGPRs (-> spills) and an LDS/SMEM drain per stamp
#ifdef MHT_GROW_STAMPS
#define GROW_STAMP_DECL unsigned long long ts[7]
#define GROW_STAMP(k) ts[k] = wall_clock64()
#else
#define GROW_STAMP_DECL
#define GROW_STAMP(k)
#endif

constexpr int SPIN_LIMIT = 1 << 23;     // look-back watchdog: x ~0.1 us per poll

__global__ __launch_bounds__(GATE_THREADS, 4) void grow_kernel(const GateArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = a.M, W = a.W;
    const int Mpad = W * 64;
    float* zx = reinterpret_cast<float*>(smem);
    float* zy = zx + Mpad;
    LeafLds* lg = reinterpret_cast<LeafLds*>(zy + Mpad);
    unsigned long long* hw = reinterpret_cast<unsigned long long*>(lg + GATE_TILE);    // [GATE_TILE][W]
    unsigned short* cand = reinterpret_cast<unsigned short*>(hw + (size_t)GATE_TILE * W);   // [Mpad] phase-2 candidates
    __shared__ int s_base, s_total, s_stall, s_pref[GATE_TILE + 1];
    __shared__ int s_box[4][4], s_ncand;      // up to 4 target segments per tile: {min x, max x, min y, max y} as sortable ints

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x;
    // First round trip: the first slice of the scan
    const float2* z2 = reinterpret_cast<const float2*>(a.z);
    const float2 z_first = (tid < M) ? z2[tid] : make_float2(3.0e38f, 3.0e38f);
    const int L = a.L;
    const int ntiles = (L + GATE_TILE - 1) / GATE_TILE;

    // One tile per workgroup.  Static mapping (tile = blockIdx) whenever the whole grid is co-resident: a workgroup spinning
    // in the look-back then only waits for workgroups that are running, and nothing serialises on a shared word.  A grid
    // larger than the machine takes tile numbers from a ticket counter instead: whoever holds tile t is running and
    // every tile < t was handed to a workgroup that started earlier, so the look-back cannot wait for an undispatched
    // workgroup whatever the dispatch order.  (Looping over tiles inside a workgroup was measurably worse: everything
    // loop invariant got hoisted, the kernel hit the 128-register budget and spilled 32 B of scratch per thread at
    // entry -- ~7 MB of HBM writes per launch; without the loop it needs 80 registers and no scratch.)  Spins are
    // bounded (SPIN_LIMIT, ~1 s): should another resident grid starve this one, the scan is voided with a loud error
    // instead of hanging the device.
    if (ntiles == 0 && bid == 0 && tid == 0) {      // no leaves at all
        a.child_ptr[0] = 0;
        a.status->n_children = 0;
    }
    int tile = bid;
    if (tile < ntiles && ntiles > a.max_resident) {      // (workgroups beyond the last tile leave without a ticket)
        if (tid == 0) s_base = atomicAdd(a.ticket, 1);
        __syncthreads();
        tile = s_base;
        __syncthreads();
    }
    if (tile < ntiles) {
        GROW_STAMP_DECL;
        GROW_STAMP(0);
        for (int j = tid; j < Mpad; j += GATE_THREADS) {
            const float2 v = (j == tid) ? z_first : ((j < M) ? z2[j] : make_float2(3.0e38f, 3.0e38f));
            zx[j] = v.x;
            zy[j] = v.y;
        }
        for (int w = tid; w < GATE_TILE * W; w += GATE_THREADS) hw[w] = 0ull;      // hit masks of phase 2
        if (tid < 16) s_box[tid >> 2][tid & 3] = (tid & 1) ? (int)0x80000000 : 0x7fffffff;
        if (tid == 16) s_ncand = 0;
        __syncthreads();
        GROW_STAMP(1);
        // ---- phase 1: predict + precalc, one leaf per lane -----------------------------------------------------
        if (tid < GATE_TILE) {
            const int i = tile * GATE_TILE + tid;
            LeafLds& g = lg[tid];
            g.valid = i < L;
            g.cnt = 0;
            if (g.valid) {
                const int src = a.leaf_src ? a.leaf_src[i] : i;
                g.src = src;
                // Two load batches only -- (A) everything addressed by the leaf, (B) the covariance column that A's `cov`
                // names.  All of A is issued before anything is consumed.
                const uint8_t fl = a.flags[src];
                const double cn = a.cnllr[src], pd = a.pd[src];
                const int covc = a.cov[src];
                double xd[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) xd[k] = a.x[(size_t)k * a.cap_in + src];
                float P[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) P[e] = a.P[(size_t)e * a.capc_in + covc];
                g.flags = fl;
                g.f32state = (fl & F_STATE_F32) ? 1 : 0;
                g.cn = cn;
                g.pd = pd;
                if (!(a.ablate & 8)) {
                    if (g.f32state) phase1_leaf<float>(a, i, xd, P, g);
                    else phase1_leaf<double>(a, i, xd, P, g);
                }
            }
            // bounding boxes of the gates, one per run of neighbouring leaves (callers list the leaves of a target together:
            // they sit within a few hundred metres of each other, targets do not): the scan is first cut down to the
            // measurements inside one of the <= 4 boxes, only those are tested per leaf
            const int tg = g.valid ? -1 : -2, tp = __shfl_up(tg, 1);
            const float pzx = __shfl_up(g.zhx, 1), pzy = __shfl_up(g.zhy, 1), pbx = __shfl_up(g.bx, 1), pby = __shfl_up(g.by, 1);
            // a run ends where the predicted measurement jumps by more than a few gate widths
            const bool jump = tg == -1 && (fabsf(g.zhx - pzx) > 8.0f * (g.bx + pbx) || fabsf(g.zhy - pzy) > 8.0f * (g.by + pby));
            const bool head = g.valid && (tid == 0 || tg != tp || jump);
            const unsigned long long hb = __ballot(head);
            int seg = __popcll(hb & ((2ull << tid) - 1ull)) - 1;
            if (seg > 3) seg = 3;
            if (g.valid && seg >= 0) {
                float lox = g.zhx - g.bx, hix = g.zhx + g.bx, loy = g.zhy - g.by, hiy = g.zhy + g.by;
                lox -= fabsf(lox) * 2.4e-7f + 1e-30f; hix += fabsf(hix) * 2.4e-7f + 1e-30f;      // outward: a superset of the leaf's own box
                loy -= fabsf(loy) * 2.4e-7f + 1e-30f; hiy += fabsf(hiy) * 2.4e-7f + 1e-30f;
                atomicMin(&s_box[seg][0], sortable(lox)); atomicMax(&s_box[seg][1], sortable(hix));
                atomicMin(&s_box[seg][2], sortable(loy)); atomicMax(&s_box[seg][3], sortable(hiy));
            }
        }
        __syncthreads();
        GROW_STAMP(2);
        // ---- phase 2: (a) every thread tests measurements against the segment boxes and appends the survivors to a candidate
        //      list (wave ballot + one LDS atomic per wavefront; the order is irrelevant, hits are recorded by measurement
        //      index); (b) thread = (leaf, candidate): the leaf's own conservative float32 box, then the exact reference-order
        //      NIS, hits set bits in the leaf's hit mask (LDS atomicOr; ~1 hit per leaf).  No (L, M) sweep per leaf.
        if (!(a.ablate & 4)) {
            int bxs[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) bxs[q][e] = s_box[q][e];
            for (int j0 = 0; j0 < Mpad; j0 += GATE_THREADS) {
                const int j = j0 + tid;
                bool in = false;
                if (j < Mpad) {
                    const int kx = sortable(zx[j]), ky = sortable(zy[j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) in |= (kx >= bxs[q][0]) && (kx <= bxs[q][1]) && (ky >= bxs[q][2]) && (ky <= bxs[q][3]);
                }
                const unsigned long long bal = __ballot(in);
                int wbase = 0;
                if (lane == 0 && bal) wbase = atomicAdd(&s_ncand, __popcll(bal));
                wbase = __shfl(wbase, 0);
                if (in) cand[wbase + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)j;
            }
            __syncthreads();
            const int nc = s_ncand;
            for (int w = tid; w < GATE_TILE * nc; w += GATE_THREADS) {
                const int l = w & (GATE_TILE - 1), j = cand[w / GATE_TILE];
                const LeafLds& g = lg[l];
                if (!g.valid) continue;
                const float mx = zx[j], my = zy[j];
                if ((fabsf(mx - g.zhx) <= g.bx) && (fabsf(my - g.zhy) <= g.by)) {
                    bool hit;
                    if (g.f32state) {
                        float zh[2] = {(float)g.zhat[0], (float)g.zhat[1]}, zt[2], nis;
                        hit = gate_pair<float>(zh, g.sinv, mx, my, (float)a.model.eta2, zt, nis);
                    } else {
                        double zh[2] = {g.zhat[0], g.zhat[1]}, zt[2], nis;
                        hit = gate_pair<double>(zh, g.sinv, mx, my, a.model.eta2, zt, nis);
                    }
                    if (hit) atomicOr(&hw[(size_t)l * W + (j >> 6)], 1ull << (j & 63));
                }
            }
        }
        __syncthreads();
        if (a.used)                                                           // stateless seam: used-measurement mask
            for (int idx = tid; idx < GATE_TILE * W; idx += GATE_THREADS) {
                const unsigned long long word = hw[idx];
                if (word) atomicOr(&a.used[idx % W], word);
            }
        GROW_STAMP(3);
        // ---- phase 3: child offsets: in-tile prefix + two-level prefix across tiles -------------------------------------
        // Every tile publishes its child count A[tile]; the last tile of each group of 64 also publishes the group sum
        // S[group].  A tile's base = sum of S over earlier groups + sum of A over earlier tiles of its own group: two
        // L2 round trips however many tiles there are (a chained look-back would serialise ~ntiles/64 hops here,
        // because all tiles arrive at the same time).  Words carry {epoch, value} in one 64-bit agent-scope atomic:
        // the data is the flag, stale epochs read as "not yet".
        if (wave == 0) {
            int hits = 0;                                                     // hits of this lane's leaf (no separate counting phase)
            if (lane < GATE_TILE)
                for (int w = 0; w < W; ++w) hits += __popcll(hw[(size_t)lane * W + w]);
            if (lane < GATE_TILE) lg[lane].cnt = hits;
            int mine = (lane < GATE_TILE && lg[lane].valid) ? 1 + hits : 0;
            int incl = mine;
#pragma unroll
            for (int o = 1; o < GATE_TILE; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            if (lane < GATE_TILE) s_pref[lane] = incl - mine;
            const int total = __shfl(incl, GATE_TILE - 1);
            if (lane == 0) {
                s_pref[GATE_TILE] = total;
                s_total = total;
                s_base = 0;
                s_stall = 0;
                __hip_atomic_store(&a.tile_state[tile], pack_state(a.epoch, 1u, (unsigned)total),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((tile & 63) == 63 && ntiles > GATE_THREADS) {          // group leader: sum of the 64 tiles of the group (two-hop mode only)
                int v = total;
                if (lane < 63) {
                    const int tq = tile - 63 + lane;
                    unsigned long long st;
                    int spins = 0;
                    do {
                        st = __hip_atomic_load(&a.tile_state[tq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(st >> 32) != a.epoch) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_LIMIT) { s_stall = 1; break; }
                        }
                    } while ((unsigned)(st >> 32) != a.epoch);
                    v = (int)(st & 0xffffffffu);
                } else if (lane == 63) {
                    v = total;
                }
                if (lane > 63) v = 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0)
                    __hip_atomic_store(&a.group_state[tile >> 6], pack_state(a.epoch, 1u, (unsigned)v),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        {
            // few tiles (the co-resident case): read every earlier tile's count directly -- one hop instead of two (group
            // sums are published only after their leader has seen its 63 tiles); many tiles: group sums + own group
            const bool direct = ntiles <= GATE_THREADS;
            const int grp = tile >> 6, r = tile & 63;
            const int nread = (a.ablate & 2) ? 0 : (direct ? tile : grp + r);
            int acc = 0;
            for (int q = tid; q < nread; q += GATE_THREADS) {
                const unsigned long long* w = direct ? &a.tile_state[q] : ((q < grp) ? &a.group_state[q] : &a.tile_state[grp * 64 + (q - grp)]);
                unsigned long long st;
                int spins = 0;
                do {
                    st = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(st >> 32) != a.epoch) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT) { s_stall = 1; break; }
                    }
                } while ((unsigned)(st >> 32) != a.epoch);
                acc += (int)(st & 0xffffffffu);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0 && acc) atomicAdd(&s_base, acc);
        }
        __syncthreads();
        if (s_stall) {      // watchdog: a tile this one depends on never published (its workgroup is not resident): void scan
            if (tid == 0) a.status->overflow = 2;
            return;
        }
        GROW_STAMP(4);
        const int base = s_base, total = s_total;
        if (tid < GATE_TILE && lg[tid].valid) {
            const int i = tile * GATE_TILE + tid;
            const int cb = base + s_pref[tid];
            lg[tid].base = cb;
            a.child_ptr[i] = cb;
            if (i == L - 1) {
                const int all = base + total;
                a.child_ptr[L] = all;
                a.status->n_children = all;
                if (all > a.cap_out) a.status->overflow = 1;
            }
        }
        __syncthreads();
        GROW_STAMP(5);
        // ---- phase 4: one thread per child -----------------------------------------------------------------------------
        for (int r = tid; r < ((a.ablate & 1) ? 0 : ((total + 63) & ~63)); r += GATE_THREADS) {
            if (r < total) {
                int l = 0;
#pragma unroll
                for (int q = 1; q < GATE_TILE; ++q) l += (s_pref[q] <= r) ? 1 : 0;     // leaf of child r
                const LeafLds& g = lg[l];
                const int k = r - s_pref[l];
                const int c = base + r;
                const int i = tile * GATE_TILE + l;
                if (c < a.cap_out) {
                    if (g.f32state) emit_child<float>(a, g, i, c, k, hw + (size_t)l * W, zx, zy);
                    else emit_child<double>(a, g, i, c, k, hw + (size_t)l * W, zx, zy);
                }
            }
        }
#ifdef MHT_GROW_STAMPS
        if (a.dbg) {
            __syncthreads();
            if (tid == 0 && tile < 4000) {       // per-tile slots: no contention, stamps relative to the first tile's start are
                ts[6] = wall_clock64();          // not available (no global clock origin), so absolute ticks are stored
                for (int q = 0; q < 7; ++q) a.dbg[32 + (size_t)tile * 8 + q] = ts[q];
                a.dbg[32 + (size_t)tile * 8 + 7] = (unsigned long long)blockIdx.x;
            }
        }
#endif
    }
}

static inline size_t grow_lds_bytes(int W) {
    return (size_t)2 * W * 64 * sizeof(float) + GATE_TILE * sizeof(LeafLds) + (size_t)GATE_TILE * W * 8 + (size_t)W * 64 * 2 + 16;      // + candidate list
}

int launch_gate(mht_ctx* ctx, GateArgs& a, int grid_leaves_hint) {
    const int L = grid_leaves_hint > a.L ? grid_leaves_hint : a.L, W = a.W;
    if (!a.status) a.status = ctx->status;
    int ntiles = (L + GATE_TILE - 1) / GATE_TILE;
    if (ntiles < 1) ntiles = 1;
    // tile states: epoch-tagged.  They live in the ctx scratch, which mht_solve_blp also uses and which may just have been
    // reallocated: whatever is in there could pass for a tile state of this epoch, so it is cleared on every call.
    if (!a.tile_state) {
        const size_t bytes = ((size_t)ntiles + ntiles / 64 + 16) * 8;
        int rc = ctx->hitmask.ensure(bytes);
        if (rc) return rc;
        MHT_HIP_CHECK(hipMemsetAsync(ctx->hitmask.ptr, 0, bytes, ctx->stream));
        a.tile_state = static_cast<unsigned long long*>(ctx->hitmask.ptr);
        a.group_state = a.tile_state + ntiles + 4;
    }
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("MHT_GROW_ABLATE"); ablate = e ? atoi(e) : 0; }
    a.ablate = ablate;
    const size_t lds = grow_lds_bytes(W);
    // the grid must be co-resident (see the tile prefix): <= 4 workgroups per CU by registers, fewer if LDS says so
    int per_cu = (int)((160 * 1024) / (lds + 256));
    const int by_regs = 16 / (GATE_THREADS / 64);     // 4 wavefronts per SIMD at 128 registers = 16 per CU
    if (per_cu > by_regs) per_cu = by_regs;
    if (per_cu < 1) per_cu = 1;
    const int max_blocks = ctx->n_cu * per_cu;
    if (getenv("MHT_GROW_DEBUG")) {
        static bool printed = false;
        if (!printed) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, grow_kernel, GATE_THREADS, lds);
            hipFuncAttributes fa;
            (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(grow_kernel));
            fprintf(stderr, "[grow] occupancy API: %d workgroups/CU at %zu B dynamic LDS; static LDS %zu B, regs %d, local %zu B\n", nb, lds,
                    (size_t)fa.sharedSizeBytes, fa.numRegs, (size_t)fa.localSizeBytes);
            printed = true;
        }
    }
    const int blocks = ntiles;
    a.max_resident = max_blocks;                      // more tiles than that: dynamic tile numbers (see grow_kernel)
    if (ntiles > max_blocks && !a.ticket)             // ticket word behind the tile states (cleared above)
        a.ticket = reinterpret_cast<int32_t*>(a.group_state + ntiles / 64 + 4);
    size_t& attr_bytes = ctx->lds_attr_gate;
    if (lds > 48 * 1024 && lds > attr_bytes) {
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(grow_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    hipLaunchKernelGGL(grow_kernel, dim3(blocks), dim3(GATE_THREADS), lds, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

void fill_model_only(Model& o, const mht_model* m) {
    for (int i = 0; i < NP; ++i) { o.A[i] = m->A[i]; o.Q[i] = m->Q[i]; }
    for (int i = 0; i < NK; ++i) o.C[i] = m->C[i];
    for (int i = 0; i < 4; ++i) o.R[i] = m->R[i];
    o.eta2 = m->eta2;
    o.lambda_ex = m->lambda_ex;
}

void fill_model(GateArgs& a, const mht_model* m) {
    fill_model_only(a.model, m);
    a.default_pd = m->default_pd;
    a.default_miss_nllr = m->default_miss_nllr;
}

}  // namespace mht

using namespace mht;

extern "C" int mht_gate_scan(mht_ctx* ctx, const mht_model* model, const mht_nodes* in, const int32_t* leaf_src,
                             int32_t L, const float* z, int32_t M, const mht_nodes* out, int32_t* child_ptr,
                             double* nllr, uint64_t* used, int32_t* n_children) {
    MHT_REQUIRE(NX == 4, "mht_gate_scan: the tile kernel of this seam is 4-state; this is the %d-state build of the library (use mht_gate_scan_x)", NX);
    MHT_REQUIRE(ctx && model && in && out && child_ptr, "mht_gate_scan: null argument");
    MHT_REQUIRE(L >= 0 && M >= 0 && M <= MAX_MEAS, "mht_gate_scan: need 0 <= M <= %d, L >= 0 (M=%d L=%d)", MAX_MEAS, M, L);
    MHT_REQUIRE(z || M == 0, "mht_gate_scan: z is null");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    GateArgs a = {};
    fill_model(a, model);
    a.x = in->x; a.cnllr = in->cnllr; a.pd = in->pd; a.cov = in->cov; a.flags = in->flags; a.P = in->P;
    a.cap_in = in->cap; a.capc_in = in->cap_cov;
    a.leaf_src = leaf_src; a.L = L;
    a.z = z; a.M = M; a.W = (M + 63) / 64;
    a.ox = out->x; a.ocnllr = out->cnllr; a.opd = out->pd; a.oparent = out->parent; a.omeas = out->meas;
    a.ocov = out->cov; a.oflags = out->flags; a.oP = out->P; a.cap_out = out->cap; a.capc_out = out->cap_cov;
    a.child_ptr = child_ptr; a.nllr = nllr; a.used = reinterpret_cast<unsigned long long*>(used);
    a.epoch = ++ctx->gate_epoch;      // stateless use: a fresh look-back epoch per call
    MHT_HIP_CHECK(hipMemsetAsync(ctx->status, 0, sizeof(DevStatus), ctx->stream));
    int rc = launch_gate(ctx, a, L);
    if (rc) return rc;
    if (n_children) {
        DevStatus st;
        MHT_HIP_CHECK(hipMemcpyAsync(&st, ctx->status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *n_children = st.n_children;
        if (st.overflow == 2) {
            set_error("mht_gate_scan: grow_kernel stalled waiting for a tile that was never dispatched (GPU shared with another resident grid?)");
            return MHT_E_HIP;
        }
        if (st.overflow) {
            set_error("mht_gate_scan: output layer too small (need cap >= %d, cap_cov >= %d)", st.n_children, 2 * L);
            return MHT_E_CAPACITY;
        }
    }
    return MHT_OK;
}
