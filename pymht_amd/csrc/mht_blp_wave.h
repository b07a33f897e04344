// The 0-1 ILP of a SMALL cluster on ONE wavefront (included by mht_blp.hip behind the forest epilogue's helpers).
//
// Reference: Tracker._solveBLP_OR_TOOLS (pymht/tracker.py:1155-1217) per cluster (:228-236).  The workgroup solver (solve_cluster /
// solve_core<LStore>) runs the dual coordinate ascent out of LDS with a block barrier between its phases: ~6 barriers per round, 0.6-0.8 us
// each whatever they compute (an LDS round trip is ~130 cycles) -- 11 us for a cluster that is certified at zero prices, +4.3 us for every
// further round, and the ILP launch of a scan ends with the one cluster in a hundred that needs two or three of them.  Nearly all clusters
// are tiny, though (headline stream: 2-5 targets, 130-400 columns): here ONE wavefront runs the whole cluster --
//   * the columns in a wave-private LDS block (28 bytes each: cost, reduced cost, the <= 8 rows as dense 8-bit ids, the ancestor entry the
//     survivors' sweep needs), prices / usage counters / marks of the <= 255 rows next to them, member tables in lanes 0 .. K-1;
//   * the phases are ordered by the wavefront's in-order LDS pipeline (fence + wave barrier), never by s_barrier; no cross-wavefront
//     reduction anywhere (DPP only).
// It runs the SAME coordinate rounds as solve_core<LStore> -- minimisers (lowest column among equals), usage, certificate, nomination,
// regrets, price step, slack step: the same arithmetic in the same order, so the prices and the certified selection are the workgroup
// solver's, bit for bit -- and gives up (returns false, nothing written to global memory) exactly where that solver would leave the
// coordinate rounds (pair enumeration, exact search, subgradient steps, branch and bound) or when the cluster does not fit; the caller
// then runs the workgroup solver from scratch.  A certified cluster is finished here: selection, termination test, prune decision, report
// rows, surviving leaf ranges, records for the next grow launch (finish_target / blp_publish), status CERTIFIED after `it` rounds.
#pragma once

namespace mht {

constexpr int WV_MAXK = 16;           // targets of a cluster the wavefront takes
constexpr int WV_NOROW = 255;         // dense row id of "no row" (its price stays 0, its counters are never touched)
constexpr int WV_MAXR = 255;          // rows
constexpr int WV_COLS_SOLO = 1024;    // columns in the one-sector launches (every cluster of the headline stream)
// LDS of one wavefront's solve: a forest with UW 64-bit words of measurement nodes, clusters of up to `cols` columns
__host__ __device__ constexpr size_t wave_solver_lds(int UW, int cols) {
    return (((size_t)UW * 12 + 15) & ~(size_t)15) + 256 * 8 + 256 * 4 + 256 * 4 + 40 * 4 + 2 * WV_MAXK * 8 + (size_t)cols * 28;
}

#define WV_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ double wv_rl64(double v, int src) {      // v of lane `src` (uniform)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ int wv_byte(unsigned lo, unsigned hi, int d) { return (int)(((d < 4 ? lo : hi) >> ((d & 3) * 8)) & 0xffu); }
// does any of the eight row ids equal m (m < 255)?
__device__ __forceinline__ bool wv_has_row(unsigned lo, unsigned hi, int m) {
    const unsigned mm = (unsigned)m * 0x01010101u;
    const unsigned x0 = lo ^ mm, x1 = hi ^ mm;
    return ((((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1)) & 0x80808080u) != 0u;
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int wv_dpp_min_i32(int v) {
    return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int wv_min_i32(int v) {      // minimum over the wavefront, valid in every lane
    v = wv_dpp_min_i32<0x111, 0xf>(v);
    v = wv_dpp_min_i32<0x112, 0xf>(v);
    v = wv_dpp_min_i32<0x114, 0xf>(v);
    v = wv_dpp_min_i32<0x118, 0xf>(v);
    v = wv_dpp_min_i32<0x142, 0xa>(v);
    v = wv_dpp_min_i32<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
struct alignas(16) WvCol { unsigned lo, hi; double rc; };      // a column: its rows (first: 8 x 16-bit node ids over all 16 bytes), its reduced cost

// lane k < K: target index `tk` of member k (ascending) and its record `pre` (load_target).  All 64 lanes of the wavefront call.
// cap: columns the LDS block `wl` (wave_solver_lds(UW, cap) bytes, 16-byte aligned) holds.
__device__ __forceinline__ bool solve_wave(const BlpArgs& a, const ClRef cr, const int tk, const TgtPre& pre, unsigned char* wl, const int cap, const int dbg_bx) {
    const int lane = threadIdx.x & 63, K = cr.K, c = cr.c;
    const int UW = (a.n_mnodes + 63) >> 6;
    const unsigned long long t_begin = wall_clock64();
    // ---- LDS of the wavefront ---------------------------------------------------------------------------------------------------------
    unsigned long long* uw = reinterpret_cast<unsigned long long*>(wl);
    int* wbase = reinterpret_cast<int*>(uw + UW);
    double* u = reinterpret_cast<double*>(wl + (((size_t)UW * 12 + 15) & ~(size_t)15));
    int* usage = reinterpret_cast<int*>(u + 256);
    int* mark = usage + 256;
    int* colbL = mark + 256;                       // [17] first column of member k, colbL[K] = columns
    int* lixL = colbL + 20;                        // [16] (slack step: by member)
    double* mnL = reinterpret_cast<double*>(colbL + 40);      // [16]
    double* brcL = mnL + WV_MAXK;                  // [16]
    WvCol* col = reinterpret_cast<WvCol*>(brcL + WV_MAXK);    // [cap]
    double* costL = reinterpret_cast<double*>(col + cap);     // [cap]
    int* ancL = reinterpret_cast<int*>(costL + cap);          // [cap] ancestor-table entry of the column's child at level j - 1 of its target
    // ---- column ranges of the members: lane k holds member k --------------------------------------------------------------------------
    const int n_mine = lane < K ? pre.ce - pre.cb : 0;
    int incl = n_mine;
#pragma unroll
    for (int o = 1; o < WV_MAXK; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const int my_cb = incl - n_mine, my_ce = incl;      // local column range of member `lane`
    const int nH = __builtin_amdgcn_readlane(incl, WV_MAXK - 1);      // (lanes >= K add nothing)
    if (nH > cap || __any(lane < K && n_mine < 1)) return false;
    if (lane < K) colbL[lane] = my_cb;
    if (lane == 0) colbL[K] = nH;
    for (int w = lane; w < UW; w += 64) uw[w] = 0ull;
    for (int r = lane; r < 256; r += 64) { u[r] = 0.0; usage[r] = 0; mark[r] = 0; }
    WV_SYNC();
    // ---- the columns: cost, path record and the survivors' ancestor entry of column h, four columns per lane in flight -----------------
    for (int h0 = 0; h0 < nH; h0 += 256) {
        int g[4], jj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = h0 + q * 64 + lane;
            int m = 0;
#pragma unroll 1
            for (int i = 1; i < K; ++i) m += (h >= colbL[i]) ? 1 : 0;
            g[q] = __shfl(pre.cb, m) + (h - __shfl(my_cb, m));
            jj[q] = __shfl(pre.j, m);
            if (h >= nH) { g[q] = __builtin_amdgcn_readfirstlane(pre.cb); jj[q] = 0; }      // (clamped: unconditional loads)
        }
        double cs[4];
        int4 p0[4], p1[4];
        int an[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cs[q] = a.cost[g[q]];
            const int4* rec = reinterpret_cast<const int4*>(a.path + (size_t)g[q] * 8);
            p0[q] = rec[0]; p1[q] = rec[1];
            an[q] = a.apath[(size_t)g[q] * 8 + (jj[q] > 0 ? jj[q] - 1 : 0)];      // (sweep_prefetch)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = h0 + q * 64 + lane;
            if (h < nH) {
                costL[h] = cs[q];
                ancL[h] = jj[q] > 0 ? an[q] : -1;
                // node ids as 16-bit halves for now (-1 -> 0xffff)
                *reinterpret_cast<uint4*>(&col[h]) = make_uint4(((unsigned)p0[q].x & 0xffffu) | ((unsigned)p0[q].y << 16), ((unsigned)p0[q].z & 0xffffu) | ((unsigned)p0[q].w << 16),
                                                                ((unsigned)p1[q].x & 0xffffu) | ((unsigned)p1[q].y << 16), ((unsigned)p1[q].z & 0xffffu) | ((unsigned)p1[q].w << 16));
                const int ev[8] = {p0[q].x, p0[q].y, p0[q].z, p0[q].w, p1[q].x, p1[q].y, p1[q].z, p1[q].w};
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    if (ev[d] >= 0) atomicOr(&uw[ev[d] >> 6], 1ull << (ev[d] & 63));
            }
        }
    }
    WV_SYNC();
    // dense row ids: exclusive prefix of the popcounts of the bitset
    int nR = 0;
    {
        int carry = 0;
        for (int base = 0; base < UW; base += 64) {
            const int w = base + lane;
            const int pc = (w < UW) ? __popcll(uw[w]) : 0;
            int in2 = pc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(in2, o);
                if (lane >= o) in2 += v;
            }
            if (w < UW) wbase[w] = carry + in2 - pc;
            carry += __builtin_amdgcn_readlane(in2, 63);
        }
        nR = carry;
    }
    if (nR > WV_MAXR) return false;
    WV_SYNC();
    for (int h = lane; h < nH; h += 64) {
        const uint4 v = *reinterpret_cast<const uint4*>(&col[h]);
        const unsigned e16[8] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16, v.z & 0xffffu, v.z >> 16, v.w & 0xffffu, v.w >> 16};
        unsigned o[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned e = e16[d] != 0xffffu ? e16[d] : 0u;
            const unsigned id = (unsigned)(wbase[e >> 6] + __popcll(uw[e >> 6] & ((1ull << (e & 63)) - 1ull)));
            o[d] = e16[d] != 0xffffu ? id : (unsigned)WV_NOROW;
        }
        col[h].lo = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        col[h].hi = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
    }
    WV_SYNC();
    const unsigned long long t_setup = wall_clock64();
    // ---- coordinate rounds (solve_core<LStore>: the rounds in front of the exact searches) ---------------------------------------------
    const int ca_rounds = (K == 2) ? CA_ROUNDS_PAIR : CA_ROUNDS;
    const int ca_end = a.max_iter < ca_rounds ? a.max_iter : ca_rounds;
    const int enum_after = (nH > ENUM_WIDE) ? ENUM_AFTER_WIDE : ENUM_AFTER;
    const int enum_at = (ca_end > 0 && K >= 3 && K <= ENUM_MAXK && !a.no_enum) ? (ca_end < enum_after ? ca_end : enum_after) : -1;
    int my_bh = -1, my_lix = -1;          // lane k: minimiser (local column) of member k, the row it nominates
    double my_brc = 0.0, my_mn = -1.0;    // ... its reduced cost, its regret
    unsigned my_rlo = 0xffffffffu, my_rhi = 0xffffffffu;      // ... its rows
    int iters = 0;
    unsigned long long stamp1 = 0, stamp3 = 0;
    for (int it = 0;; ++it) {
        if (it >= ca_end || it == enum_at) return false;      // (the workgroup solver leaves the coordinate rounds here)
        iters = it;
        // A: per member the minimiser of the reduced cost (lowest column among equals) and the usage of its rows; the reduced costs stay in LDS
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            const int cbk = __builtin_amdgcn_readlane(my_cb, k), cek = __builtin_amdgcn_readlane(my_ce, k);
            double bv = DINF;
            int bi = -1;
            for (int h = cbk + lane; h < cek; h += 128) {      // two columns in flight per lane
                const int h1 = h + 64, h1c = h1 < cek ? h1 : h;
                const unsigned lo0 = col[h].lo, hi0 = col[h].hi, lo1 = col[h1c].lo, hi1 = col[h1c].hi;
                const double c0 = costL[h], c1 = costL[h1c];
                const double a0 = u[lo0 & 0xff], a1 = u[(lo0 >> 8) & 0xff], a2 = u[(lo0 >> 16) & 0xff], a3 = u[lo0 >> 24];
                const double a4 = u[hi0 & 0xff], a5 = u[(hi0 >> 8) & 0xff], a6 = u[(hi0 >> 16) & 0xff], a7 = u[hi0 >> 24];
                const double b0 = u[lo1 & 0xff], b1 = u[(lo1 >> 8) & 0xff], b2 = u[(lo1 >> 16) & 0xff], b3 = u[lo1 >> 24];
                const double b4 = u[hi1 & 0xff], b5 = u[(hi1 >> 8) & 0xff], b6 = u[(hi1 >> 16) & 0xff], b7 = u[hi1 >> 24];
                const double rc0 = ((((((((c0 + a0) + a1) + a2) + a3) + a4) + a5) + a6) + a7);
                const double rc1 = ((((((((c1 + b0) + b1) + b2) + b3) + b4) + b5) + b6) + b7);
                col[h].rc = rc0;
                if (bi < 0 || rc0 < bv) { bv = rc0; bi = h; }
                if (h1 < cek) {
                    col[h1].rc = rc1;
                    if (rc1 < bv) { bv = rc1; bi = h1; }
                }
            }
            const double gmin = wave_min_value(bi >= 0 ? bv : DINF);
            const int gi = wv_min_i32((bi >= 0 && bv == gmin) ? bi : 0x7fffffff);      // (lowest column among equals)
            if (gi == 0x7fffffff) return false;      // (cannot happen: every member has a column)
            const unsigned lo = col[gi].lo, hi = col[gi].hi;      // (uniform address)
            if (lane == k) { my_bh = gi; my_brc = gmin; my_rlo = lo; my_rhi = hi; }
            if (lane < 8) {
                const int id = wv_byte(lo, hi, lane);
                if (id != WV_NOROW) atomicAdd(&usage[id], 1);
            }
        }
        WV_SYNC();
        if (it == 0) stamp1 = wall_clock64();
        // C: certificate flags; every member nominates the lowest conflicted row of its minimiser
        bool cf = false, sl = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lane + 64 * q;
            if (r < nR) {
                const int us = usage[r];
                const double um = u[r];
                cf = cf || us >= 2;
                sl = sl || (um > 0.0 && us == 0);
            }
        }
        my_lix = -1;
        if (lane < K) {
            int act = 0x7fffffff;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int id = wv_byte(my_rlo, my_rhi, d);
                const int us = usage[id];      // (row 255: never counted)
                if (us >= 2 && id < act) act = id;
            }
            if (act != 0x7fffffff) { my_lix = act; atomicAdd(&mark[act], 1); }
        }
        const bool conflict = __any(cf), slack = __any(sl);
        WV_SYNC();
        if (it == 0) stamp3 = wall_clock64();
        if (!conflict && !slack) break;           // conflict-free minimisers that use every priced row: optimal
        if (it == 1 && K == 2) return false;      // (enumerate_pair)
        // coordinate step
        bool counters_reset = false;
        if (conflict) {
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const int m = __builtin_amdgcn_readlane(my_lix, k);
                bool active = false;
                if (m >= 0) active = mark[m] == usage[m];      // (uniform address)
                double alt = DINF;
                if (active) {
                    const int cbk = __builtin_amdgcn_readlane(my_cb, k), cek = __builtin_amdgcn_readlane(my_ce, k);
                    for (int h = cbk + lane; h < cek; h += 128) {
                        const int h1 = h + 64, h1c = h1 < cek ? h1 : h;
                        const WvCol v0 = col[h], v1 = col[h1c];
                        if (!wv_has_row(v0.lo, v0.hi, m)) alt = fmin(alt, v0.rc);
                        if (h1 < cek && !wv_has_row(v1.lo, v1.hi, m)) alt = fmin(alt, v1.rc);
                    }
                    alt = wave_min_value(alt);
                }
                if (lane == k) my_mn = active ? alt - my_brc : -1.0;
            }
            // price increase of every active row, by its lowest-index user (every lane walks the members: v_readlane ignores EXEC)
            {
                double r1 = -1.0, r2 = -1.0;
                bool lowest = true;
#pragma unroll 1
                for (int j = 0; j < K; ++j) {
                    const int mj = __builtin_amdgcn_readlane(my_lix, j);
                    const double v = wv_rl64(my_mn, j);
                    if (mj == my_lix) {
                        if (j < lane) lowest = false;
                        if (v > r1) { r2 = r1; r1 = v; }
                        else if (v > r2) r2 = v;
                    }
                }
                if (lane < K && my_lix >= 0 && my_mn >= 0.0 && lowest && r2 >= 0.0 && r2 < DINF) u[my_lix] += r2 + 0.5 * fmin(r1 - r2, 1.0);
            }
            if (!slack) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int r = lane + 64 * q; if (r < nR) { usage[r] = 0; mark[r] = 0; } }
                counters_reset = true;
            } else {
                WV_SYNC();
                if (lane < K && my_lix >= 0) mark[my_lix] = 0;      // (the slack step re-uses the marks)
            }
        }
        if (slack) {
            // priced rows without a user: lowered to just below the cheapest taker (coordinate_step<LStore>'s slack part)
            const unsigned INF_BITS = 0x7f800000u;
            if (lane < K) { lixL[lane] = my_lix; mnL[lane] = my_mn; brcL[lane] = my_brc; }
            WV_SYNC();
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int r = lane + 64 * q; if (r < nR && u[r] > 0.0 && usage[r] == 0) mark[r] = (int)INF_BITS; }
            WV_SYNC();
            for (int h = lane; h < nH; h += 64) {
                int k = 0;
#pragma unroll 1
                for (int i = 1; i < K; ++i) k += (h >= colbL[i]) ? 1 : 0;
                const WvCol v = col[h];
                const bool busy = conflict && lixL[k] >= 0 && mnL[k] >= 0.0;
                const double gap = busy ? 0.0 : v.rc - brcL[k];
                float gf = (float)gap;
                if ((double)gf < gap) gf = __uint_as_float(__float_as_uint(gf) + 1u);      // round up (gap >= 0)
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const int e = wv_byte(v.lo, v.hi, d);
                    if (e != WV_NOROW && mark[e] != 0) atomicMin(reinterpret_cast<unsigned*>(&mark[e]), __float_as_uint(gf));
                }
            }
            WV_SYNC();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = lane + 64 * q;
                if (r < nR && u[r] > 0.0 && usage[r] == 0) {
                    const unsigned b = (unsigned)mark[r];
                    const double g = (b == INF_BITS) ? DINF : (double)__uint_as_float(b);
                    u[r] = fmax(0.0, u[r] - (g * (1.0 + 9.5367431640625e-7) + 1e-9));
                    mark[r] = 0;
                }
            }
        }
        if (!counters_reset) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int r = lane + 64 * q; if (r < nR) { usage[r] = 0; mark[r] = 0; } }
        }
        WV_SYNC();
    }
    const unsigned long long t_solved = wall_clock64();
    // ---- certified: every member is finished like a lone target (blp_singles) -----------------------------------------------------------
    int key = KEY_DEAD, rf = 0;
    if (lane < K) {
        const int g = pre.cb + (my_bh - my_cb);
        a.sel[tk] = g;
        if (a.sel_rel) a.sel_rel[tk] = g - pre.cb;
        key = finish_target(a, tk, g, pre, true, &rf);
    }
    // surviving leaf ranges (sweep_survivors' result, from the entries that came with the columns): count and first survivor per member
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const int cbk = __builtin_amdgcn_readlane(my_cb, k), cek = __builtin_amdgcn_readlane(my_ce, k);
        const int keyk = __builtin_amdgcn_readlane(key, k);
        int count = 0, first = 0x7fffffff;
        if (keyk == KEY_ALL) { count = cek - cbk; first = cbk; }
        else if (keyk != KEY_DEAD)
            for (int h0 = cbk; h0 < cek; h0 += 64) {
                const int h = h0 + lane;
                const unsigned long long bal = __ballot(h < cek && ancL[h < cek ? h : cbk] == keyk);
                if (bal && first == 0x7fffffff) first = h0 + __ffsll((long long)bal) - 1;
                count += __popcll(bal);
            }
        if (lane == 0) {
            const int t = __builtin_amdgcn_readlane(tk, k), gb = __builtin_amdgcn_readlane(pre.cb, k);
            const int fg = first == 0x7fffffff ? first : gb + (first - cbk);
            st_commit(&a.t_count[t], (int32_t)count, a.wt_commit != 0); st_commit(&a.t_firstsurv[t], (int32_t)fg, a.wt_commit != 0);
            if (a.rec0) blp_publish(a, t, keyk, __builtin_amdgcn_readlane(pre.j, k), __builtin_amdgcn_readlane(rf, k), count, fg);
        }
    }
    if (lane == 0) {
        st_commit(&a.cl_status[c], (int32_t)MHT_BLP_CERTIFIED, a.wt_commit != 0);
        st_commit(&a.cl_iters[c], (int32_t)iters, a.wt_commit != 0);
        a.cl_nodes[c] = 0;
        const unsigned long long t_end = wall_clock64();
        if (a.cl_time) {
            a.cl_time[8 * c] = (int)(t_setup - t_begin);
            a.cl_time[8 * c + 1] = (int)(t_end - t_begin);
            a.cl_time[8 * c + 2] = (int)(stamp1 - t_begin);
            a.cl_time[8 * c + 3] = (int)(stamp1 - t_begin);
            a.cl_time[8 * c + 4] = (int)(stamp3 - t_begin);
            a.cl_time[8 * c + 5] = (int)(t_solved - t_begin);
            a.cl_time[8 * c + 6] = 0;
            a.cl_time[8 * c + 7] = 0;
        }
        if (a.dbg && dbg_bx >= 0 && dbg_bx < 3900) {
            unsigned long long* gd = a.dbg + 32 + (size_t)dbg_bx * 16;
            gd[8] = t_begin; gd[9] = t_setup; gd[10] = t_solved; gd[11] = t_end;
        }
    }
    return true;
}

}  // namespace mht
