// Device side of the M-of-N initiator (see mht_init.hip for the description and the reference lines): shared by initiator_kernel
// (stand-alone entry, mht_initiator_step) and the forest's post_scan_kernel (commit -> initiation -> admission in one launch).
#pragma once
#include "mht_kernels.h"

namespace mht {

constexpr int INIT_THREADS = 1024;
constexpr int INIT_ECAP = 1 << 15;      // edges of one GNN problem

struct InitDev {                        // device-resident state of one initiator
    int n_seeds, n_prelim, have_last, overflow;
    double last_time;
    int n_born, n_unused_out, pad0, pad1;
};

// An AIS message on its way to the initiator (m_of_n.py:265-280): the messages of the scan in LIST order; the ones no track took (used
// flag 0) start preliminary tracks.
struct AisInitMsg { double state[4]; double dT; int32_t mmsi; int32_t pad; };      // dT = time of the scan - time of the message

struct InitArgs {
    InitDev* st;
    float* seeds;                       // [Mcap][2] last scan's leftovers
    float* pstate; float* pcov; int32_t* pn; int32_t* pm;      // preliminary tracks [Pcap]: state [4], covariance [16], n, m
    float* pstate2; float* pcov2; int32_t* pn2; int32_t* pm2;  // compaction target (swapped by the host every scan)
    int Mcap, Pcap;
    const float* z; int M;              // the scan, dev (M,2) float32
    const unsigned long long* used;     // [ceil(M/64)] bit j set = measurement j was gated by a track (null: all measurements are unused)
    const unsigned char* used_b;        // the same as one byte per measurement (what the forest's grow kernel writes); takes precedence
    double now;
    int Mreq, Nreq; double v_max, gamma, merge_threshold, default_pd;
    float C[8], R[4], P0[16];
    float sigma_q;                      // pv.Q scale (models/constants.py: sigmaQ_tracker)
    // scratch (global)
    int32_t* e_row; int32_t* e_col; double* e_cost; int32_t* e_next;   // edge lists [INIT_ECAP]
    int32_t* node_parent; int32_t* row_head; int32_t* row_next; int32_t* comp_head; int32_t* match_row; int32_t* match_col;
    double* bf_dist; int32_t* bf_pred; int32_t* comp_nodes;
    int32_t* upos;                      // [Mcap] unused-list position -> measurement index in z
    float* K;                           // [Pcap][8]
    float* pred;                        // [Pcap][4]
    int32_t* tmeas;                     // [Pcap] matched unused-list index of the track or -1
    // AIS-started preliminary tracks (null / 0: none)
    int32_t* pmmsi; int32_t* pmmsi2;    // [Pcap] identity of a track an AIS message started (0: a radar track), compacted with the others
    const AisInitMsg* ais; int nA;      // this scan's messages
    const unsigned char* ais_used;      // [nA] 1 = a track took the message (tracker.py:267-270), or null: none was taken
    double* ais_x64;                    // [Acap][4] float64 states of the tracks started in this scan (they are float32 from the update on)
    int Acap;
    // output: born candidates, in the reference's order
    double* born_x; float* born_P; uint8_t* born_flags; double* born_pd; int32_t* born_meas; int32_t* born_n; int born_cap;
    // host-mapped ring (or null): word [scan & 63] = scan << 32 | preliminary tracks kept behind that scan << 16 | candidates born in that scan, as soon as
    // the numbers exist -- the forest's host side sizes the next grids with them instead of born_cap (mht_forest.hip: Forest::births_between)
    unsigned long long* bhint; int scan_no;
    // the forest's sticky capacity flag (FCounts::overflow) when the initiator runs behind a forest's scan, else null: a candidate, track or
    // edge list that did not fit voids the forest like a full node pool does (MHT_E_CAPACITY at the scan's report) -- the stand-alone seam
    // reports InitDev::overflow from mht_initiator_born
    int32_t* forest_overflow;
};

// np.linalg.inv of the 4 x 4 float32 matrix of PreliminaryTrack.compareSimilarity (m_of_n.py:205-206): numpy.linalg computes in
// float64 and casts the result to float32 (mht_math.h: inv2).  The covariances of this initiator come from pv.P0 through pv.Phi / pv.Q
// and the radar update only (m_of_n.py:268-269, :304, :449): x and y never couple, the eight cross entries are EXACT zeros, and LAPACK's
// LU of such a matrix is its LU of the two 2 x 2 blocks {0, 2} and {1, 3} (the other operations multiply by or add an exact zero).  So:
// two float64 2 x 2 inverses, rounded once.  A matrix with a non-zero cross entry (not reachable with the reference's model) takes
// the float32 elimination of rounds 1-2 (a decision test only: d' S^-1 d <= 1).
// LU with partial pivoting of an n x n float32 system (n <= 4), as LAPACK sgetrf/sgetri order it closely enough: inverse by solving
// for the unit columns
static __device__ __attribute__((noinline)) bool inv_small_f32(const float* a_in, int n, float* out) {
    float a[16], b[16];
    for (int i = 0; i < n * n; ++i) { a[i] = a_in[i]; b[i] = 0.f; }
    for (int i = 0; i < n; ++i) b[i * n + i] = 1.f;
    for (int c = 0; c < n; ++c) {
        int p = c;
        float best = fabsf(a[c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabsf(a[r * n + c]) > best) { best = fabsf(a[r * n + c]); p = r; }
        if (best == 0.f) return false;
        if (p != c) for (int k = 0; k < n; ++k) { float t = a[c * n + k]; a[c * n + k] = a[p * n + k]; a[p * n + k] = t; t = b[c * n + k]; b[c * n + k] = b[p * n + k]; b[p * n + k] = t; }
        for (int r = c + 1; r < n; ++r) {
            const float l = a[r * n + c] / a[c * n + c];
            for (int k = c; k < n; ++k) a[r * n + k] = fmaf(-l, a[c * n + k], a[r * n + k]);
            for (int k = 0; k < n; ++k) b[r * n + k] = fmaf(-l, b[c * n + k], b[r * n + k]);
        }
    }
    for (int col = 0; col < n; ++col)
        for (int r = n - 1; r >= 0; --r) {
            float v = b[r * n + col];
            for (int k = r + 1; k < n; ++k) v = fmaf(-a[r * n + k], out[k * n + col], v);
            out[r * n + col] = v / a[r * n + r];
        }
    return true;
}


static __device__ inline bool inv_small(const float* s, int n, float* out) {
    if (n != 4 || s[1] != 0.f || s[3] != 0.f || s[4] != 0.f || s[6] != 0.f || s[9] != 0.f || s[11] != 0.f || s[12] != 0.f || s[14] != 0.f) {
        // (a real call, through copies: the general elimination indexes its arrays dynamically -- inlined, every call site carried its 128 bytes of
        // scratch and ~60 live registers; with the caller's own arrays as arguments they would have to live in memory on the fast path as well)
        float ti[16], to[16];
        for (int e = 0; e < n * n; ++e) ti[e] = s[e];
        const bool ok = inv_small_f32(ti, n, to);
        for (int e = 0; e < n * n; ++e) out[e] = to[e];
        return ok;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const float m[4] = {s[blk * 5], s[blk * 5 + 2], s[blk * 5 + 8], s[blk * 5 + 10]};      // rows / columns {blk, blk + 2}
        if (m[0] == 0.f && m[2] == 0.f) return false;
        float r[4];
        inv2(m, r);
        out[blk * 5] = r[0]; out[blk * 5 + 2] = r[1]; out[blk * 5 + 8] = r[2]; out[blk * 5 + 10] = r[3];
    }
    return true;
}

// ---- GNN: minimum-cost maximum-cardinality matching of the allowed graph -------------------------------------------------------
// rows 0..n1-1, columns 0..n2-1, edges (e_row, e_col, e_cost)[0..E).  One thread per connected component: successive shortest
// augmenting paths (Bellman-Ford on the residual graph; components have a handful of nodes).  match_row[r] = column or -1.
template <int NT, typename G>
static __device__ __forceinline__ void gnn_core(const G& a, int n1, int n2, int E) {
    const int tid = threadIdx.x;
    const int V = n1 + n2;
    for (int v = tid; v < V; v += NT) { a.node_parent[v] = v; if (v < n1) { a.row_head[v] = -1; a.comp_head[v] = -1; } }      // (heads by ROW: a component's root is its smallest node, a row)
    for (int r = tid; r < n1; r += NT) a.match_row[r] = -1;
    for (int c = tid; c < n2; c += NT) a.match_col[c] = -1;
    __threadfence_block();
    __syncthreads();
    if (E == 0) return;      // (uniform) nothing is allowed: nothing is matched -- the usual scan of a stream whose leftovers are clutter
    auto find = [&](int v) { int p = a.node_parent[v]; while (p != v) { v = p; p = a.node_parent[v]; } return v; };
    for (int e = tid; e < E; e += NT) {       // components: lock-free union (smaller root wins => the root is a row node)
        int ra = find(a.e_row[e]), rb = find(n1 + a.e_col[e]);
        while (ra != rb) {
            if (ra > rb) { const int t = ra; ra = rb; rb = t; }
            const int old = atomicCAS(&a.node_parent[rb], rb, ra);
            if (old == rb) break;
            rb = find(old);
            ra = find(ra);
        }
    }
    __threadfence_block();
    __syncthreads();
    int myroot[2] = {-1, -1};                            // (at most 2048 nodes: two per thread)
    for (int q = 0, v = tid; q < 2 && v < V; ++q, v += NT) myroot[q] = find(v);
    __syncthreads();
    for (int q = 0, v = tid; q < 2 && v < V; ++q, v += NT) a.node_parent[v] = myroot[q];      // flattened: node -> root
    for (int v = tid + 2 * NT; v < V; v += NT) a.node_parent[v] = find(v);            // (larger problems: racy but benign: roots are fixed points)
    // adjacency: edges chained per row (row_head / e_next), rows with edges chained per component (comp_head / row_next)
    for (int e = tid; e < E; e += NT) a.e_next[e] = atomicExch(&a.row_head[a.e_row[e]], e);
    __threadfence_block();
    __syncthreads();
    for (int r = tid; r < n1; r += NT)
        if (a.row_head[r] >= 0) a.row_next[r] = atomicExch(&a.comp_head[a.node_parent[r]], r);
    __threadfence_block();
    __syncthreads();
    // one thread per component: successive shortest augmenting paths.  repeat { Bellman-Ford over the residual graph from all
    // free rows; the free column with the smallest distance; augment } until no free column is reachable
    for (int root = tid; root < n1; root += NT) {
        if (a.comp_head[root] < 0) continue;
        for (int guard_aug = 0; guard_aug <= n1; ++guard_aug) {
            for (int r = a.comp_head[root]; r >= 0; r = a.row_next[r]) {
                a.bf_dist[r] = (a.match_row[r] < 0) ? 0.0 : 1e300;
                a.bf_pred[r] = -1;
                for (int e = a.row_head[r]; e >= 0; e = a.e_next[e]) { a.bf_dist[n1 + a.e_col[e]] = 1e300; a.bf_pred[n1 + a.e_col[e]] = -1; }
            }
            bool changed = true;
            for (int it = 0; it < V && changed; ++it) {
                changed = false;
                for (int r = a.comp_head[root]; r >= 0; r = a.row_next[r]) {
                    const double dr = a.bf_dist[r];
                    if (dr >= 1e299) continue;
                    for (int e = a.row_head[r]; e >= 0; e = a.e_next[e]) {
                        const int c = a.e_col[e];
                        if (a.match_row[r] == c) continue;                  // matched edge: only backwards
                        const double nd = dr + a.e_cost[e];
                        if (nd < a.bf_dist[n1 + c]) {
                            a.bf_dist[n1 + c] = nd; a.bf_pred[n1 + c] = e; changed = true;
                            const int r2 = a.match_col[c];
                            if (r2 >= 0) {                                  // follow the matched edge back to its row
                                double w = 0.0;
                                for (int e2 = a.row_head[r2]; e2 >= 0; e2 = a.e_next[e2]) if (a.e_col[e2] == c) { w = a.e_cost[e2]; break; }
                                if (nd - w < a.bf_dist[r2]) { a.bf_dist[r2] = nd - w; a.bf_pred[r2] = c; }
                            }
                        }
                    }
                }
            }
            int bc = -1;
            double bd = 1e299;
            for (int r = a.comp_head[root]; r >= 0; r = a.row_next[r])
                for (int e = a.row_head[r]; e >= 0; e = a.e_next[e]) {
                    const int c = a.e_col[e];
                    if (a.match_col[c] < 0 && (a.bf_dist[n1 + c] < bd || (a.bf_dist[n1 + c] == bd && c < bc))) { bd = a.bf_dist[n1 + c]; bc = c; }
                }
            if (bc < 0) break;
            // augment along the predecessor chain: column bc <- edge <- row <- (the column that row gives up) <- ...
            int c = bc;
            for (int guard = 0; guard < V && c >= 0; ++guard) {
                const int e = a.bf_pred[n1 + c];
                const int r = a.e_row[e];
                const int prev_c = a.match_row[r];
                a.match_row[r] = c;
                a.match_col[c] = r;
                c = prev_c;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
}


// The tables of the solver: in global scratch (InitArgs, any size), or -- GnnLds -- in the few KB of LDS the kernels that run the initiator NEXT to the
// ILP launch can have (a 155 KB workgroup of that launch leaves 8 KB of a CU's LDS): the per-component part is one THREAD walking chained lists, ~60
// dependent look-ups for the largest component of a dense scan, 0.7 us each in global memory (profiles/r06_initiator_phases.txt).  16-bit indices
// where no atomic touches them.
struct GnnGlobal {
    int32_t* e_row; int32_t* e_col; double* e_cost; int32_t* e_next; int32_t* node_parent; int32_t* row_head; int32_t* row_next; int32_t* comp_head;
    int32_t* match_row; int32_t* match_col; double* bf_dist; int32_t* bf_pred;
};
struct GnnLds {
    short* e_row; short* e_col; double* e_cost; short* e_next; int32_t* node_parent; int32_t* row_head; short* row_next; int32_t* comp_head;
    short* match_row; short* match_col; double* bf_dist; short* bf_pred;
};
__device__ __forceinline__ size_t gnn_lds_bytes(int n1, int n2, int E) {
    const size_t V = (size_t)n1 + n2;
    return (size_t)E * 8 + V * 8 + V * 4 + ((size_t)2 * n1 * 4) + (size_t)E * 6 + V * 2 + (size_t)n1 * 2 + (size_t)n1 * 2 + (size_t)n2 * 2 + 64;
}
template <int NT>
static __device__ void gnn_solve(const InitArgs& a, int n1, int n2, int E, unsigned char* lds = nullptr, int lds_bytes = 0) {
    const int tid = threadIdx.x;
    const int V = n1 + n2;
    if (lds && E > 0 && V < 32000 && E < 32000 && gnn_lds_bytes(n1, n2, E) <= (size_t)lds_bytes) {      // (uniform)
        GnnLds g;
        unsigned char* p = lds;
        g.e_cost = reinterpret_cast<double*>(p); p += (size_t)E * 8;
        g.bf_dist = reinterpret_cast<double*>(p); p += (size_t)V * 8;
        g.node_parent = reinterpret_cast<int32_t*>(p); p += (size_t)V * 4;
        g.row_head = reinterpret_cast<int32_t*>(p); p += (size_t)n1 * 4;
        g.comp_head = reinterpret_cast<int32_t*>(p); p += (size_t)n1 * 4;
        g.e_row = reinterpret_cast<short*>(p); p += (size_t)E * 2;
        g.e_col = reinterpret_cast<short*>(p); p += (size_t)E * 2;
        g.e_next = reinterpret_cast<short*>(p); p += (size_t)E * 2;
        g.bf_pred = reinterpret_cast<short*>(p); p += (size_t)V * 2;
        g.row_next = reinterpret_cast<short*>(p); p += (size_t)n1 * 2;
        g.match_row = reinterpret_cast<short*>(p); p += (size_t)n1 * 2;
        g.match_col = reinterpret_cast<short*>(p);
        for (int e = tid; e < E; e += NT) { g.e_row[e] = (short)a.e_row[e]; g.e_col[e] = (short)a.e_col[e]; g.e_cost[e] = a.e_cost[e]; }
        __syncthreads();
        gnn_core<NT>(g, n1, n2, E);
        for (int r = tid; r < n1; r += NT) a.match_row[r] = g.match_row[r];
        for (int c = tid; c < n2; c += NT) a.match_col[c] = g.match_col[c];
        __threadfence_block();
        __syncthreads();
        return;
    }
    gnn_core<NT>(GnnGlobal{a.e_row, a.e_col, a.e_cost, a.e_next, a.node_parent, a.row_head, a.row_next, a.comp_head, a.match_row, a.match_col, a.bf_dist, a.bf_pred}, n1, n2, E);
}

// One scan of the initiator, by ONE workgroup of NT threads (INIT_THREADS in initiator_kernel; the forest runs it inside post_scan_kernel,
// between the scan's commit and the admission of the new targets).
// AIS = false compiles the seeding phase (1b) out: the kernels on the path of every streamed scan (cluster_init_kernel; post_scan_kernel
// without messages) keep the register budget they had -- 1024 threads leave 128 registers, the phase's matrices spill 400 bytes per lane.
// NT = threads of the workgroup (1024 in the kernels that run it)
#ifdef MHT_INIT_STAMPS
#define INIT_STAMP(k) do { if (threadIdx.x == 0) init_t[k] = wall_clock64(); } while (0)
#else
#define INIT_STAMP(k)
#endif
__device__ __forceinline__ unsigned long long lt_mask64(int lane) { return (1ull << lane) - 1ull; }
// PreliminaryTrack.compareSimilarity: d' inv(P + R_ais) d <= 1.  deltaState.T.dot(S_inv).dot(deltaState) (m_of_n.py:207): vector x matrix = gemv (two
// interleaved FMA chains per column, added), then vector . vector = sdot (float32 products summed in float64, one rounding at the end)
__device__ __forceinline__ bool prelim_similar(float d0, float d1, float d2, float d3, const float (&Si)[16]) {
    const float t0 = fmaf(d2, Si[8], d0 * Si[0]) + fmaf(d3, Si[12], d1 * Si[4]), t1 = fmaf(d2, Si[9], d0 * Si[1]) + fmaf(d3, Si[13], d1 * Si[5]);
    const float t2 = fmaf(d2, Si[10], d0 * Si[2]) + fmaf(d3, Si[14], d1 * Si[6]), t3 = fmaf(d2, Si[11], d0 * Si[3]) + fmaf(d3, Si[15], d1 * Si[7]);
    double sacc = (double)(t0 * d0);
    sacc += (double)(t1 * d1); sacc += (double)(t2 * d2); sacc += (double)(t3 * d3);
    return (float)sacc <= 1.0f;
}

constexpr int INIT_GNN_LDS = 7424;      // LDS the kernels that run the initiator next to the ILP launch give the assignment's tables (8 KB in all with the body's own: see GnnLds)
template <bool AIS = true, int NT = INIT_THREADS>
static __device__ void initiator_body(const InitArgs& a, unsigned char* gnn_lds = nullptr, int gnn_lds_bytes = 0) {
#ifdef MHT_INIT_STAMPS
    unsigned long long init_t[10] = {};
#endif
    INIT_STAMP(0);
    __shared__ int s_cnt[8], s_scan[(NT / 64 + 1 + 3) & ~3];      // (multiples of 16 bytes: the dynamic LDS of the kernel this is inlined into stays aligned)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    InitDev& st = *a.st;
    const int n_pre = st.n_prelim, n_seed = st.n_seeds, have_last = st.have_last;
    const double last = st.last_time;
    if (tid < 8) s_cnt[tid] = 0;
    __syncthreads();
    // ---- the unused measurements, in ascending index order (MeasurementList.filterUnused) ----------------------------------------
    int nU = 0;
    {
        int running = 0;
        for (int base = 0; base < a.M; base += NT) {
            const int j = base + tid;
            const bool un = j < a.M && !(a.used_b ? a.used_b[j] != 0 : (a.used && ((a.used[j >> 6] >> (j & 63)) & 1ull)));
            const unsigned long long bal = __ballot(un);
            if (lane == 0) s_scan[wave] = __popcll(bal);
            __syncthreads();
            int off = running;
            for (int w = 0; w < wave; ++w) off += s_scan[w];
            int tot = 0;
            for (int w = 0; w < NT / 64; ++w) tot += s_scan[w];
            if (un) a.upos[off + __popcll(bal & ((1ull << lane) - 1ull))] = j;
            running += tot;
            __syncthreads();
        }
        nU = running;
    }
    __threadfence_block();
    __syncthreads();
    INIT_STAMP(1);
    // ---- (1) preliminary tracks (m_of_n.py:246-378) --------------------------------------------------------------------------------
    // AIS messages no track took (in list order): they start preliminary tracks below, and with any of them the scan is processed even
    // without a single unused radar measurement (m_of_n.py:289-292)
    int nAu = 0;
    if (AIS) for (int q = 0; q < a.nA; ++q) nAu += (a.ais_used && a.ais_used[q]) ? 0 : 1;      // (uniform; a few dozen messages)
    const bool frozen = (nU == 0 && nAu == 0);
    int E = 0;
    const int n_pre0 = n_pre;
    int n_pre_all = n_pre0;
    if (n_pre0 > 0 && have_last) {
        const double dt = a.now - last;
        // pv.Phi(dt), pv.Q(dt): float64 arithmetic cast to float32, Q then scaled by sigmaQ in float32 (models/pv.py:17-34)
        float F[16] = {1, 0, (float)dt, 0, 0, 1, 0, (float)dt, 0, 0, 1, 0, 0, 0, 0, 1};
        float Q[16];
        for (int i = 0; i < 16; ++i) Q[i] = 0.f;
        const float q4 = (float)(dt * dt * dt * dt / 4.0) * a.sigma_q, q3 = (float)(dt * dt * dt / 3.0) * a.sigma_q, q2 = (float)(dt * dt) * a.sigma_q;
        Q[0] = Q[5] = q4; Q[2] = Q[8] = Q[7] = Q[13] = q3; Q[10] = Q[15] = q2;
        for (int i = tid; i < n_pre; i += NT) {
            const float* x = a.pstate + (size_t)i * 4;
            const float* P = a.pcov + (size_t)i * 16;
            float xp[4], FP[16], Ft[16], Pb[16];
            for (int r = 0; r < 4; ++r) xp[r] = gemv_row<float, 4>(F + r * 4, x);      // F.dot(state): matrix x vector = BLAS gemv (m_of_n.py:187; mht_math.h::gemv_row)
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Ft[r * 4 + c] = F[c * 4 + r];
            gemm_chain<float, float, float, 4, 4, 4>(F, P, FP);
            gemm_chain<float, float, float, 4, 4, 4>(FP, Ft, Pb);
            for (int e = 0; e < 16; ++e) Pb[e] += Q[e];
            float Ct[8], CP[8], S[4], Sinv[4], PCt[8], K[8];
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 4; ++c) Ct[c * 2 + r] = a.C[r * 4 + c];
            gemm_chain<float, float, float, 2, 4, 4>(a.C, Pb, CP);
            gemm_chain<float, float, float, 2, 4, 2>(CP, Ct, S);
            for (int e = 0; e < 4; ++e) S[e] += a.R[e];
            inv2(S, Sinv);
            gemm_chain<float, float, float, 4, 4, 2>(Pb, Ct, PCt);
            gemm_chain<float, float, float, 4, 2, 2>(PCt, Sinv, K);
            for (int e = 0; e < 4; ++e) a.pred[(size_t)i * 4 + e] = xp[e];
            for (int e = 0; e < 16; ++e) a.pcov[(size_t)i * 16 + e] = Pb[e];      // (covariance = P_bar until the update below)
            if (frozen) continue;      // (see below: an empty list leaves state and counters alone)
            for (int e = 0; e < 8; ++e) a.K[(size_t)i * 8 + e] = K[e];
            a.tmeas[i] = -1;
        }
        __threadfence_block();
        __syncthreads();
    }
    // ---- (1b) messages no track took start preliminary tracks (m_of_n.py:262-280), one after the other: each is tested against every
    //      track there is by then, the ones started a moment ago included -------------------------------------------------------------
    if (AIS && nAu > 0) {
        __shared__ int s_hn[4];
        int& s_hit = s_hn[0];
        int& s_nall = s_hn[1];
        if (tid == 0) s_nall = n_pre0;
        __syncthreads();
        for (int q = 0; q < a.nA; ++q) {
            if (a.ais_used && a.ais_used[q]) continue;                     // (uniform)
            const AisInitMsg m = a.ais[q];
            if (tid == 0) s_hit = 0;
            __syncthreads();
            const int nall = s_nall;
            for (int p = tid; p < n_pre0; p += NT) if (a.pmmsi[p] == m.mmsi) s_hit = 1;      // a track with this identity exists (m_of_n.py:262-267)
            __syncthreads();
            if (s_hit) { __syncthreads(); continue; }
            // state = Phi(dT) m.state (float32 matrix x float64 vector = float64 gemv), covariance = Phi P0 Phi^T + Q (classDefinitions.py:470-475)
            const double dT = m.dT;
            const float Ff[16] = {1, 0, (float)dT, 0, 0, 1, 0, (float)dT, 0, 0, 1, 0, 0, 0, 0, 1};
            double cs[4];
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {
                const double p0 = (double)Ff[r * 4] * m.state[0], p1 = (double)Ff[r * 4 + 1] * m.state[1], p2 = (double)Ff[r * 4 + 2] * m.state[2], p3 = (double)Ff[r * 4 + 3] * m.state[3];
                cs[r] = (p0 + p2) + (p1 + p3);
            }
            for (int p = tid; p < nall; p += NT) {               // PreliminaryTrack.compareSimilarity (m_of_n.py:196-201) of every track with the candidate
                double d[4];
                _Pragma("unroll") for (int e = 0; e < 4; ++e) d[e] = (p < n_pre0 ? (double)a.pstate[(size_t)p * 4 + e] : a.ais_x64[(size_t)(p - n_pre0) * 4 + e]) - cs[e];
                float S[16], Si[16];
                _Pragma("unroll") for (int e = 0; e < 16; ++e) S[e] = a.pcov[(size_t)p * 16 + e] + ((e % 5 == 0) ? 9.0f : 0.f);
                if (inv_small(S, 4, Si)) {
                    double t[4];
                    _Pragma("unroll") for (int c = 0; c < 4; ++c) t[c] = fma(d[2], (double)Si[8 + c], d[0] * (double)Si[c]) + fma(d[3], (double)Si[12 + c], d[1] * (double)Si[4 + c]);
                    const double sim = ((t[0] * d[0] + t[1] * d[1]) + t[2] * d[2]) + t[3] * d[3];
                    if (sim <= 1.0) s_hit = 1;
                }
            }
            __syncthreads();
            if (!s_hit) {
                if (nall < a.Pcap && nall - n_pre0 < a.Acap) {
                    if (tid == 0) {
                        float Q[16], Ft[16], FP[16], Pb[16];
                        _Pragma("unroll") for (int i = 0; i < 16; ++i) Q[i] = 0.f;
                        const float q4 = (float)(dT * dT * dT * dT / 4.0) * a.sigma_q, q3 = (float)(dT * dT * dT / 3.0) * a.sigma_q, q2 = (float)(dT * dT) * a.sigma_q;
                        Q[0] = Q[5] = q4; Q[2] = Q[8] = Q[7] = Q[13] = q3; Q[10] = Q[15] = q2;
                        _Pragma("unroll") for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Ft[r * 4 + c] = Ff[c * 4 + r];
                        gemm_chain<float, float, float, 4, 4, 4>(Ff, a.P0, FP);
                        gemm_chain<float, float, float, 4, 4, 4>(FP, Ft, Pb);
                        _Pragma("unroll") for (int e = 0; e < 16; ++e) Pb[e] += Q[e];
                        float Ct[8], CP[8], S2[4], Sinv[4], PCt[8], K[8];
                        _Pragma("unroll") for (int r = 0; r < 2; ++r) for (int c = 0; c < 4; ++c) Ct[c * 2 + r] = a.C[r * 4 + c];
                        gemm_chain<float, float, float, 2, 4, 4>(a.C, Pb, CP);
                        gemm_chain<float, float, float, 2, 4, 2>(CP, Ct, S2);
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) S2[e] += a.R[e];
                        inv2(S2, Sinv);
                        gemm_chain<float, float, float, 4, 4, 2>(Pb, Ct, PCt);
                        gemm_chain<float, float, float, 4, 2, 2>(PCt, Sinv, K);
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) {
                            a.ais_x64[(size_t)(nall - n_pre0) * 4 + e] = cs[e];
                            a.pstate[(size_t)nall * 4 + e] = (float)cs[e];
                            a.pred[(size_t)nall * 4 + e] = (float)cs[e];      // np.array(predicted_states, dtype=float32) (m_of_n.py:282-284)
                        }
                        _Pragma("unroll") for (int e = 0; e < 16; ++e) a.pcov[(size_t)nall * 16 + e] = Pb[e];
                        _Pragma("unroll") for (int e = 0; e < 8; ++e) a.K[(size_t)nall * 8 + e] = K[e];
                        a.pn[nall] = 0; a.pm[nall] = 0; a.pmmsi[nall] = m.mmsi; a.tmeas[nall] = -1;
                        s_nall = nall + 1;
                    }
                } else if (tid == 0) st.overflow = 1;
            }
            __threadfence_block();
            __syncthreads();
        }
        n_pre_all = s_nall;
        __syncthreads();
    }
    {
        const int n_pre = n_pre_all;      // (from here on: the tracks of this scan, AIS-started ones included)
        // The reference returns from __processPreliminaryTracks right after the prediction when there is nothing to work on -- no unused
        // measurement and no unused message (m_of_n.py:289-292): the covariances have been propagated, states, counters and the track
        // list stay as they are.
        if (n_pre > 0 && !frozen) {
        if (nU == 0) {
            for (int i = tid; i < n_pre; i += NT) a.match_row[i] = -1;
            __threadfence_block();
            __syncthreads();
        } else {
        // gate: (track, unused measurement) pairs with NIS <= gamma -> edges with the Euclidean distance as cost
        const long long npairs = (long long)n_pre * nU;
        for (long long w = tid; w < npairs; w += NT) {
            const int i = (int)(w / nU), k = (int)(w % nU);
            const int j = a.upos[k];
            const float* xp = a.pred + (size_t)i * 4;
            const float* Pb = a.pcov + (size_t)i * 16;
            float zh[2], Ct[8], CP[8], S[4], Sinv[4];
            for (int r = 0; r < 2; ++r) zh[r] = gemv_row<float, 4>(a.C + r * 4, xp);      // C.dot(predicted_state): gemv (m_of_n.py:283)
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 4; ++c) Ct[c * 2 + r] = a.C[r * 4 + c];
            gemm_chain<float, float, float, 2, 4, 4>(a.C, Pb, CP);
            gemm_chain<float, float, float, 2, 4, 2>(CP, Ct, S);
            for (int e = 0; e < 4; ++e) S[e] += a.R[e];
            inv2(S, Sinv);
            const float dx = a.z[2 * j] - zh[0], dy = a.z[2 * j + 1] - zh[1];
            const float t0 = fmaf(dy, Sinv[2], dx * Sinv[0]), t1 = fmaf(dy, Sinv[3], dx * Sinv[1]);
            const float nis = t0 * dx + t1 * dy;
            if ((double)nis <= a.gamma) {
                const int e = atomicAdd(&s_cnt[0], 1);
                if (e < INIT_ECAP) { a.e_row[e] = i; a.e_col[e] = k; a.e_cost[e] = (double)sqrtf(dx * dx + dy * dy); }
            }
        }
        __syncthreads();
        E = s_cnt[0];
        if (E > INIT_ECAP) { if (tid == 0) st.overflow = 1; E = INIT_ECAP; }
        __threadfence_block();
        __syncthreads();
        gnn_solve<NT>(a, n_pre, nU, E, gnn_lds, gnn_lds_bytes);
        }
        // Kalman update of the matched tracks, counters
        for (int i = tid; i < n_pre; i += NT) {
            const int k = a.match_row[i];
            float* x = a.pstate + (size_t)i * 4;
            const float* xp = a.pred + (size_t)i * 4;
            if (k >= 0) {
                const int j = a.upos[k];
                float* Pb = a.pcov + (size_t)i * 16;
                const float* K = a.K + (size_t)i * 8;
                float zh[2];
                for (int r = 0; r < 2; ++r) zh[r] = gemv_row<float, 4>(a.C + r * 4, xp);      // m_of_n.py:302
                const float dz[2] = {a.z[2 * j] - zh[0], a.z[2 * j + 1] - zh[1]};
                float Kd[4], KC[16], KCP[16];
                for (int r = 0; r < 4; ++r) Kd[r] = gemv_row<float, 2>(K + r * 2, dz);      // K.dot(delta_vector): gemv, two separately rounded products (m_of_n.py:303)
                for (int e = 0; e < 4; ++e) x[e] = xp[e] + Kd[e];
                gemm_chain<float, float, float, 4, 2, 4>(K, a.C, KC);
                gemm_chain<float, float, float, 4, 4, 4>(KC, Pb, KCP);
                for (int e = 0; e < 16; ++e) Pb[e] = Pb[e] - KCP[e];
                a.pm[i] += 1;
                a.tmeas[i] = k;
            } else {
                for (int e = 0; e < 4; ++e) x[e] = xp[e];
            }
            a.pn[i] += 1;
        }
        }
        __threadfence_block();
        __syncthreads();
    }
    // verdicts (none when the scan was not processed, see above), births (in track order), compaction of the surviving preliminary tracks
    // into the second buffer
    int n_keep = 0, n_born = 0;
    {
        const int n_pre = n_pre_all;
        int run_keep = 0, run_born = 0;
        for (int base = 0; base < n_pre; base += NT) {
            const int i = base + tid;
            int keep = 0, born = 0;
            if (i < n_pre && frozen) keep = 1;
            else if (i < n_pre) {
                const float* x = a.pstate + (size_t)i * 4;
                const int verdict = (a.pm[i] >= a.Mreq) ? 1 : ((a.pn[i] >= a.Nreq) ? -1 : 0);
                const float speed = sqrtf(x[2] * x[2] + x[3] * x[3]);
                if ((double)speed > a.v_max * 1.5 || verdict == -1) { keep = 0; }
                else if (verdict == 1) { born = 1; }
                else keep = 1;
            }
            const unsigned long long bk = __ballot(keep), bb = __ballot(born);
            if (lane == 0) { s_scan[wave] = __popcll(bk) | (__popcll(bb) << 16); }
            __syncthreads();
            int offk = run_keep, offb = run_born, totk = 0, totb = 0;
            for (int w = 0; w < NT / 64; ++w) {
                const int v = s_scan[w];
                if (w < wave) { offk += v & 0xffff; offb += v >> 16; }
                totk += v & 0xffff; totb += v >> 16;
            }
            if (keep) {
                const int p = offk + __popcll(bk & ((1ull << lane) - 1ull));
                for (int e = 0; e < 4; ++e) a.pstate2[(size_t)p * 4 + e] = a.pstate[(size_t)i * 4 + e];
                for (int e = 0; e < 16; ++e) a.pcov2[(size_t)p * 16 + e] = a.pcov[(size_t)i * 16 + e];
                a.pn2[p] = a.pn[i]; a.pm2[p] = a.pm[i];
                if (a.pmmsi) a.pmmsi2[p] = a.pmmsi[i];
            }
            if (born) {
                const int p = offb + __popcll(bb & ((1ull << lane) - 1ull));
                if (p < a.born_cap) {
                    for (int e = 0; e < 4; ++e) a.born_x[(size_t)p * 4 + e] = (double)a.pstate[(size_t)i * 4 + e];
                    for (int e = 0; e < 16; ++e) a.born_P[(size_t)p * 16 + e] = a.pcov[(size_t)i * 16 + e];
                    a.born_meas[p] = a.tmeas[i] + 1;          // measurementNumber = index in the unused list + 1 (m_of_n.py:353-358)
                } else st.overflow = 1;
            }
            run_keep += totk; run_born += totb;
            __syncthreads();
        }
        n_keep = run_keep; n_born = run_born < a.born_cap ? run_born : a.born_cap;
    }
    __threadfence_block();
    __syncthreads();
    INIT_STAMP(2);
    // ---- unused' = unused measurements no preliminary track took (ascending) -> comp_nodes[0..nU2) holds their unused-list indices
    int nU2 = 0;
    {
        int running = 0;
        const bool any_match = n_pre_all > 0 && nU > 0;      // (the assignment ran: match_col is this scan's)
        for (int base = 0; base < nU; base += NT) {
            const int k = base + tid;
            const bool free = k < nU && !(any_match && a.match_col[k] >= 0);
            const unsigned long long bal = __ballot(free);
            if (lane == 0) s_scan[wave] = __popcll(bal);
            __syncthreads();
            int off = running, tot = 0;
            for (int w = 0; w < NT / 64; ++w) { if (w < wave) off += s_scan[w]; tot += s_scan[w]; }
            if (free) a.comp_nodes[off + __popcll(bal & ((1ull << lane) - 1ull))] = k;
            running += tot;
            __syncthreads();
        }
        nU2 = running;
    }
    __threadfence_block();
    __syncthreads();
    INIT_STAMP(3);
    // ---- (2) pair the leftovers with last scan's initiators (m_of_n.py:380-478) ---------------------------------------------------
    int n_pre_now = n_keep;
    if (n_seed > 0 && nU2 > 0) {
        if (tid == 0) s_cnt[1] = 0;
        __syncthreads();
        const double dts = a.now - last;                       // (all initiators carry last scan's time stamp)
        const double gate = a.v_max * dts;
        const long long npairs = (long long)n_seed * nU2;
        for (long long w = tid; w < npairs; w += NT) {
            const int i = (int)(w / nU2), q = (int)(w % nU2);
            const int j = a.upos[a.comp_nodes[q]];
            const float dx = a.z[2 * j] - a.seeds[2 * i], dy = a.z[2 * j + 1] - a.seeds[2 * i + 1];      // float32 differences ...
            const double d = sqrt((double)dx * (double)dx + (double)dy * (double)dy);                    // ... float64 norm
            if (!(d > gate)) {
                const int e = atomicAdd(&s_cnt[1], 1);
                if (e < INIT_ECAP) { a.e_row[e] = i; a.e_col[e] = q; a.e_cost[e] = d; }
            }
        }
        __syncthreads();
        int E2 = s_cnt[1];
        if (E2 > INIT_ECAP) { if (tid == 0) st.overflow = 1; E2 = INIT_ECAP; }
        __threadfence_block();
        __syncthreads();
        INIT_STAMP(4);
        gnn_solve<NT>(a, n_seed, nU2, E2, gnn_lds, gnn_lds_bytes);
        INIT_STAMP(5);
        // new preliminary tracks in initiator order, each tested against every track kept so far (sequential like the reference: m_of_n.py:440-452).
        // 64 initiators at a time (every wavefront fetches the same matches and each lane forms its own candidate):
        //   (a) the block's candidates against the tracks there were at the block's start -- all threads, a track's S^-1 once for all candidates;
        //   (b) the order dependence INSIDE the block -- a candidate is also tested against the block's earlier candidates that were kept, whose
        //       covariance is P0 -- resolved by wavefront 0 with shuffles, no barrier;  (c) the kept ones appended.
        // Two barriers per 64 initiators; until round 6 it was three per MATCH with a global round trip between them (38-43 of the kernel's
        // 120 us on a stream with ~60 initiators per scan, profiles/r06_initiator_phases.txt).
        __shared__ int s_sn[4];      // (16 bytes, see s_cnt)
        unsigned* s_old = reinterpret_cast<unsigned*>(&s_sn[2]);      // [2] bit b: candidate b of the block is similar to a track kept before the block
        __shared__ float s_si0[16];  // inv(P0 + R_ais), [0] of s_sn: it exists
        int& s_np = s_sn[1];
        if (tid == 0) s_np = n_keep;
        __syncthreads();
        bool have_si0 = false;
        // PreliminaryTrack.compareSimilarity: d' inv(P + R_ais) d <= 1.  deltaState.T.dot(S_inv).dot(deltaState) (m_of_n.py:207): vector x matrix =
        // gemv (two interleaved FMA chains per column, added), then vector . vector = sdot (float32 products summed in float64, one rounding at the end)
        for (int base = 0; base < n_seed; base += 64) {
          const int i_l = base + lane;
          const int qv = (i_l < n_seed) ? a.match_row[i_l] : -1;      // (global memory, written before the barrier)
          float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
          if (qv >= 0) {
              const int j = a.upos[a.comp_nodes[qv]];
              c0 = a.z[2 * j]; c1 = a.z[2 * j + 1];
              c2 = (c0 - a.seeds[2 * i_l]) / (float)dts; c3 = (c1 - a.seeds[2 * i_l + 1]) / (float)dts;
          }
          const unsigned long long todo = __ballot(qv >= 0);      // (uniform: the same in every wavefront)
          if (todo == 0ull) continue;
          if (tid < 2) s_old[tid] = 0u;
          __syncthreads();
          const int np0 = s_np;
          {   // (a)
              unsigned long long hit = 0ull;
              // (uniform bound: the shuffles below need every lane of the wavefront; lanes beyond the last track redo track 0.  One slot more than
              // there are tracks: its thread inverts S of a NEW track, covariance P0, for (b) -- through this call site: inv_small's general
              // fallback indexes dynamically, every inlined copy is another 128 bytes of scratch per lane)
#pragma nounroll
              for (int pb = 0; pb <= np0; pb += NT) {
                  const bool have = pb + tid < np0, virt = pb + tid == np0 && !have_si0;
                  const int p = have ? pb + tid : 0;
                  float xs[4], S[16], Si[16];
                  for (int e = 0; e < 4; ++e) xs[e] = a.pstate2[(size_t)p * 4 + e];
                  for (int e = 0; e < 16; ++e) S[e] = (virt ? a.P0[e] : a.pcov2[(size_t)p * 16 + e]) + ((e % 5 == 0) ? 9.0f : 0.f);
                  const bool inv_ok = inv_small(S, 4, Si);
                  if (virt) { for (int e = 0; e < 16; ++e) s_si0[e] = Si[e]; s_sn[0] = inv_ok ? 1 : 0; }
                  const bool ok = have && inv_ok;
                  unsigned long long td = todo;
#pragma nounroll
                  while (td) {      // (uniform)
                      const int b = __ffsll((long long)td) - 1;
                      td &= td - 1ull;
                      const float d[4] = {xs[0] - __shfl(c0, b), xs[1] - __shfl(c1, b), xs[2] - __shfl(c2, b), xs[3] - __shfl(c3, b)};
                      if (ok && prelim_similar(d[0], d[1], d[2], d[3], Si)) hit |= 1ull << b;
                  }
              }
              unsigned lo = (unsigned)hit, hi = (unsigned)(hit >> 32);
              for (int o = 32; o > 0; o >>= 1) { lo |= __shfl_xor(lo, o); hi |= __shfl_xor(hi, o); }
              if (lane == 0 && (lo | hi)) { if (lo) atomicOr(&s_old[0], lo); if (hi) atomicOr(&s_old[1], hi); }
          }
          __syncthreads();
          if (tid < 64) {   // (b), (c): wavefront 0
              const unsigned long long old = ((unsigned long long)s_old[1] << 32) | s_old[0];
              float Si0[16];                                 // a new track's covariance is P0: S = P0 + R_ais, the same for all of them
              for (int e = 0; e < 16; ++e) Si0[e] = s_si0[e];
              const bool inv0 = s_sn[0] != 0;
              bool mine = qv >= 0 && !((old >> lane) & 1ull);      // still a candidate
              unsigned long long td = todo & ~old;
              while (td) {      // (uniform) in initiator order: candidate b is kept; later candidates similar to it are not
                  const int b = __ffsll((long long)td) - 1;
                  td &= td - 1ull;
                  const float d[4] = {__shfl(c0, b) - c0, __shfl(c1, b) - c1, __shfl(c2, b) - c2, __shfl(c3, b) - c3};      // (track - candidate)
                  if (mine && lane > b && inv0 && prelim_similar(d[0], d[1], d[2], d[3], Si0)) mine = false;
                  td &= __ballot(mine);
              }
              const unsigned long long kept = __ballot(mine);
              const int n_new = __popcll(kept);
              if (mine) {
                  const int np = np0 + __popcll(kept & lt_mask64(lane));
                  if (np < a.Pcap) {
                      const float cand[4] = {c0, c1, c2, c3};
                      for (int e = 0; e < 4; ++e) a.pstate2[(size_t)np * 4 + e] = cand[e];
                      for (int e = 0; e < 16; ++e) a.pcov2[(size_t)np * 16 + e] = a.P0[e];
                      a.pn2[np] = 0; a.pm2[np] = 0; if (a.pmmsi2) a.pmmsi2[np] = 0;
                  } else st.overflow = 1;
              }
              if (lane == 0) s_np = (np0 + n_new < a.Pcap) ? np0 + n_new : a.Pcap;
          }
          have_si0 = true;
          __threadfence_block();
          __syncthreads();
        }
        n_pre_now = s_np;
    }
    INIT_STAMP(6);
    // ---- (3) next scan's initiators = leftovers nobody paired (ascending) ----------------------------------------------------------
    int n_left = 0;
    {
        int running = 0;
        const bool paired = n_seed > 0 && nU2 > 0;
        for (int base = 0; base < nU2; base += NT) {
            const int q = base + tid;
            const bool left = q < nU2 && !(paired && a.match_col[q] >= 0);
            const unsigned long long bal = __ballot(left);
            if (lane == 0) s_scan[wave] = __popcll(bal);
            __syncthreads();
            int off = running, tot = 0;
            for (int w = 0; w < NT / 64; ++w) { if (w < wave) off += s_scan[w]; tot += s_scan[w]; }
            if (left) {
                const int p = off + __popcll(bal & ((1ull << lane) - 1ull));
                const int j = a.upos[a.comp_nodes[q]];
                // (the seeds of THIS scan are still being read above?  no: every read of a.seeds sits before the last barrier)
                if (p < a.Mcap) { a.bf_dist[2 * p] = (double)a.z[2 * j]; a.bf_dist[2 * p + 1] = (double)a.z[2 * j + 1]; }
            }
            running += tot;
            __syncthreads();
        }
        n_left = running < a.Mcap ? running : a.Mcap;
    }
    __threadfence_block();
    __syncthreads();
    for (int p = tid; p < n_left; p += NT) { a.seeds[2 * p] = (float)a.bf_dist[2 * p]; a.seeds[2 * p + 1] = (float)a.bf_dist[2 * p + 1]; }
    INIT_STAMP(7);
    // ---- merge confirmed candidates closer than the threshold (m_of_n.py:133-154): greedy, in order; a handful at most -------------
    if (tid == 0) {
        int nb = n_born, out = 0;
        // `used` flags in born_flags (scratch until the end)
        for (int i = 0; i < nb; ++i) a.born_flags[i] = 0;
        for (int i = 0; i < nb; ++i) {
            if (a.born_flags[i]) continue;
            double sx[4] = {0, 0, 0, 0};
            float sP[16];
            for (int e = 0; e < 16; ++e) sP[e] = 0.f;
            int cnt = 0, first = -1;
            float fsx[4] = {0, 0, 0, 0};
            for (int o = 0; o < nb; ++o) {
                const float dx = (float)a.born_x[(size_t)i * 4] - (float)a.born_x[(size_t)o * 4], dy = (float)a.born_x[(size_t)i * 4 + 1] - (float)a.born_x[(size_t)o * 4 + 1];
                const float d = sqrtf(dx * dx + dy * dy);
                if ((double)d < a.merge_threshold) {
                    if (!a.born_flags[o]) {
                        if (first < 0) first = o;
                        for (int e = 0; e < 4; ++e) fsx[e] += (float)a.born_x[(size_t)o * 4 + e];
                        for (int e = 0; e < 16; ++e) sP[e] += a.born_P[(size_t)o * 16 + e];
                        ++cnt;
                    }
                    a.born_flags[o] = 1;
                }
            }
            (void)sx;
            // results are written in place at `out` <= i (entries before i are final or consumed)
            if (cnt == 1) {
                if (out != first) {
                    for (int e = 0; e < 4; ++e) a.born_x[(size_t)out * 4 + e] = a.born_x[(size_t)first * 4 + e];
                    for (int e = 0; e < 16; ++e) a.born_P[(size_t)out * 16 + e] = a.born_P[(size_t)first * 16 + e];
                    a.born_meas[out] = a.born_meas[first];
                }
            } else {
                for (int e = 0; e < 4; ++e) a.born_x[(size_t)out * 4 + e] = (double)(fsx[e] / (float)cnt);
                for (int e = 0; e < 16; ++e) a.born_P[(size_t)out * 16 + e] = sP[e] / (float)cnt;
                a.born_meas[out] = 0;                          // the merged Target has no measurementNumber (m_of_n.py:150)
            }
            ++out;
        }
        for (int i = 0; i < out; ++i) { a.born_flags[i] = F_STATE_F32 | F_SCORE_F32; a.born_pd[i] = a.default_pd; }
        *a.born_n = out;
        if (a.bhint) __hip_atomic_store(a.bhint + (a.scan_no & 63), ((unsigned long long)(unsigned)a.scan_no << 32) | ((unsigned long long)(unsigned)(n_pre_now & 0xffff) << 16) | (unsigned long long)(unsigned)(out & 0xffff),
                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        st.n_born = out;
        st.n_prelim = n_pre_now;
        st.n_seeds = n_left;
        st.have_last = 1;
        st.last_time = a.now;
        st.n_unused_out = nU;
        if (st.overflow && a.forest_overflow) *a.forest_overflow = 1;
#ifdef MHT_INIT_STAMPS
        init_t[8] = wall_clock64();
        if (a.scan_no == 100 || a.scan_no == 101 || a.scan_no == 200 || a.scan_no == 201)
            printf("[init %d] nU %d n_seed %d: head %.2f | phase1 %.2f | unused' %.2f | pairs %.2f | gnn %.2f | new tracks %.2f | leftovers %.2f | merge+end %.2f | total %.2f us\n", a.scan_no, nU, n_seed,
                   1e-2 * (double)(init_t[1] - init_t[0]), 1e-2 * (double)(init_t[2] - init_t[1]), 1e-2 * (double)(init_t[3] - init_t[2]), 1e-2 * (double)(init_t[4] - init_t[3]),
                   1e-2 * (double)(init_t[5] - init_t[4]), 1e-2 * (double)(init_t[6] - init_t[5]), 1e-2 * (double)(init_t[7] - init_t[6]), 1e-2 * (double)(init_t[8] - init_t[7]), 1e-2 * (double)(init_t[8] - init_t[0]));
#endif
    }
}

}  // namespace mht
