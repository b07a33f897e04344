// Target side of a scan's end (shared by mht_forest.hip and mht_gate.hip): the compaction of the target table after track
// termination + N-scan pruning (tracker.py:353-381, :1219-1231), the next scan's leaf ranges, the scan report.  It runs either
// as commit_kernel (when the host needs the result now: report, births) or, deferred, in workgroup 0 of the NEXT scan's
// grow_kernel, whose tiles derive the same tables for themselves in LDS (grow_kernel, "deferred commit").
#pragma once
#include "mht_kernels.h"

namespace mht {

constexpr int COMMIT_THREADS = 512;   // commit_kernel (standalone); fgrow_kernel runs the same body with its own width

struct FCounts {          // device-side counters of the forest
    int nT;               // targets in the NEXT table
    int L;                // leaves in the NEXT leaf list
    int n_nodes;          // nodes in the newest layer (children + roots born after the scan)
    int n_roots;          // roots born into the newest layer
    int id_counter;       // Tracker.trackIdCounter
    int overflow;         // sticky capacity flag
    int n_children;       // children of the last scan
    int L_in;             // leaves gated in the last scan
    int nTv[2];           // nT by table version: nTv[s & 1] = targets in the table scan s runs on.  The tiles of a grow_kernel that
                          // carries the previous scan's commit read the OLD count here while that commit rewrites nT
    // a grow launch that also carries the ADMISSION of what the initiator gave birth to (mht_fgrow.hip: fgrow_adm_kernel): workgroup 0
    // posts the committed scan's number, the first newborn slot and the number of newborn targets here; the workgroups that would grow
    // newborn targets wait for it (and leave at once when nothing was born: no fence on either side then)
    int pad_;
    unsigned long long adm_flag;      // scan << 32 | first newborn slot << 16 | number of newborn targets
    // a grow launch that OVERLAPS the previous scan's ILP launch (launched any-order behind it, mht_forest.hip): the workgroups of
    // blp_uf_kernel count themselves off here when their results are released (never reset: the host knows the total) ...
    unsigned long long blp_done;
    // ... and the commit that rides in that grow launch posts the scan whose compacted indices (new_index) are valid; bit 32: some target
    // died in it (the indices differ from the slots)
    unsigned long long ni_flag;
    // the scan whose initiator (initiator_side_kernel, launched any-order behind the scan's ILP launch) has finished: the admission in the
    // next grow launch waits for it when that launch overlaps (FDyn::adm_wait)
    unsigned long long init_flag;
    // the staging kernel's tag (stage_scan_kernel): the scan in the device buffer is complete when it equals FDyn::z_tag
    unsigned long long z_flag;
    unsigned long long init_tick;      // ticket {scan, count} of initiator_side_kernel (first_come_ticket)
    // the scan whose ILP launch has begun, i.e. whose grow launch is complete (blp_uf_kernel, first workgroup): what the scan's initiator
    // waits for when it is launched on a queue of its own (initiator_side_kernel)
    unsigned long long ilp_begun;
};
// Ticket among the few workgroups of a launch that may play a role: 0 for the first one to arrive in launch `tag`, 1, 2, ... for the
// others.  The word carries the tag of the launch it was last used in, so nothing has to be reset (launches may skip the scheme).
__device__ __forceinline__ unsigned first_come_ticket(unsigned long long* word, unsigned tag) {
    unsigned long long w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if ((unsigned)(w >> 32) != tag) {
            const unsigned long long old = atomicCAS(word, w, ((unsigned long long)tag << 32) | 1ull);
            if (old == w) return 0u;
            w = old;
        } else {
            return (unsigned)atomicAdd(word, 1ull);      // (low half: arrivals so far)
        }
    }
}
// Spin on a word another kernel / workgroup publishes (agent-scope loads, s_sleep between polls).  Bounded: a wait that does not end
// within ~2 s gives up (returns false) instead of hanging the device; the caller voids the scan.
constexpr unsigned long long SPIN_TICKS = 200000000ull;      // 10 ns ticks
template <typename PRED>
__device__ __forceinline__ bool spin_until(const unsigned long long* p, PRED ok, unsigned long long& v) {
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ok(v)) return true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ok(v)) return true;
        if (wall_clock64() - t0 > SPIN_TICKS) return false;
    }
}

struct TTable {           // one buffer of the target table
    int32_t* id; int32_t* window; int32_t* depth; int32_t* shift; int32_t* root_scan; int32_t* root_node;
    double* root_cnllr; uint8_t* root_f32;
    int32_t* first;       // node index (newest layer) of the target's first leaf; its leaves are contiguous
    int32_t* leaf_off;    // [T+1] exclusive prefix of the leaf counts
};

struct ReportHeader {     // device image of mht_scan_report up to the host pointers
    int32_t scan, n_targets, n_alive, n_leaves_in, n_children, n_leaves_out, n_clusters, n_ilp, n_branched, n_limit,
        blp_iters_max, error, used_words, n_births, pad[2];
    int32_t t_process, t_cluster, t_optim, t_scan;      // device time of the stages in 10 ns ticks (mht_scan_report)
};

struct CommitDyn { int scan, M, W; unsigned long long wait_done; int keep_used = 0; };      // keep_used: the used-measurement bytes are cleared by the caller (the scan's initiator may still be reading them)      // wait_done != 0: the scan's ILP launch may still be running -- wait until FCounts::blp_done has reached it      // what changes from scan to scan (everything in CommitArgs repeats with period 2 x ring length)

struct CommitArgs {
    TTable cur, nxt;
    const int32_t* sel; const int32_t* t_status; const int32_t* t_jdrop; const int32_t* t_count; const int32_t* t_firstsurv;
    const int32_t* w_root_scan; const int32_t* w_root_node; const double* w_root_cnllr; const uint8_t* w_root_f32;
    int R; int cap; int Tcap;      // (the scan number and the size of its scan, M / W = ceil(M/64), travel separately: CommitDyn)
    int vnext;            // version index of the table this commit produces: (scan + 1) & 1
    int32_t* new_index;
    FCounts* cnt; DevStatus* status;
    int32_t* cl_counts; const int32_t* cl_status; const int32_t* cl_iters; const int32_t* multi_list;
    unsigned char* used_bytes; unsigned long long* used_words;
    ReportHeader* hdr; mht_target_report* rec;
    const unsigned* vcount;        // value ids handed out by the covariance table of the forest (mht_vtab.h), published in hint[1]
    unsigned long long* hint;      // host-mapped word or null: {scan, targets alive after it}, so that the host can size the next grids
                                   // without fetching a report
    int32_t* log;                  // development: 16 words per scan (ring of 64 scans), read with mht_forest_debug_read("commit_log")
};

// Target side of termination + N-scan pruning (tracker.py:353-381, :1219-1231): compact the target table, move the
// roots, build the next scan's leaf ranges, write the scan report.  One workgroup: everything here is O(targets).
// NT = threads of the workgroup; `sm` = 2 * NT / 64 + 8 ints of LDS scratch (handed in by the kernel: a static __shared__
// here would shift the dynamic LDS base of the kernels this is inlined into off its 16-byte alignment).
// Returns the number of targets alive behind the scan (every thread), -1 for a void scan.
template <int NT, typename CARGS>
__device__ __forceinline__ int commit_body(const CARGS& a, const CommitDyn dyn, int* sm) {
    constexpr int PRUNE_THREADS = NT;
    int* s_scan = sm;
    int* s_scan2 = sm + NT / 64;
    int& s_total = sm[2 * (NT / 64)];
    int& s_total2 = sm[2 * (NT / 64) + 1];
    int& s_branched = sm[2 * (NT / 64) + 2];
    int& s_limit = sm[2 * (NT / 64) + 3];
    int& s_itmax = sm[2 * (NT / 64) + 4];
    const int tid = threadIdx.x;
    if (dyn.wait_done) {
        // this grow launch overlaps the scan's ILP launch: its workgroups count themselves off behind an agent-scope release of what they
        // wrote; every wavefront waits for the last of them, then drops what its CU may have cached
        unsigned long long v;
        const bool ok = spin_until(&a.cnt->blp_done, [&](unsigned long long x) { return x >= dyn.wait_done; }, v);
        if (!ok && tid == 0) { a.status->overflow = 2; atomicOr(&a.status->pad[0], 1 << 6); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // first round trip, everything at once: the scalars and the first chunk of per-target look-ups (index clamped by the
    // table's capacity; entries beyond the real count are masked afterwards)
    const int s_over = a.status->overflow, c_over = a.cnt->overflow, nT = a.cnt->nT, nCh = a.status->n_children;
    const int nC = a.cl_counts[0], n_ilp = a.cl_counts[1], e_over = a.cl_counts[3];
    const int tcl = tid < a.Tcap ? tid : 0;
    int v_st = a.t_status[tcl], v_cnt = a.t_count[tcl], v_j = a.t_jdrop[tcl], v_first = a.t_firstsurv[tcl];
    int v_id = a.cur.id[tcl], v_win = a.cur.window[tcl], v_dep = a.cur.depth[tcl];
    int v_rs = a.w_root_scan[tcl], v_rn = a.w_root_node[tcl];
    double v_rc = a.w_root_cnllr[tcl];
    uint8_t v_rf = a.w_root_f32[tcl];
    // (the ILP statistics' first batch of look-ups goes out with the above and its dependent one behind it, instead of as two more
    // round trips at the end of the workgroup's critical path; entries of multi_list beyond this scan's count are stale but valid ids)
    const int c_first = a.multi_list[tcl];
    // (the compiler sinks the per-target look-ups above behind the branch below -- a second dependent round trip behind the overlapping
    // launch's wait, seen in the ISA -- unless something that may touch memory stands between them and the branch)
    asm volatile("" ::: "memory");
    if (s_over || c_over) {        // void scan: report the error, leave the forest alone (it must be recreated)
        if (tid == 0) {
            ReportHeader& h = *a.hdr;
            h.scan = dyn.scan; h.n_targets = 0; h.n_alive = 0; h.n_leaves_in = a.cur.leaf_off[nT];
            h.n_children = nCh; h.n_leaves_out = 0; h.n_clusters = 0; h.n_ilp = 0; h.n_branched = 0; h.n_limit = 0;
            h.blp_iters_max = 0; h.error = (s_over == 2) ? MHT_E_HIP : MHT_E_CAPACITY; h.used_words = 0;
            h.t_process = h.t_cluster = h.t_optim = h.t_scan = 0;
            a.cnt->overflow = 1;
        }
        return -1;
    }
    const int L_in = a.cur.leaf_off[nT] - a.status->n_dead;      // (only needed at the very end; slots minus what similar-state pruning emptied)
    if (tid == 0) { s_branched = 0; s_limit = 0; s_itmax = 0; }
    const int cf = (c_first >= 0 && c_first < a.Tcap) ? c_first : 0;
    const int st_first = a.cl_status[cf], it_first = a.cl_iters[cf];
    int running = 0, lrun = 0;
    for (int base = 0; base < nT; base += PRUNE_THREADS) {
        const int t = base + tid;
        const bool in = t < nT;
        if (base > 0) {                 // further chunks (more than 512 targets): clamped, the look-ups go out together
            const int tc = in ? t : 0;
            v_st = a.t_status[tc]; v_cnt = a.t_count[tc]; v_j = a.t_jdrop[tc]; v_first = a.t_firstsurv[tc];
            v_id = a.cur.id[tc]; v_win = a.cur.window[tc]; v_dep = a.cur.depth[tc];
            v_rs = a.w_root_scan[tc]; v_rn = a.w_root_node[tc]; v_rc = a.w_root_cnllr[tc]; v_rf = a.w_root_f32[tc];
        }
        const int al = in && (v_st == 0);
        const int cntl = v_cnt, j = v_j, first = v_first, id = v_id, win = v_win, dep = v_dep, rs = v_rs, rn = v_rn;
        const double rc = v_rc;
        const uint8_t rf = v_rf;
        const int leaves = al ? cntl : 0;
        // one block scan for both the compacted target index and the leaf offset
        const int lane = tid & 63, wv = tid >> 6;
        int incl = al, incl2 = leaves;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o), u2 = __shfl_up(incl2, o);
            if (lane >= o) { incl += u; incl2 += u2; }
        }
        if (lane == 63) { s_scan[wv] = incl; s_scan2[wv] = incl2; }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, acc2 = 0;
            for (int i = 0; i < PRUNE_THREADS / 64; ++i) {
                const int v = s_scan[i], v2 = s_scan2[i];
                s_scan[i] = acc; s_scan2[i] = acc2;
                acc += v; acc2 += v2;
            }
            s_total = acc; s_total2 = acc2;
        }
        __syncthreads();
        const int pos = running + s_scan[wv] + incl - al, lpos = lrun + s_scan2[wv] + incl2 - leaves;
        running += s_total;
        lrun += s_total2;
        if (in) {
            mht_target_report& r = a.rec[t];
            r.new_index = al ? pos : -1;
            r.n_leaves = leaves;
            __hip_atomic_store(&a.new_index[t], al ? pos : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (written through: read by the launch's other workgroups, see ni_flag)
            if (al) {
                a.nxt.id[pos] = id;
                a.nxt.window[pos] = (j > 0) ? ((win & 0xff) | (WIN_REBUILT_ALL << WIN_REBUILT_SHIFT)) : win;      // (the root advanced: the reference rebuilds the target's association set from its tree, tracker.py:1222-1227)
                a.nxt.depth[pos] = dep + 1 - j;
                a.nxt.shift[pos] = j;
                a.nxt.root_scan[pos] = rs;
                a.nxt.root_node[pos] = rn;
                a.nxt.root_cnllr[pos] = rc;
                a.nxt.root_f32[pos] = rf;
                a.nxt.first[pos] = first;
                a.nxt.leaf_off[pos] = lpos;
            }
        }
        if (base + PRUNE_THREADS < nT) __syncthreads();      // s_scan is re-used by the next chunk
    }
    const int nAlive = running, Lnext = lrun;
    if (dyn.wait_done) {
        // the compacted indices are what the launch's target workgroups are waiting for (tchild / tcend and the union-find are indexed by
        // them): posted as soon as they are out, the rest of the commit follows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&a.cnt->ni_flag, (unsigned long long)(unsigned)dyn.scan | ((unsigned long long)(nAlive != nT ? 1 : 0) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ILP statistics
    if (tid < n_ilp) {                                      // only this scan's ILPs: the entries of other clusters are stale
        if (st_first == MHT_BLP_BRANCHED) atomicAdd(&s_branched, 1);
        if (st_first == MHT_BLP_NODE_LIMIT) atomicAdd(&s_limit, 1);
        if (st_first) atomicMax(&s_itmax, it_first);
    }
    for (int i = tid + PRUNE_THREADS; i < n_ilp; i += PRUNE_THREADS) {
        const int c = a.multi_list[i];
        const int st = a.cl_status[c];
        if (st == MHT_BLP_BRANCHED) atomicAdd(&s_branched, 1);
        if (st == MHT_BLP_NODE_LIMIT) atomicAdd(&s_limit, 1);
        if (st) atomicMax(&s_itmax, a.cl_iters[c]);
    }
    // used-measurement bytes -> bit mask of the report; bytes cleared for the next scan
    for (int base = 0; base < dyn.W * 64; base += PRUNE_THREADS) {
        const int jm = base + tid;
        const int u = (jm < dyn.M) ? a.used_bytes[jm] : 0;
        if (u && !dyn.keep_used) a.used_bytes[jm] = 0;
        const unsigned long long bits = __ballot(u != 0);
        if ((tid & 63) == 0 && jm < dyn.W * 64) a.used_words[jm >> 6] = bits;
    }
    __syncthreads();
    if (tid == 0) {
        a.nxt.leaf_off[nAlive] = Lnext;
        ReportHeader& h = *a.hdr;
        h.scan = dyn.scan;
        h.n_targets = nT;
        h.n_alive = nAlive;
        h.n_leaves_in = L_in;
        h.n_children = nCh;
        h.n_leaves_out = Lnext;
        h.n_clusters = nC;
        h.n_ilp = n_ilp;
        h.n_branched = s_branched;
        h.n_limit = s_limit;
        h.blp_iters_max = s_itmax;
        h.error = e_over ? MHT_E_CAPACITY : 0;
        h.used_words = dyn.W;
        h.n_births = 0;
        {   // per-stage device times (tracker.py:192-294 keeps toc['Process'], ['Cluster'], ['Optim'] per scan): from the stamps of the launches
            const unsigned long long t0 = a.status->t[0], t1 = a.status->t[1], t2 = a.status->t[2], t3 = a.status->t[3], t4 = a.status->t[4];
            const unsigned long long to = (t3 && t3 < t2) ? t3 : t2;      // the optimisation stage starts with similar-state pruning when it ran
            auto ticks = [](unsigned long long b, unsigned long long e) { return (e > b && e - b < 0x7fffffffull) ? (int32_t)(e - b) : 0; };
            h.t_process = ticks(t0, t1); h.t_cluster = ticks(t1, to); h.t_optim = ticks(to, t4); h.t_scan = ticks(t0, t4);
        }
        if (a.log) {
            int32_t* g = a.log + (dyn.scan & 63) * 16;
            g[0] = dyn.scan; g[1] = nT; g[2] = nAlive; g[3] = L_in; g[4] = nCh; g[5] = Lnext; g[6] = nC; g[7] = n_ilp;
            g[8] = s_branched; g[9] = s_limit; g[10] = s_itmax; g[11] = e_over; g[12] = a.cl_counts[2]; g[13] = a.cl_counts[5];
            g[14] = a.status->n_dead; g[15] = dyn.M;
        }
        a.cnt->L_in = L_in;
        a.cnt->n_children = nCh;
        a.cnt->nT = nAlive;
        a.cnt->nTv[a.vnext] = nAlive;
        a.cnt->L = Lnext;
        if (a.hint) __hip_atomic_store(a.hint, ((unsigned long long)(unsigned)dyn.scan << 32) | (unsigned)nAlive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.hint && a.vcount) __hip_atomic_store(a.hint + 1, (unsigned long long)*a.vcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // fill of the value table
        // (the scan's cluster statistics, for the size of the next ILP launches: scan | multi-target clusters | single-target clusters | team-sized ones)
        if (a.hint) __hip_atomic_store(a.hint + 2, ((unsigned long long)((unsigned)dyn.scan & 0xffffu) << 48) | ((unsigned long long)((unsigned)n_ilp & 0xffffu) << 32) |
                                                   ((unsigned long long)((unsigned)a.cl_counts[2] & 0xffffffu) << 8) | (unsigned long long)((unsigned)a.cl_counts[5] & 0xffu),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.cnt->n_nodes = nCh;
        a.cnt->n_roots = 0;        // roots born after this scan go to the end of the layer: node root_base + n_roots
        // the counters of this scan's status word have been read: zero for the scan after the next (its grow launch may start before the
        // next scan's ILP launch has ended, so nobody else can do it; the overflow flag stays -- a void scan kills the forest)
        a.status->n_children = 0; a.status->n_dead = 0;
        // (the per-scan status word -- one of two, by scan parity -- is cleared by the cluster kernel of the next scan: a
        // commit that rides in the next grow_kernel must not touch what that kernel's tiles are reading)
    }
    return nAlive;
}

}  // namespace mht
