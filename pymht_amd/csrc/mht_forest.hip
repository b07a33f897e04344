// The device-resident hypothesis forest: Tracker.addMeasurementList (pymht/tracker.py:162-307) without host
// round trips between the stages.
//
// Data layout in HBM (all structure-of-arrays, owned by the ctx):
//   layers[R]      ring of mht_nodes, one per scan of the N-scan window and two to spare (R = N+4, RING_EXTRA below: a host that folds
//                  the report of scan k while scan k+2 is already queued still finds all N+2 levels above a leaf of scan k): layer s % R holds every
//                  hypothesis created at scan s (children of that scan, then roots born after it).  A node refers
//                  to its parent by index into the previous layer (pyTarget.Target.parent).
//   leaves         implicit: target t owns the nodes first[t] .. first[t]+count-1 of the newest layer (contiguous, in
//                  pyTarget.getLeafNodes DFS order); leaf_off[] is the prefix of the counts; targets in __targetList__ order.
//   path[PD][.]    per newest-layer node the measurement nodes (ring_slot*Mpad + m) on its root->node path:
//                  the rows of its ILP column (tracker.py:1042-1113) and the key for N-scan pruning.
//   target table   id, window, depth below the root, root node/score -- double buffered, compacted on termination.
//   assoc[T][AW]   bitsets = the reference's __associatedMeasurements__ (tracker.py:78), rebuilt every scan.
// Per scan three launches on one stream, no memsets, no host round trip: grow (mht_gate.hip), cluster (mht_cluster.hip),
// blp (mht_blp.hip: selection + per target the termination test tracker.py:891-916, the N-scan prune decision
// pyTarget.py:343-356, the new root, the report record and the surviving leaf range).  The target-side commit
// (mht_commit.h: compaction of the target table tracker.py:353-381 / :1219-1231, next scan's leaf ranges, the scan report)
// is deferred: it rides in workgroup 0 of the next scan's grow launch, or runs as commit_kernel when the host asks for the
// committed state first (report, births, exports) -- see Forest::commit_pending.
#include "mht_kernels.h"
#include "mht_commit.h"
#include <hip/hip_ext.h>
#include "mht_admit.h"
#include "mht_init_dev.h"
#include <string.h>
#include <math.h>
#include <new>
#include <stdlib.h>

static void hp_print();      // (development, see forest_step_impl)
namespace mht {

constexpr int MAXR = 16;
// Layers of the node ring beyond the N-scan window.  A leaf's chain back to its root spans N + 2 layers; the streaming drop-in path folds
// the report of scan k while scan k + 2 is queued (tracker.py here: _queue_report), and a track that DIED in scan k keeps its root where
// it was -- its history is fetched at that fold (mht_forest_chain), so the layer of scan k - 1 - N must still be there when scan k + 2
// has been written: N + 4 layers (round 3 had N + 3: the root and the committed history of tracks that died mid-stream were lost).
constexpr int RING_EXTRA = 4;
constexpr int EV_POOL = 64;
constexpr int Z_RING = 8;           // pinned staging buffers of mht_forest_step_host (a consumer guard every Z_GUARD scans, see step_host_impl)
constexpr int Z_GUARD = 4;
constexpr int BIRTH_CAP = 256;      // candidates of the device initiator per scan that the report can hold

struct LayerView { const double* x; const double* cnllr; const int32_t* parent; const int32_t* meas; const uint8_t* flags; const int32_t* cov; const float* P; };

// The scan report goes to the host from inside the kernel that completes it: the workgroup copies header, used-measurement
// mask, birth records and the target rows from the device block into pinned, device-mapped host memory with 16-byte stores
// (posted PCIe writes: ~70 KB at the headline size).  A copy engine / blit kernel on the stream costs more than the copy: the
// host issues two more calls per scan and the stream idles ~10 us in front of every engine switch.
template <int NT> __device__ __forceinline__ void publish_report(const PublishArgs& p) {
    if (!p.dst) return;
    __threadfence();
    __syncthreads();
    const ReportHeader* h = reinterpret_cast<const ReportHeader*>(p.src);
    const int n_births = h->n_births, nT = h->n_targets;
    const uint4* s4 = reinterpret_cast<const uint4*>(p.src);
    uint4* d4 = reinterpret_cast<uint4*>(p.dst);
    const int head = (p.birth_off + n_births * (int)sizeof(mht_birth_report) + 15) / 16;      // header + mask + births present
    for (int i = threadIdx.x; i < head; i += NT) d4[i] = s4[i];
    const int r0 = p.rec_off / 16, rn = (nT * (int)sizeof(mht_target_report) + 15) / 16;
    for (int i = threadIdx.x; i < rn; i += NT) d4[r0 + i] = s4[r0 + i];
    __threadfence_system();
}
static_assert(sizeof(mht_target_report) % 16 == 0 && sizeof(mht_birth_report) % 8 == 0, "report blocks are copied in 16-byte pieces");

// the scan from the pinned staging ring (device-mapped) to its device buffer: one small workgroup on the stream, no copy engine
// (flag != null: the consumers on the ctx stream do not wait for this launch through an event -- they poll the word: the scan is written
// through, acknowledged, then the tag is posted)
__global__ void stage_scan_kernel(const float4* src, float4* dst, int n16, unsigned long long* flag = nullptr, unsigned long long tag = 0) {
    if (!flag) { for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i]; return; }
    for (int i = threadIdx.x; i < n16; i += blockDim.x) {
        const float4 v = src[i];
        unsigned long long* q = reinterpret_cast<unsigned long long*>(dst + i);
        __hip_atomic_store(q, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(COMMIT_THREADS) void commit_publish_kernel(const CommitArgs a, const CommitDyn dyn, const PublishArgs pub) {
    __shared__ int s_commit[2 * (COMMIT_THREADS / 64) + 8];
    commit_body<COMMIT_THREADS>(a, dyn, s_commit);
    publish_report<COMMIT_THREADS>(pub);
}

__global__ __launch_bounds__(COMMIT_THREADS) void commit_kernel(const CommitArgs a, const CommitDyn dyn) {
    __shared__ int s_commit[2 * (COMMIT_THREADS / 64) + 8];
    commit_body<COMMIT_THREADS>(a, dyn, s_commit);
}

__global__ __launch_bounds__(1024) void add_targets_kernel(const AddArgs a) {
    __shared__ int s_adm[ADM_LDS_INTS];
    add_targets_body<1024>(a, s_adm);
}

// The end of a scan in the drop-in API path, ONE launch of one workgroup instead of three: the scan's commit (target table, report),
// step 7 on the measurements the commit found unused (initiator_body), Tracker.initiateTarget for what it confirmed.
static_assert(INIT_THREADS == 1024, "post_scan_kernel runs commit, initiator and admission with one block size");
// Which of the scan's AIS messages a track took (tracker.py:267-270: the identities in an association set behind the scan's termination
// and N-scan pruning): a target whose root moved has its set rebuilt from the surviving tree, one whose root did not move has lost
// nothing -- either way: the identities of the SURVIVING leaves of the targets that are still alive.  Runs behind the commit.
struct AisUsedArgs { const int32_t* mmsi; const int32_t* first; const int32_t* leaf_off; const FCounts* cnt; const AisInitMsg* msgs; int nA; unsigned char* used; };
static __device__ void ais_used_body(const AisUsedArgs& u) {
    const int tid = threadIdx.x;
    for (int q = tid; q < u.nA; q += 1024) u.used[q] = 0;
    __threadfence_block();
    __syncthreads();
    const int nT = u.cnt->nT;
    for (int t = tid; t < nT; t += 1024) {
        const int f0 = u.first[t], n = u.leaf_off[t + 1] - u.leaf_off[t];
        int last = 0;      // (the leaves of a track carry one identity, or a few: look each up once)
        for (int i = 0; i < n; ++i) {
            const int mm = u.mmsi[f0 + i];
            if (mm == 0 || mm == last) continue;
            last = mm;
            for (int q = 0; q < u.nA; ++q) if (u.msgs[q].mmsi == mm) { u.used[q] = 1; break; }
        }
    }
    __threadfence_block();
    __syncthreads();
}
template <bool AIS>
__global__ __launch_bounds__(1024) void post_scan_kernel(const CommitArgs cm, const CommitDyn dyn, const InitArgs in, const AddArgs ad, const int do_commit,
                                                         const PublishArgs pub, const int run_init, const AisUsedArgs au) {
    __shared__ int s_commit[2 * (1024 / 64) + 8];
    __shared__ int s_adm[ADM_LDS_INTS];
    if (do_commit) {
        commit_body<1024>(cm, dyn, s_commit);
        __threadfence_block();
        __syncthreads();
    }
    if (!cm.hdr->error) {      // (void scan: nothing to initiate)
        if (run_init) {        // (streaming: the initiator already ran next to the scan's clustering, cluster_init_kernel)
            if (AIS && au.nA > 0) ais_used_body(au);
            initiator_body<AIS>(in);
            __threadfence_block();
            __syncthreads();
        }
        add_targets_body<1024>(ad, s_adm);
    }
    publish_report<1024>(pub);
}

// The device initiator as a launch of its own, on the forest's side stream: it needs the scan and the used-measurement bytes of the
// scan's grow launch, nothing of the clustering or the ILPs, and what it gives birth to is admitted in the NEXT scan's grow launch --
// so it runs next to the scan's cluster and ILP launches instead of lengthening the cluster launch (cluster_init_kernel: 15 us against
// 8.7 for the clustering alone).
__global__ __launch_bounds__(INIT_THREADS) void initiator_side_kernel(const InitArgs in, const DevStatus* status, const int32_t* sticky_overflow, unsigned long long* done_flag = nullptr,
                                                                      const unsigned long long* z_flag = nullptr, unsigned long long z_tag = 0,
                                                                      unsigned long long* tick = nullptr, const unsigned long long* begun = nullptr, unsigned long long* dbg = nullptr) {
    if (tick) {      // launched with one workgroup per XCD: the first one to start (the XCD the ILP launch drained first) is the initiator
        __shared__ unsigned s_r;
        if (threadIdx.x == 0) s_r = first_come_ticket(tick, (unsigned)in.scan_no);
        __syncthreads();
        if (s_r != 0u) return;
    }
    // A wait that gives up (spin_until: ~2 s) voids the scan like the commit's and the target workgroups' waits do: the initiator must not
    // read a scan that is not staged or the used-measurement bytes of a grow launch that has not finished.  The sticky error word is set
    // (the forest is dead, every later call reports MHT_E_CAPACITY-class failure) and nothing is initiated.
    __shared__ int s_gave_up;
    if (threadIdx.x == 0) s_gave_up = 0;
    __syncthreads();
    if (z_tag) {      // (the scan's staging, see stage_scan_kernel)
        unsigned long long v;
        if (!spin_until(z_flag, [&](unsigned long long x) { return x >= z_tag; }, v)) {
            s_gave_up = 1;
            if (threadIdx.x == 0 && dbg) { dbg[0] = 1; dbg[1] = v; dbg[2] = z_tag; dbg[3] = (unsigned)in.scan_no; }
        }
    }
    if (begun) {      // on a queue of its own: nothing orders this launch behind the scan's grow launch but the word the scan's ILP launch posts when it starts
        unsigned long long v;
        if (!spin_until(begun, [&](unsigned long long x) { return x >= (unsigned long long)(unsigned)in.scan_no; }, v)) {
            s_gave_up = 1;
            if (threadIdx.x == 0 && dbg) { dbg[0] = 2; dbg[1] = v; dbg[2] = (unsigned)in.scan_no; dbg[3] = wall_clock64(); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const bool gave_up = s_gave_up != 0;
    if (gave_up && threadIdx.x == 0 && sticky_overflow) *const_cast<int32_t*>(sticky_overflow) = 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_gnn[INIT_GNN_LDS];      // (8 KB with the rest: what a 155 KB workgroup of the ILP launch leaves of a CU's LDS)
    if (!gave_up && !((status && status->overflow) || (sticky_overflow && *sticky_overflow))) initiator_body<false>(in, s_gnn, INIT_GNN_LDS);      // (void scan: nothing is initiated)
    if (done_flag) {      // the next scan's grow launch may be running already: its admission waits for this word (FCounts::init_flag)
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(done_flag, (unsigned long long)(unsigned)in.scan_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// constant-turn forest (mht_kernels.h: CtGrow): per ring layer the covariances of its nodes' children and of the roots born into it
struct CtLayers { const float* Pbar[MAXR]; const float* Phat[MAXR]; const float* Proot[MAXR]; int on; };
__device__ __forceinline__ void export_cov_ct(const CtLayers& c, int li, int lprev, int key, float* P, double* P64) {
    const float* src = key < 0 ? c.Proot[li] + (size_t)(-2 - key) * NP : ((key & 1) ? c.Phat[lprev] : c.Pbar[lprev]) + (size_t)(key >> 1) * NP;
    for (int e = 0; e < NP; ++e) { if (P64) P64[e] = (double)src[e]; if (P) P[e] = src[e]; }
}
struct LeavesArgs {
    mht_nodes layer; TTable tab; const FCounts* cnt; VTab vt;
    int capacity; double* x; float* P; double* cnllr; int32_t* meas; int32_t* target; int32_t* id; int32_t* node; uint8_t* flags;
    double* P64;      // != null (mht_forest_leaves_f64): every covariance as float64 -- exact for the float32 ones, the reference's own for F_COV_F64 leaves
};
struct LeavesCt { CtLayers c; int li, lprev; };
// the covariance of node nd of a layer whose keys belong to table v, as float32 (P) or float64 (P64)
__device__ __forceinline__ void export_cov(const VTab& v, int key, uint8_t fl, float* P, double* P64) {
    const int id = v.child[key];
    if (fl & F_COV_F64) {
        double t[NP];
        vt_load64(v, id, t);
        for (int e = 0; e < NP; ++e) { if (P64) P64[e] = t[e]; if (P) P[e] = (float)t[e]; }
    } else {
        float t[NP];
        vt_load(v, id, t);
        for (int e = 0; e < NP; ++e) { if (P64) P64[e] = (double)t[e]; if (P) P[e] = t[e]; }
    }
}
__global__ void leaves_kernel(const LeavesArgs a, const LeavesCt ct) {
    const int nT = a.cnt->nT;
    const int L = a.cnt->L < a.capacity ? a.cnt->L : a.capacity;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
        int lo = 0, hi = nT;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.tab.leaf_off[mid] <= i) lo = mid; else hi = mid; }
        const int t = lo, nd = a.tab.first[lo] + (i - a.tab.leaf_off[lo]);
        for (int k = 0; k < NX; ++k) a.x[i * NX + k] = a.layer.x[(size_t)k * a.layer.cap + nd];
        if (ct.c.on) export_cov_ct(ct.c, ct.li, ct.lprev, a.layer.cov[nd], a.P ? a.P + (size_t)i * NP : nullptr, a.P64 ? a.P64 + (size_t)i * NP : nullptr);
        else export_cov(a.vt, a.layer.cov[nd], a.layer.flags[nd], a.P ? a.P + (size_t)i * NP : nullptr, a.P64 ? a.P64 + (size_t)i * NP : nullptr);
        a.cnllr[i] = a.layer.cnllr[nd];
        a.meas[i] = a.layer.meas[nd];
        a.target[i] = t;
        a.id[i] = a.tab.id[t];
        a.node[i] = nd;
        a.flags[i] = a.layer.flags[nd];
    }
}

struct ChainArgs { mht_nodes layers[MAXR]; VTab vt[2]; int lgen[MAXR]; int R; int scan, node, max_len; int32_t* nodes; int32_t* meas; double* x; double* cnllr; float* P; int32_t* n_out; double* P64; uint8_t* flags; };
__device__ __forceinline__ int chain_walk(const ChainArgs& a, const CtLayers& ct, int nd, int32_t* nodes, int32_t* meas, double* x, double* cnllr, float* P, double* P64, uint8_t* flags) {
    int sc = a.scan, n = 0;
    while (nd >= 0 && n < a.max_len && sc >= 0) {
        const mht_nodes& l = a.layers[sc % a.R];
        nodes[n] = nd;
        meas[n] = l.meas[nd];
        cnllr[n] = l.cnllr[nd];
        for (int k = 0; k < NX; ++k) x[n * NX + k] = l.x[(size_t)k * l.cap + nd];
        const VTab& v = a.vt[a.lgen[sc % a.R]];      // (the generation of the value table this layer's keys belong to)
        if (ct.on) export_cov_ct(ct, sc % a.R, (sc + a.R - 1) % a.R, l.cov[nd], P ? P + (size_t)n * NP : nullptr, P64 ? P64 + (size_t)n * NP : nullptr);
        else export_cov(v, l.cov[nd], l.flags[nd], P ? P + (size_t)n * NP : nullptr, P64 ? P64 + (size_t)n * NP : nullptr);
        if (flags) flags[n] = l.flags[nd];
        ++n;
        nd = l.parent[nd];
        --sc;
    }
    return n;
}
__global__ void chain_kernel(const ChainArgs a, const CtLayers ct) {
    if (threadIdx.x || blockIdx.x) return;
    *a.n_out = chain_walk(a, ct, a.node, a.nodes, a.meas, a.x, a.cnllr, a.P, a.P64, a.flags);
}
// A batch of chains (mht_forest_chains_begin): one thread per chain, every chain a record of its own in a block of host-mapped pinned memory
// (the start nodes are read from its head) -- nothing is copied, nobody waits; layout of a record: ChainRec
struct ChainRec { size_t o_m, o_x, o_c, o_P, o_fl, stride; };
__host__ __device__ __forceinline__ ChainRec chain_rec(int len, int PB) {
    ChainRec r;
    r.o_m = 8 + (size_t)len * 4;                                  // [0] n, [8] nodes
    r.o_x = (r.o_m + (size_t)len * 4 + 7) & ~(size_t)7;
    r.o_c = r.o_x + (size_t)len * (NX * 8);
    r.o_P = r.o_c + (size_t)len * 8;
    r.o_fl = r.o_P + (size_t)len * ((size_t)NP * PB);
    r.stride = (r.o_fl + (size_t)len + 15) & ~(size_t)15;
    return r;
}
__global__ void chains_kernel(const ChainArgs a, const CtLayers ct, const int32_t* start, int count, char* out, int PB) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const ChainRec r = chain_rec(a.max_len, PB);
    char* o = out + (size_t)i * r.stride;
    const int n = chain_walk(a, ct, start[i], (int32_t*)(o + 8), (int32_t*)(o + r.o_m), (double*)(o + r.o_x), (double*)(o + r.o_c), PB == 4 ? (float*)(o + r.o_P) : nullptr,
                             PB == 8 ? (double*)(o + r.o_P) : nullptr, (uint8_t*)(o + r.o_fl));
    *(int32_t*)o = n;
}

// ------------------------------------------------------------------------------------------------------------
struct Arena {
    char* base = nullptr; size_t size = 0, off = 0;
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += n * sizeof(T);
        return p;
    }
};

struct Forest {
    mht_forest_config cfg;
    mht_model model;
    int Tcap, Ncap, Mpad, R, PD, AW, n_mnodes, Ecap, SegCap;
    VTab vt = {};                     // covariances by value, shared by all targets and scans (mht_vtab.h): the CURRENT generation
    // Value ids are never recycled within a generation.  When the table is three quarters full the live leaves are re-keyed into the
    // other generation's (empty) table -- vt_rebuild_kernel, between two scans -- and the forest goes on there; the layers of the
    // ring that were written before keep their keys into the old one (layer_gen), which is cleared when the ring has come round.
    VTab vts[2] = {}; int vgen = 0; int layer_gen[MAXR] = {}; int last_rebuild_scan = -1000000; int32_t* vt_remap = nullptr;
    int rebuilds = 0;
    unsigned long long vt_used_seen = 0, vt_rate = 0;      // ids in use at the last commit the host saw; largest per-commit consumption seen (a burst predicts the next)
    Arena arena;
    mht_nodes layer[MAXR];
    int32_t* path[2]; int32_t* apath[2]; double* cost2[2]; int32_t* tchild; int32_t* tcend;      // cost2: ILP costs of the newest layer's nodes, by scan parity (the next scan's grow launch may overlap this scan's ILP launch)
    int pds = 8;                      // ints per path / ancestor record (8 or 16)
    // AIS forest (mht_forest_create_ex, MHT_FOREST_AIS; mht_kernels.h: AisGrow): identities per node, record pool of the fused children,
    // the messages of the next scan (mht_forest_set_ais arms them, the next step consumes them)
    // constant-turn forest (six-state build, MHT_FOREST_CT; mht_kernels.h: CtGrow): per ring layer the covariances of its nodes' children and of
    // the roots born into it; for the newest layer's leaves the predictions and gains forest_ct_kernel leaves for the grow launch
    bool ct = false; double ct_T = 0.0;
    float* ct_Pbar[MAXR] = {}; float* ct_Phat[MAXR] = {}; float* ct_Proot[MAXR] = {};
    float4* ct_gains = nullptr; double* ct_xbar = nullptr; double* ct_zhat = nullptr; unsigned long long* ct_hw_spill = nullptr;
    bool ais = false; int ais_half = 0;
    int32_t* l_mmsi[MAXR] = {}; int32_t* l_hmmsi[MAXR] = {};
    int32_t* ais_nf = nullptr; int32_t* ais_off = nullptr; AisRec* ais_rec = nullptr; int ais_rec_cap = 0; unsigned* ais_count = nullptr;
    char* ais_groups_dev = nullptr; char* ais_msgs_dev = nullptr; int ais_group_cap = 64, ais_msg_cap = 0;
    bool ais_armed = false; int ais_nG = 0, ais_nA = 0; double ais_eta2 = 0.0, ais_lambda = 0.0;
    unsigned* alloc2[2]; int block_cap = 0, over_base = 0, region_cap = 0, root_base = 0;   // child counters of the regions of the node index space
    TTable tab[2];
    unsigned char* used_bytes[2];     // by scan parity: the commit of scan s may run while scan s+1 is marking its own bytes
    DevStatus* status2;               // [2] per-scan status words, by scan parity (same reason)
    unsigned long long* init_dbg;     // [8] initiator_side_kernel: which wait gave up (development)
    unsigned* edges; int32_t* edge_count;
    int32_t *edge_t, *edge_m, *t_label, *t_cluster, *cl_ptr, *cl_members, *multi_list, *single_list, *cl_counts, *big_list;
    int32_t* cl_owner;      // [Tcap] cluster-sharded step: device of every multi-target cluster (LPT by column count)
    // clustering inside the grow launch (mht_kernels.h: FDyn::uf_epoch): owner word per measurement node, parent word per target; uf_ok:
    // the ILP launch's workgroups can derive the cluster tables themselves (MHT_NO_UF=1: the clustering kernel on every scan, as before)
    unsigned long long* uf_owner = nullptr; unsigned long long* uf_parent2[2] = {nullptr, nullptr}; bool uf_ok = false; int uf_scans = 0;
    // overlap of a scan's ILP launch with the next scan's grow launch (mht_kernels.h: TGT_REC_*, FDyn::ovl): the per-target records, the
    // scan whose ILP launch published them, the total its workgroups will have counted off (FCounts::blp_done), launches made any-order
    unsigned long long z_tag_step = 0;      // != 0: the scan being stepped was staged without an event wait on the ctx stream (FDyn::z_tag); z_wait_slot: its slot
    int z_wait_slot = -1;
    int init_flag_scan = 0;      // last scan whose initiator posts FCounts::init_flag
    unsigned long long* rec0 = nullptr; int pub_scan = 0; unsigned long long blp_done_total = 0; bool ovl_ok = true; int ovl_launches = 0;
    int32_t* cl_gtab = nullptr; bool cluster_big = false;      // the clustering tables in HBM when they do not fit LDS (mht_cluster.hip: cluster_big_kernel)
    int32_t* team_list; TeamState* team_state2[2]; TeamResult* team_res; bool teams = true;      // (team_state2: by scan parity)      // branch-and-bound teams (mht_blp.hip); MHT_BLP_NO_TEAMS=1: off
    double* u; int32_t* usage; int32_t* mark;
    int32_t *best_h, *bb_ch, *bb_best, *bb_last_idx; double *best_rc, *bb_cost, *bb_uused, *bb_last_rc, *bb_rest, *bb_min;
    int32_t *sel, *cl_status, *cl_iters, *cl_nodes, *cl_time; unsigned long long* grow_dbg; int32_t* commit_log; double* bb_snap; int32_t* bb_busy; int bb_snap_rows = 0;
    int32_t *t_status, *t_jdrop, *t_count, *t_firstsurv, *new_index, *near; double* t_score;
    int32_t *w_root_scan, *w_root_node; double* w_root_cnllr; uint8_t* w_root_f32;
    FCounts* cnt;
    bool rep_by_flag[2] = {false, false}; unsigned long long rep_tag[2] = {0, 0}; bool rep_flag_ok = true;      // the host block's report is complete when its done words carry rep_tag (no event)
    char* report_dev2[2]; char* report_host; size_t report_bytes, rec_off, used_off, birth_off, done_off;      // device report blocks by scan parity
    // no copy engine on the scan's path: a small kernel pulls the scan out of the pinned ring, the kernel that completes the report
    // pushes it into pinned host memory (publish_report)
    char* report_host_dev[2] = {nullptr, nullptr}; float* z_host_dev = nullptr;      // device addresses of the pinned host blocks
    int published_scan = 0;      // scan whose report the device has been told to write into report_host2[scan & 1]
    // streaming API path (mht_forest_scan with an initiator): the report of a scan is pushed to the host by extra workgroups of the
    // NEXT scan's grow launch (fgrow_kernel: publish_part), or by publish_kernel if the host asks for it before there is one
    bool pub_deferred = false; PublishArgs pub_args = {}; int pub_slot = 0;
    int init_ran_scan = 0;       // last scan whose initiator ran inside its cluster launch (mht_forest_scan)
    const float* z_cur = nullptr;
    char* report_host2[2] = {nullptr, nullptr}; hipEvent_t rep_ev[2] = {nullptr, nullptr}; int rep_slot = 0; bool rep_inflight = false; bool rep_started[2] = {false, false};
    int host_block_scan[2] = {0, 0};      // scan whose report the host block holds (or is receiving: rep_ev of the block), 0 = none
    hipStream_t stage_stream = nullptr; bool stage_stream_tried = false;
    // the streamed scans' initiator launches go onto the SIDE stream, each one behind the staging kernel of the scan after its own (it is
    // queued by the next call, or by whoever needs its births first: launch_deferred_init): see forest_step_impl
    bool init_ev_lazy = false; bool init_deferred = false; bool init_side_q = true; bool serial_prof = false;      // serial_prof: no launch may wait for a launch on another queue (the staging goes by event too)
    InitArgs init_def_args; const DevStatus* init_def_status = nullptr; unsigned long long init_def_ztag = 0;
    hipEvent_t grow_ev = nullptr, init_ev = nullptr; bool init_ev_pending = false; bool init_side = false;      // MHT_INIT_SIDE=1: the initiator as a launch of its own on the side stream (default: inside the cluster launch)
    float* z_dev; float* z_host; hipEvent_t z_ev[Z_RING] = {}; bool z_used[Z_RING] = {}; int z_slot = 0;
    hipEvent_t z_guard_ev[2] = {nullptr, nullptr}; long long z_count = 0; int z_guard_due = -1;      // consumer guard of the staging ring (step_host_impl)
    // small staging for add_targets / leaves / chain
    Scratch stage_dev; void* stage_host = nullptr; size_t stage_host_bytes = 0;
    // mht_forest_chains_begin: a ring of host-mapped pinned blocks, one per ticket (a ticket lives until CHAIN_SLOTS later ones were issued)
    static constexpr int CHAIN_SLOTS = 8;
    struct ChainSlot { char* host = nullptr; char* dev = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; long long ticket = -1; int count = 0, len = 0, pb = 4; bool pending = false; };
    ChainSlot chain_slots[CHAIN_SLOTS]; long long chain_next = 0;
    // host-side mirrors
    int scan = 0; int nT_ub = 0; int L_ub = 0; bool report_pending = false; int last_M = 0; bool dead = false;
    int nT_ub_prev = 0;      // ... of the scan before it (grid of a grow launch that carries that scan's commit)
    int nT_ub_step = 0;      // upper bound of the number of targets of the last launched scan (rows of its report)
    int births_since_step = 0;   // candidates added after the last launched scan (they are not in its report)
    // the target-side commit of the last launched scan has not run yet: it rides in the next grow_kernel, or is launched
    // on its own by whoever needs the committed state first (report, births, exports)
    bool commit_pending = false; CommitArgs pending = {}; CommitDyn pending_dyn = {};
    bool ct_spill = false;         // testing: MHT_CT_SPILL=1 at creation -- fgrow_ct_kernel keeps every target's hit masks in the global spill block
    bool grid_by_hint = true;      // MHT_BLP_GRID_HINT=0 at creation: the ILP launch sized by the target count alone, as until round 5
    // streaming drop-in path: the admission of what the scan's initiator gave birth to is pending WITH the commit -- both ride in
    // workgroup 0 of the next scan's grow launch (fgrow_adm_kernel), or run as post_scan_kernel when somebody needs the state first
    bool adm_pending = false; AddArgs adm = {}; bool adm_fuse = true;      // MHT_ADM_FUSE=0: admission in a launch of its own behind every scan
    bool shard_open = false; int shard_plan_s = 0, shard_plan_W = 0, shard_M = 0, shard_xn = 0;      // cluster-sharded step between _begin and _end
    long long blp_time_limit = 0;   // wall-clock budget per ILP in 10 ns ticks, 0 = none (mht_forest_set_blp_time_limit)
    float prune_thr = 0.f;       // similar-state pruning (mht_similar.hip): threshold in metres, 0 = off (mht_forest_set_prune_similar)
    int in_groups = 0;           // mht_group_create snapshots the ILP argument blocks of its members: settings behind them are frozen while > 0
    int similar_ran_scan = -100; // last scan prune_similar_kernel ran on: its children may carry F_DEAD when they are leaves (FDyn::maybe_dead)
    bool force_hbm = false;      // testing: MHT_BLP_FORCE_HBM=1 at creation runs every ILP through the HBM storage policy
    bool no_enum = false;        // testing: MHT_BLP_NO_ENUM=1 at creation: no exact search for small clusters (branch and bound instead)
    // grid sizing without reports: the commit publishes {scan, targets alive} in a host-mapped word; with the births the host issued
    // since that scan this bounds the current target count (targets only disappear otherwise)
    unsigned long long* hint_host = nullptr; unsigned long long* hint_dev = nullptr;
    // births issued behind scan s, before scan s+1 (ring by s % 64): host-driven ones exactly, the device initiator's as an upper
    // bound (its capacity) until the scan's report says how many candidates there were
    int births_after[64] = {}; int births_init_ub[64] = {};
    int init_mreq = 0;      // m_required of the radar-only device initiator behind this forest's scans (0: unknown / AIS messages may start tracks: no bound from the preliminary tracks)
    unsigned long long* bhint_host = nullptr; unsigned long long* bhint_dev = nullptr;      // ring of 64: what the device initiator says it gave birth to (InitArgs::bhint)
    long long births_between(int k, int s) const {      // births issued behind scans k .. s-1
        long long b = 0;
        for (int j = k; j < s; ++j) {
            int ub = births_init_ub[j % 64];
            if (ub > 0 && bhint_host) {      // the initiator of scan j has finished: its candidates are counted
                const unsigned long long h = reinterpret_cast<volatile unsigned long long*>(bhint_host)[j % 64];
                if ((int)(h >> 32) == j) { if ((int)(h & 0xffffu) < ub) ub = (int)(h & 0xffffu); }
                else if (init_mreq > 0) {
                    // still running: a track confirmed in scan j has been updated in m_required scans, one per scan, so it was a preliminary track
                    // behind every scan j - m_required .. j - 1 -- the count the initiator posted for any of them bounds scan j's births (typically
                    // a few dozen against a capacity of 256: that many fewer idle target workgroups in the next grow launch)
                    for (int q = j - 1; q >= j - init_mreq && q >= 1; --q) {
                        const unsigned long long hq = reinterpret_cast<volatile unsigned long long*>(bhint_host)[q % 64];
                        if ((int)(hq >> 32) == q && (int)((hq >> 16) & 0xffffu) < ub) ub = (int)((hq >> 16) & 0xffffu);
                    }
                }
            }
            b += births_after[j % 64] + ub;
        }
        return b;
    }
    int targets_ub(int s_table) const {      // upper bound of the targets in the table scan `s_table` runs on
        int ub = nT_ub;
        if (hint_host) {
            const unsigned long long h = *reinterpret_cast<volatile unsigned long long*>(hint_host);
            const int k = (int)(h >> 32), na = (int)(h & 0xffffffffu);
            if (k >= 1 && k < s_table && s_table - k < 60) {
                const long long b = na + births_between(k, s_table);
                if (b < ub) ub = (int)b;
            }
        }
        return ub;
    }
    bool debug = false;      // MHT_GROW_DEBUG set at creation: phase stamps (with -DMHT_GROW_STAMPS), forced storage policies
    bool timing = false; int timed_steps = 0; int ev_slot = 0; hipEvent_t (*evp)[5] = nullptr;   // pool of EV_POOL event sets

    void layout(Arena& ar) {
        for (int s = 0; s < R; ++s) {
            mht_nodes& l = layer[s];
            l.cap = Ncap; l.cap_cov = 0;
            l.x = ar.take<double>((size_t)NX * Ncap); l.cnllr = ar.take<double>(Ncap); l.pd = ar.take<double>(Ncap);
            l.parent = ar.take<int32_t>(Ncap); l.meas = ar.take<int32_t>(Ncap); l.cov = ar.take<int32_t>(Ncap);
            l.flags = ar.take<uint8_t>(Ncap); l.P = nullptr;
        }
        if (ct) {
            for (int s = 0; s < R; ++s) { ct_Pbar[s] = ar.take<float>((size_t)NP * Ncap); ct_Phat[s] = ar.take<float>((size_t)NP * Ncap); ct_Proot[s] = ar.take<float>((size_t)NP * Tcap); }
            ct_gains = ar.take<float4>((size_t)GKQ * Ncap); ct_xbar = ar.take<double>((size_t)NX * Ncap); ct_zhat = ar.take<double>((size_t)2 * Ncap);
            ct_hw_spill = ar.take<unsigned long long>((size_t)Tcap * FG_CAP * (Mpad / 64));      // (fgrow_ct_kernel: full-width hit masks of a target with more candidates than its LDS words hold; touched only then)
        }
        if (ais) {
            for (int s = 0; s < R; ++s) { l_mmsi[s] = ar.take<int32_t>(Ncap); l_hmmsi[s] = ar.take<int32_t>(Ncap); }
            ais_nf = ar.take<int32_t>(Ncap); ais_off = ar.take<int32_t>(Ncap);
            ais_rec_cap = Ncap; ais_rec = ar.take<AisRec>(ais_rec_cap); ais_count = ar.take<unsigned>(16);
            ais_msg_cap = Mpad;
            ais_groups_dev = ar.take<char>((size_t)ais_group_cap * sizeof(mht_ais_group));
            ais_msgs_dev = ar.take<char>((size_t)ais_msg_cap * sizeof(mht_ais_msg));
        }
        for (int b = 0; b < 2; ++b) {
            path[b] = ar.take<int32_t>((size_t)pds * Ncap);
            apath[b] = ar.take<int32_t>((size_t)pds * Ncap);
            TTable& t = tab[b];
            t.id = ar.take<int32_t>(Tcap); t.window = ar.take<int32_t>(Tcap); t.depth = ar.take<int32_t>(Tcap);
            t.shift = ar.take<int32_t>(Tcap); t.root_scan = ar.take<int32_t>(Tcap); t.root_node = ar.take<int32_t>(Tcap);
            t.root_cnllr = ar.take<double>(Tcap); t.root_f32 = ar.take<uint8_t>(Tcap);
            t.first = ar.take<int32_t>(Tcap); t.leaf_off = ar.take<int32_t>((size_t)Tcap + 1);
        }
        cost2[0] = ar.take<double>(Ncap); cost2[1] = ar.take<double>(Ncap);
        tchild = ar.take<int32_t>((size_t)Tcap + 1); tcend = ar.take<int32_t>((size_t)Tcap + 1);
        for (int g = 0; g < 2; ++g) {
            VTab& v = vts[g];
            v.vcap = vt.vcap; v.hmask = vt.hmask;
            v.Pv = ar.take<unsigned long long>((size_t)VT_PW * v.vcap); v.pdv = ar.take<double>(v.vcap);
            v.Gk = ar.take<float4>((size_t)2 * GKQ * v.vcap); v.child = ar.take<int32_t>((size_t)2 * v.vcap);
            v.slots = ar.take<unsigned long long>((size_t)v.hmask + 1); v.count = ar.take<unsigned>(16);
        }
        vt_remap = ar.take<int32_t>((size_t)2 * vt.vcap);
        alloc2[0] = ar.take<unsigned>((size_t)FG_REGIONS * 32); alloc2[1] = ar.take<unsigned>((size_t)FG_REGIONS * 32);
        used_bytes[0] = ar.take<unsigned char>(Mpad); used_bytes[1] = ar.take<unsigned char>(Mpad);
        status2 = ar.take<DevStatus>(2);
        init_dbg = ar.take<unsigned long long>(8);
        edge_t = ar.take<int32_t>(Ecap); edge_m = ar.take<int32_t>(Ecap);
        edges = ar.take<unsigned>((size_t)EDGE_SEGS * SegCap);
        edge_count = ar.take<int32_t>(EDGE_SEGS + 4);
        t_label = ar.take<int32_t>(Tcap); t_cluster = ar.take<int32_t>(Tcap); cl_ptr = ar.take<int32_t>((size_t)Tcap + 1);
        cl_members = ar.take<int32_t>(Tcap); multi_list = ar.take<int32_t>(Tcap); single_list = ar.take<int32_t>(Tcap);
        cl_counts = ar.take<int32_t>(8); big_list = ar.take<int32_t>(Tcap);
        cl_owner = ar.take<int32_t>(Tcap);
        if (cluster_big) cl_gtab = ar.take<int32_t>(cluster_big_ints(Tcap, n_mnodes));
        uf_owner = ar.take<unsigned long long>(n_mnodes); uf_parent2[0] = ar.take<unsigned long long>(Tcap); uf_parent2[1] = ar.take<unsigned long long>(Tcap); rec0 = ar.take<unsigned long long>(Tcap);
        team_list = ar.take<int32_t>(TEAM_MAX); team_state2[0] = ar.take<TeamState>(TEAM_MAX); team_state2[1] = ar.take<TeamState>(TEAM_MAX); team_res = ar.take<TeamResult>((size_t)TEAM_MAX * TEAM_W);
        // (TEAM_W copies of the ILP kernel's HBM scratch: a team member of a giant cluster works on its own, mht_blp.hip)
        u = ar.take<double>((size_t)n_mnodes * TEAM_W); usage = ar.take<int32_t>((size_t)n_mnodes * TEAM_W); mark = ar.take<int32_t>((size_t)n_mnodes * TEAM_W);
        bb_snap_rows = n_mnodes > 1024 ? n_mnodes : 1024;
        bb_snap = ar.take<double>((size_t)BB_SLOTS * BB_RE_LEVELS * bb_snap_rows); bb_busy = ar.take<int32_t>(BB_SLOTS);
        const size_t S = ((size_t)2 * Tcap + 2) * TEAM_W;
        best_h = ar.take<int32_t>(S); bb_ch = ar.take<int32_t>(S); bb_best = ar.take<int32_t>(S); bb_last_idx = ar.take<int32_t>(S);
        best_rc = ar.take<double>(S); bb_cost = ar.take<double>(S); bb_uused = ar.take<double>(S); bb_last_rc = ar.take<double>(S);
        bb_rest = ar.take<double>(S); bb_min = ar.take<double>(S);
        sel = ar.take<int32_t>(Tcap); cl_status = ar.take<int32_t>(Tcap); cl_iters = ar.take<int32_t>(Tcap); cl_nodes = ar.take<int32_t>(Tcap); cl_time = ar.take<int32_t>((size_t)8 * Tcap); grow_dbg = ar.take<unsigned long long>(32 + 16 * 4000); commit_log = ar.take<int32_t>(64 * 16);
        t_status = ar.take<int32_t>(Tcap); t_jdrop = ar.take<int32_t>(Tcap); t_count = ar.take<int32_t>(Tcap); t_firstsurv = ar.take<int32_t>(Tcap);
        new_index = ar.take<int32_t>(Tcap); near = ar.take<int32_t>(Tcap > 2048 ? Tcap : 2048); t_score = ar.take<double>(Tcap);      // (near: per candidate of one admission call, at most 2048 -- mht_forest_add_targets chunks)
        w_root_scan = ar.take<int32_t>(Tcap); w_root_node = ar.take<int32_t>(Tcap); w_root_cnllr = ar.take<double>(Tcap); w_root_f32 = ar.take<uint8_t>(Tcap);
        cnt = ar.take<FCounts>(1);
        report_dev2[0] = ar.take<char>(report_bytes); report_dev2[1] = ar.take<char>(report_bytes);
        z_dev = ar.take<float>((size_t)Z_RING * 2 * Mpad);
    }
};

void forest_destroy(mht_ctx* ctx) {
    Forest* f = ctx->forest;
    if (!f) return;
    ::hp_print();
    if (f->stage_stream) (void)hipStreamSynchronize(f->stage_stream);
    if (f->arena.base) (void)hipFree(f->arena.base);
    for (int b = 0; b < 2; ++b) {
        if (f->report_host2[b]) (void)hipHostFree(f->report_host2[b]);
        if (f->rep_ev[b]) (void)hipEventDestroy(f->rep_ev[b]);
    }
    if (f->z_host) (void)hipHostFree(f->z_host);
    for (int b = 0; b < Z_RING; ++b) if (f->z_ev[b]) (void)hipEventDestroy(f->z_ev[b]);
    for (int b = 0; b < 2; ++b) if (f->z_guard_ev[b]) (void)hipEventDestroy(f->z_guard_ev[b]);
    if (f->hint_host) (void)hipHostFree(f->hint_host);
    if (f->bhint_host) (void)hipHostFree(f->bhint_host);
    if (f->stage_host) (void)hipHostFree(f->stage_host);
    for (auto& cs : f->chain_slots) { if (cs.ev) (void)hipEventDestroy(cs.ev); if (cs.host) (void)hipHostFree(cs.host); }
    f->stage_dev.release();
    if (f->stage_stream) (void)hipStreamDestroy(f->stage_stream);
    if (f->grow_ev) (void)hipEventDestroy(f->grow_ev);
    if (f->init_ev) (void)hipEventDestroy(f->init_ev);
    if (f->evp) {
        for (int k = 0; k < EV_POOL; ++k) for (int i = 0; i < 5; ++i) (void)hipEventDestroy(f->evp[k][i]);
        delete[] f->evp;
    }
    delete f;
    ctx->forest = nullptr;
}

static int stage_host_ensure(Forest* f, size_t bytes) {
    if (bytes <= f->stage_host_bytes) return MHT_OK;
    if (f->stage_host) MHT_HIP_CHECK(hipHostFree(f->stage_host));
    f->stage_host = nullptr;
    f->stage_host_bytes = 0;
    MHT_HIP_CHECK(hipHostMalloc(&f->stage_host, bytes + 4096, hipHostMallocDefault));
    f->stage_host_bytes = bytes + 4096;
    return MHT_OK;
}

// The initiator of a streamed scan k: one workgroup on the side stream, queued BEHIND the staging kernel of scan k + 1 (by the call for that
// scan, or here by whoever needs its births first).  Why not on the ctx stream: a queue hands an XCD its next dispatch only when the previous
// one has drained there, so the next scan's grow launch would start behind the initiator (ILP end + 12 us instead of ILP start + 25 us,
// tools/api_timeline.py).  Why not in front of that staging kernel: the next grow launch's target workgroups -- resident on every CU -- wait
// for the staged scan; a staging kernel queued behind the initiator would wait for it, and the initiator (1024 threads) for a CU.
// It waits itself for the word its scan's ILP launch posts at entry (= its scan's grow launch is complete).
static int launch_deferred_init(mht_ctx* ctx, Forest* f) {
    if (!f->init_deferred) return MHT_OK;
    f->init_deferred = false;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(initiator_side_kernel, dim3(1), dim3(INIT_THREADS), 0, f->stage_stream, f->init_def_args, f->init_def_status,
                       static_cast<const int32_t*>(&f->cnt->overflow), &f->cnt->init_flag, static_cast<const unsigned long long*>(&f->cnt->z_flag), f->init_def_ztag,
                       static_cast<unsigned long long*>(nullptr), static_cast<const unsigned long long*>(&f->cnt->ilp_begun), f->init_dbg);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}
int forest_sync_side(mht_ctx* ctx) {
    Forest* f = ctx ? ctx->forest : nullptr;
    if (f && f->stage_stream && f->init_ev_pending) {
        { const int rc = launch_deferred_init(ctx, f); if (rc) return rc; }
        MHT_HIP_CHECK(hipStreamSynchronize(f->stage_stream));
    }
    return MHT_OK;
}
// the initiator's launch on another stream has to be complete for what is queued on the ctx stream next
static int wait_init_ev(mht_ctx* ctx, Forest* f) {
    if (!f->init_ev_pending) return MHT_OK;
    { const int rc = launch_deferred_init(ctx, f); if (rc) return rc; }
    if (f->init_ev_lazy) { MHT_HIP_CHECK(hipEventRecord(f->init_ev, f->stage_stream)); f->init_ev_lazy = false; }
    MHT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, f->init_ev, 0));
    f->init_ev_pending = false;
    return MHT_OK;
}
// runs the pending commit now (see Forest::commit_pending)
static PublishArgs publish_args(Forest* f) {      // the report of scan f->scan goes to the pinned block of its parity
    PublishArgs p;
    p.src = f->report_dev2[f->scan & 1]; p.dst = f->report_host_dev[f->scan & 1]; p.rec_off = (int)f->rec_off; p.birth_off = (int)f->birth_off;
    return p;
}
static int flush_commit(mht_ctx* ctx, Forest* f, bool publish = false) {
    if (!f->commit_pending) return MHT_OK;
    if (f->adm_pending) {      // commit + admission of the initiator's births, as one launch (what mht_forest_scan deferred)
        { const int rc = wait_init_ev(ctx, f); if (rc) return rc; }      // (the initiator ran on the side stream)
        PublishArgs pub = publish ? publish_args(f) : PublishArgs{};
        hipLaunchKernelGGL(post_scan_kernel<false>, dim3(1), dim3(1024), 0, ctx->stream, f->pending, f->pending_dyn, InitArgs{}, f->adm, 1, pub, 0, AisUsedArgs{});
        MHT_HIP_CHECK(hipGetLastError());
        if (publish) f->published_scan = f->scan;
        f->adm_pending = false;
        f->commit_pending = false;
        return MHT_OK;
    }
    if (publish) {      // (the host block of this parity may still be in the host's hands: two scans ago)
        hipLaunchKernelGGL(commit_publish_kernel, dim3(1), dim3(COMMIT_THREADS), 0, ctx->stream, f->pending, f->pending_dyn, publish_args(f));
        f->published_scan = f->scan;
    } else
    hipLaunchKernelGGL(commit_kernel, dim3(1), dim3(COMMIT_THREADS), 0, ctx->stream, f->pending, f->pending_dyn);
    MHT_HIP_CHECK(hipGetLastError());
    f->commit_pending = false;
    return MHT_OK;
}

__global__ __launch_bounds__(1024) void publish_kernel(const PublishArgs pub) { publish_report<1024>(pub); }
// the deferred report push (Forest::pub_deferred) now, as a launch of its own
static int flush_publish(mht_ctx* ctx, Forest* f) {
    if (!f->pub_deferred) return MHT_OK;
    if (f->adm_pending) { const int rc = flush_commit(ctx, f); if (rc) return rc; }      // (the report is complete behind the commit and the admission)
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1024), 0, ctx->stream, f->pub_args);
    MHT_HIP_CHECK(hipGetLastError());
    MHT_HIP_CHECK(hipEventRecord(f->rep_ev[f->pub_slot], ctx->stream));
    f->rep_by_flag[f->pub_slot] = false;
    f->rep_started[f->pub_slot] = true;
    f->host_block_scan[f->pub_slot] = f->scan;
    f->pub_deferred = false;
    return MHT_OK;
}

static LayerView view_of(const mht_nodes& l) { return LayerView{l.x, l.cnllr, l.parent, l.meas, l.flags, l.cov, l.P}; }

}  // namespace mht

using namespace mht;

static int forest_create_impl(mht_ctx* ctx, const mht_model* model, const mht_forest_config* cfg, uint32_t flags) {
    MHT_REQUIRE(ctx && model && cfg, "mht_forest_create: null argument");
    MHT_REQUIRE((flags & ~(uint32_t)(MHT_FOREST_AIS | MHT_FOREST_CT)) == 0, "mht_forest_create_ex: unknown flags 0x%x", flags);
    if (flags & MHT_FOREST_AIS) {
        MHT_REQUIRE(NX == 4, "mht_forest_create_ex: AIS messages report four states (models/ais.py); this is the %d-state build", NX);
        MHT_REQUIRE(cfg->n_scan <= 7, "mht_forest_create_ex: an AIS forest keeps two rows per level in a 16-entry path record: n_scan must be <= 7 (got %d)", cfg->n_scan);
    }
    MHT_REQUIRE(!ctx->forest, "mht_forest_create: the ctx already owns a forest");
    MHT_REQUIRE(cfg->n_scan >= 1 && cfg->n_scan + RING_EXTRA <= MAXR, "mht_forest_create: n_scan must be in [1, %d]", MAXR - RING_EXTRA);
    MHT_REQUIRE(cfg->max_meas >= 1 && cfg->max_meas <= 4096, "mht_forest_create: max_meas must be in [1, 4096]");
    MHT_REQUIRE(cfg->max_targets >= 1 && cfg->max_targets <= 8192, "mht_forest_create: max_targets must be in [1, 8192]");
    MHT_REQUIRE(cfg->max_nodes >= 2 * cfg->max_targets + 512, "mht_forest_create: max_nodes must be at least 2 * max_targets + 512");
    MHT_REQUIRE((cfg->n_scan + RING_EXTRA) * (((cfg->max_meas + 63) / 64) * 64) <= 65536,
                "mht_forest_create: (n_scan + %d) x max_meas = %d measurement nodes exceed the 16 bits of an edge record", RING_EXTRA, (cfg->n_scan + RING_EXTRA) * (((cfg->max_meas + 63) / 64) * 64));
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    Forest* f = new (std::nothrow) Forest();
    MHT_REQUIRE(f, "mht_forest_create: out of host memory");
    f->cfg = *cfg;
    if (f->cfg.blp_max_iter < 0) f->cfg.blp_max_iter = 200;
    if (f->cfg.blp_node_limit <= 0) f->cfg.blp_node_limit = 1 << 20;
    f->model = *model;
    f->Tcap = cfg->max_targets;
    f->Ncap = cfg->max_nodes;
    f->Mpad = ((cfg->max_meas + 63) / 64) * 64;
    f->R = cfg->n_scan + RING_EXTRA;
    f->PD = cfg->n_scan + 1;
    f->n_mnodes = f->R * f->Mpad;
    f->AW = f->n_mnodes / 64;
    {   // covariance-value table: ids are never recycled; the headline stream meets ~4 k distinct values in its first 120 scans and
        // ~10 new ones per scan afterwards (a full table voids the scan with MHT_E_CAPACITY)
        long long vc = 2ll * f->Ncap;
        if (vc < (1 << 16)) vc = 1 << 16;
        if (vc > (1 << 20)) vc = 1 << 20;
        // (an AIS forest: the float64 covariances of AIS-updated targets are target specific -- nothing is shared across targets behind
        // a message -- and take two ids each, as do their pseudo parents: ~4 ids per leaf of a promoted target and scan, 200 k per scan at
        // the headline size with half of the ships reporting, and a generation must last R + 2 scans.  1.9 GB per generation at 8 M ids:
        // sized for the 288 GB part)
        if (flags & MHT_FOREST_AIS) { vc = 16ll * f->Ncap; if (vc < (1 << 20)) vc = 1 << 20; if (vc > (1 << 23)) vc = 1 << 23; }
        if (const char* e = getenv("MHT_VTAB_CAP")) vc = atoll(e);
        f->vt.vcap = (int)vc;
        unsigned hs = 1;
        while (hs < 4u * (unsigned)vc) hs <<= 1;
        f->vt.hmask = hs - 1;
    }
    f->Ecap = 4 * f->Ncap;              // edges the clustering kernel can take beyond its LDS list (spill arrays)
    f->SegCap = f->Ncap / 8 + 1024;     // edges per segment (64 segments)
    {   // node index space of a layer: [static block per target slot | overflow area in FG_REGIONS regions | roots born into the layer]
        const long long usable = (long long)f->Ncap - f->Tcap;
        long long bc = usable * 3 / 4 / f->Tcap;
        if (bc > 256) bc = 256;
        if (bc < 16) bc = 0;          // (tiny pools: everything through the overflow counters)
        if (const char* e = getenv("MHT_BLOCK_CAP")) bc = atoi(e) < bc ? atoi(e) : bc;      // development: 0 = no static blocks
        f->block_cap = (int)bc;
        f->over_base = f->Tcap * f->block_cap;
        f->region_cap = (int)((usable - f->over_base) / FG_REGIONS);
        f->root_base = f->over_base + FG_REGIONS * f->region_cap;
    }
    f->debug = getenv("MHT_GROW_DEBUG") != nullptr;
    { const char* e = getenv("MHT_CT_SPILL"); f->ct_spill = e && e[0] == '1'; }
    { const char* e = getenv("MHT_BLP_GRID_HINT"); f->grid_by_hint = !(e && e[0] == '0'); }
    { const char* e = getenv("MHT_BLP_FORCE_HBM"); f->force_hbm = e && e[0] == '1'; }
    { const char* e = getenv("MHT_BLP_NO_ENUM"); f->no_enum = e && e[0] == '1'; }
    { const char* e = getenv("MHT_BLP_NO_TEAMS"); f->teams = !(e && e[0] == '1'); }
    { const char* e = getenv("MHT_ADM_FUSE"); f->adm_fuse = !(e && e[0] == '0'); }
    { const char* e = getenv("MHT_INIT_SIDE"); f->init_side = (e && e[0] == '1'); }      // (measured: the two event operations per scan cost the host more than the 6 us save the device -- 91 against 76 us per streamed scan)
    MHT_HIP_CHECK(hipEventCreateWithFlags(&f->grow_ev, hipEventDisableTiming));
    MHT_HIP_CHECK(hipEventCreateWithFlags(&f->init_ev, hipEventDisableTiming));
    f->pds = f->PD <= 8 ? 8 : 16;
    f->cluster_big = !cluster_fits_lds(f->Tcap, f->n_mnodes);
    { const char* e = getenv("MHT_NO_UF"); f->uf_ok = !(e && e[0] == '1') && blp_uf_fits(f->Tcap, f->n_mnodes); }
    { const char* e = getenv("MHT_NO_OVERLAP"); f->ovl_ok = !(e && e[0] == '1'); }
    { const char* e = getenv("MHT_INIT_QUEUE"); f->init_side_q = !(e && e[0] == '0'); }
    { const char* e = getenv("MHT_REPORT_FLAG"); f->rep_flag_ok = !(e && e[0] == '0'); }
    // (rocprofv3 --pmc runs ONE kernel at a time across all queues, in the order the queues happen to be served: a launch that waits for a
    // launch on another queue never sees it start.  Under counter collection the initiator stays on the ctx stream.)
    { const char* e = getenv("ROCPROF_COUNTER_COLLECTION"); if (e && e[0] == '1') { f->init_side_q = false; f->serial_prof = true; } }
    if (flags & MHT_FOREST_CT) {       // the transition is rebuilt per hypothesis from its turn rate (pymht_amd/models/ct.py): T = A[4][5]
        if (NX != 6 || (flags & MHT_FOREST_AIS)) { delete f; set_error("mht_forest_create_ex: MHT_FOREST_CT needs the six-state build of the library and no MHT_FOREST_AIS"); return MHT_E_INVALID; }
        f->ct = true;
        f->ct_T = (double)model->A[(NX >= 6 ? 4 : 0) * NX + (NX >= 6 ? 5 : 0)];
        f->ovl_ok = false;
    }
    if (flags & MHT_FOREST_AIS) {      // two halves per record: radar rows, AIS rows
        f->ais = true;
        f->ais_half = f->PD <= 4 ? 4 : 8;
        f->pds = 2 * f->ais_half;
    }

    f->used_off = sizeof(ReportHeader);
    f->birth_off = (f->used_off + (size_t)(f->Mpad / 64) * 8 + 15) & ~(size_t)15;
    f->rec_off = f->birth_off + (size_t)BIRTH_CAP * sizeof(mht_birth_report);
    f->done_off = (f->rec_off + (size_t)f->Tcap * sizeof(mht_target_report) + 63) & ~(size_t)63;      // (PublishArgs::done: the pushing workgroups' words, a cache line of their own)
    f->report_bytes = f->done_off + 128;
    Arena probe;
    f->layout(probe);                     // first pass: size
    const size_t total = probe.off + 4096;
    void* base = nullptr;
    if (hipMalloc(&base, total) != hipSuccess) {
        delete f;
        set_error("mht_forest_create: hipMalloc of %zu bytes failed", total);
        return MHT_E_HIP;
    }
    f->arena.base = static_cast<char*>(base);
    f->arena.size = total;
    f->arena.off = 0;
    f->layout(f->arena);
    ctx->forest = f;
    {   // the prune epilogue addresses layer k of the ring arithmetically: all layers must be laid out identically
        const ptrdiff_t stride = reinterpret_cast<const char*>(f->layer[1].x) - reinterpret_cast<const char*>(f->layer[0].x);
        bool ok = true;
        for (int k = 1; k < f->R; ++k) {
            const mht_nodes &l0 = f->layer[0], &lk = f->layer[k];
            ok = ok && reinterpret_cast<const char*>(lk.x) - reinterpret_cast<const char*>(l0.x) == k * stride &&
                 reinterpret_cast<const char*>(lk.cnllr) - reinterpret_cast<const char*>(l0.cnllr) == k * stride &&
                 reinterpret_cast<const char*>(lk.meas) - reinterpret_cast<const char*>(l0.meas) == k * stride &&
                 reinterpret_cast<const char*>(lk.flags) - reinterpret_cast<const char*>(l0.flags) == k * stride;
        }
        if (!ok) {
            forest_destroy(ctx);
            set_error("mht_forest_create: internal error, ring layers are not equally spaced");
            return MHT_E_INVALID;
        }
    }
    MHT_HIP_CHECK(hipMemsetAsync(base, 0, total, ctx->stream));
    for (int g = 0; g < 2; ++g) {
        MHT_HIP_CHECK(hipMemsetAsync(f->vts[g].child, 0xff, (size_t)2 * f->vt.vcap * sizeof(int32_t), ctx->stream));      // -1: no transition known
        f->vts[g].overflow = &f->cnt->overflow;
    }
    f->vt = f->vts[0];
    for (int b = 0; b < 2; ++b) {
        MHT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&f->report_host2[b]), f->report_bytes, hipHostMallocDefault));
        memset(f->report_host2[b], 0, f->report_bytes);
        MHT_HIP_CHECK(hipEventCreateWithFlags(&f->rep_ev[b], hipEventDisableTiming));
    }
    f->report_host = f->report_host2[0];
    MHT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&f->z_host), (size_t)Z_RING * 2 * f->Mpad * sizeof(float), hipHostMallocDefault));
    for (int b = 0; b < Z_RING; ++b) MHT_HIP_CHECK(hipEventCreateWithFlags(&f->z_ev[b], hipEventDisableTiming));
    MHT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&f->z_host_dev), f->z_host, 0));
    for (int b = 0; b < 2; ++b) MHT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&f->report_host_dev[b]), f->report_host2[b], 0));
    // (the blocks of mht_forest_chains_begin: pinned memory is allocated HERE, not when the first track dies in the middle of a stream -- hipHostMalloc takes
    // milliseconds; 128 KB hold ~130 chains of a window of 5, a larger batch grows its block)
    for (auto& cs : f->chain_slots) {
        MHT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&cs.host), 128 * 1024, hipHostMallocMapped));
        MHT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&cs.dev), cs.host, 0));
        cs.bytes = 128 * 1024;
        MHT_HIP_CHECK(hipEventCreateWithFlags(&cs.ev, hipEventDisableTiming));
    }
    MHT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&f->hint_host), 64, hipHostMallocMapped));
    memset(f->hint_host, 0, 64);
    MHT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&f->hint_dev), f->hint_host, 0));
    MHT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&f->bhint_host), 64 * sizeof(unsigned long long), hipHostMallocMapped));
    memset(f->bhint_host, 0, 64 * sizeof(unsigned long long));
    MHT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&f->bhint_dev), f->bhint_host, 0));
    // (a forest whose clustering tables -- 16 B per target + 4 B per measurement node -- do not fit the kernel's 150 KiB of LDS keeps them
    // in HBM: cluster_big_kernel)
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHT_OK;
}

extern "C" int mht_forest_create(mht_ctx* ctx, const mht_model* model, const mht_forest_config* cfg) { return forest_create_impl(ctx, model, cfg, 0u); }
extern "C" int mht_forest_create_ex(mht_ctx* ctx, const mht_model* model, const mht_forest_config* cfg, uint32_t flags) {
    return forest_create_impl(ctx, model, cfg, flags);
}

// The AIS messages of the NEXT scan (Tracker.addMeasurementList(scanList, aisList), tracker.py:162): grouped as the reference walks
// them (include/mht_amd.h: mht_fuse_ais), copied to the device now, consumed -- and disarmed -- by the next step.  nA = 0 disarms.
extern "C" int mht_forest_set_ais(mht_ctx* ctx, const mht_ais_group* groups, int32_t nG, const mht_ais_msg* msgs, int32_t nA, double eta2_ais,
                                  double lambda_ais) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_set_ais: no forest");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->ais, "mht_forest_set_ais: the forest was not created with MHT_FOREST_AIS (mht_forest_create_ex)");
    if (f->in_groups > 0) { set_error("mht_forest_set_ais: the forest is a member of a group (mht_group_step takes no AIS messages)"); return MHT_E_STATE; }
    MHT_REQUIRE(nG >= 0 && nA >= 0 && nG <= f->ais_group_cap && nA <= f->ais_msg_cap, "mht_forest_set_ais: %d groups / %d messages exceed the capacity (%d / %d)",
                nG, nA, f->ais_group_cap, f->ais_msg_cap);
    if (nA == 0 || nG == 0) { f->ais_armed = false; f->ais_nG = f->ais_nA = 0; return MHT_OK; }
    MHT_REQUIRE(groups && msgs, "mht_forest_set_ais: null argument");
    MHT_REQUIRE(lambda_ais > 0.0 && eta2_ais > 0.0, "mht_forest_set_ais: eta2_ais and lambda_ais must be positive (tracker.py:438 needs a finite radarRange)");
    for (int g = 0; g < nG; ++g)
        MHT_REQUIRE(groups[g].first >= 0 && groups[g].count >= 0 && groups[g].first + groups[g].count <= nA, "mht_forest_set_ais: group %d outside the message list", g);
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    // (pageable sources: the runtime stages them before the call returns, the caller's arrays are free again)
    MHT_HIP_CHECK(hipMemcpyAsync(f->ais_groups_dev, groups, (size_t)nG * sizeof(mht_ais_group), hipMemcpyHostToDevice, ctx->stream));
    MHT_HIP_CHECK(hipMemcpyAsync(f->ais_msgs_dev, msgs, (size_t)nA * sizeof(mht_ais_msg), hipMemcpyHostToDevice, ctx->stream));
    f->ais_armed = true; f->ais_nG = nG; f->ais_nA = nA; f->ais_eta2 = eta2_ais; f->ais_lambda = lambda_ais;
    return MHT_OK;
}

// identities of the nodes [first, first + count) of the layer of scan `scan` (the last scan or one of the window before it): mmsi = the AIS
// message a node was updated with (0: none), hist = the identity its track is bound to (pyTarget.py:297-302).  Synchronous.
extern "C" int mht_forest_read_mmsi(mht_ctx* ctx, int32_t scan, int32_t first, int32_t count, int32_t* mmsi, int32_t* hist) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_read_mmsi: no forest");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->ais, "mht_forest_read_mmsi: the forest was not created with MHT_FOREST_AIS");
    MHT_REQUIRE(scan >= 0 && scan <= f->scan && f->scan - scan < f->R - 1, "mht_forest_read_mmsi: scan %d is outside the ring (last scan %d)", scan, f->scan);
    MHT_REQUIRE(first >= 0 && count >= 0 && (long long)first + count <= f->Ncap, "mht_forest_read_mmsi: node range outside the layer");
    if (count == 0) return MHT_OK;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    if (mmsi) MHT_HIP_CHECK(hipMemcpyAsync(mmsi, f->l_mmsi[scan % f->R] + first, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (hist) MHT_HIP_CHECK(hipMemcpyAsync(hist, f->l_hmmsi[scan % f->R] + first, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHT_OK;
}

// the identities of n given nodes of one layer (a gather: the roots that join a track's committed history sit all over the layer's index space)
__global__ void mmsi_gather_kernel(const int32_t* l_mmsi, const int32_t* l_hmmsi, const int32_t* nodes, int n, int cap, int32_t* out_mmsi, int32_t* out_hist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int nd = nodes[i];
    const bool ok = nd >= 0 && nd < cap;
    out_mmsi[i] = ok ? l_mmsi[nd] : 0;
    out_hist[i] = ok ? l_hmmsi[nd] : 0;
}
extern "C" int mht_forest_read_mmsi_nodes(mht_ctx* ctx, int32_t scan, int32_t n, const int32_t* nodes, int32_t* mmsi, int32_t* hist) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_read_mmsi_nodes: no forest");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->ais, "mht_forest_read_mmsi_nodes: the forest was not created with MHT_FOREST_AIS");
    MHT_REQUIRE(scan >= 0 && scan <= f->scan && f->scan - scan < f->R - 1, "mht_forest_read_mmsi_nodes: scan %d is outside the ring (last scan %d)", scan, f->scan);
    MHT_REQUIRE(n >= 0 && (n == 0 || nodes), "mht_forest_read_mmsi_nodes: bad node list");
    if (n == 0) return MHT_OK;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n * 4;
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // (the staging buffers are reused)
    int rc = stage_host_ensure(f, 3 * bytes + 64);
    if (rc) return rc;
    rc = f->stage_dev.ensure(3 * bytes + 64);
    if (rc) return rc;
    char* h = static_cast<char*>(f->stage_host);
    char* d = static_cast<char*>(f->stage_dev.ptr);
    memcpy(h, nodes, bytes);
    MHT_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mmsi_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, f->l_mmsi[scan % f->R], f->l_hmmsi[scan % f->R],
                       reinterpret_cast<const int32_t*>(d), n, f->Ncap, reinterpret_cast<int32_t*>(d + bytes), reinterpret_cast<int32_t*>(d + 2 * bytes));
    MHT_HIP_CHECK(hipGetLastError());
    MHT_HIP_CHECK(hipMemcpyAsync(h + bytes, d + bytes, 2 * bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (mmsi) memcpy(mmsi, h + bytes, bytes);
    if (hist) memcpy(hist, h + 2 * bytes, bytes);
    return MHT_OK;
}

extern "C" int mht_forest_add_targets_dev(mht_ctx* ctx, int32_t n, const double* x0, const float* P0, const uint8_t* flags,
                                          const double* pd, const int32_t* meas, int32_t check_neighbours,
                                          uint8_t* accepted, int32_t* ids) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_add_targets_dev: no forest");
    Forest* f = ctx->forest;
    MHT_REQUIRE(n >= 0 && (n == 0 || (x0 && P0 && flags && pd && meas)), "mht_forest_add_targets_dev: null input");
    if (n == 0) return MHT_OK;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    // (the scan's commit is still pending and the batch fits one launch: commit and admission as ONE launch -- post_scan_kernel,
    // what flush_commit runs for a pending admission -- instead of a commit launch and an admission launch)
    const bool fuse = n <= 2048 && f->commit_pending && !f->adm_pending && !f->ais && !f->pub_deferred;
    if (!fuse) { const int rc = flush_commit(ctx, f); if (rc) return rc; }
    AddArgs a = {};
    a.n = n; a.x0 = x0; a.pd = pd; a.P0 = P0; a.meas = meas; a.flags = flags; a.ids = ids; a.accepted = accepted;
    a.check = check_neighbours; a.thr = f->cfg.merge_threshold;
    const int nb = (f->scan + 1) & 1;
    a.layer = f->layer[f->scan % f->R];
    a.tab = f->tab[nb]; a.vidx = nb;
    a.path = f->path[f->scan & 1]; a.apath = f->apath[f->scan & 1]; a.PD = f->pds;
    a.cnt = f->cnt; a.scan = f->scan; a.Nwin = f->cfg.n_scan; a.Tcap = f->Tcap;
    a.near = f->near;
    fill_model_only(a.model, &f->model); a.vt = f->vt; a.root_base = f->root_base; a.ct_Proot = f->ct ? f->ct_Proot[f->scan % f->R] : nullptr;
    if (f->ais) { a.mmsi = f->l_mmsi[f->scan % f->R]; a.hmmsi = f->l_hmmsi[f->scan % f->R]; }
    MHT_REQUIRE(n <= f->Tcap, "mht_forest_add_targets: %d candidates exceed max_targets", n);
    if (fuse) { f->adm = a; f->adm_pending = true; const int rc = flush_commit(ctx, f); if (rc) return rc; }
    else
    // the kernel keeps the candidates admitted so far in LDS (2048 entries): larger batches go in chunks, candidates of
    // earlier chunks are leaves of the forest by then and are tested as such
    for (int c0 = 0; c0 < n; c0 += 2048) {
        AddArgs ac = a;
        ac.n = n - c0 < 2048 ? n - c0 : 2048;
        ac.x0 = x0 + (size_t)c0 * NX; ac.pd = pd + c0; ac.P0 = P0 + (size_t)c0 * NP; ac.meas = meas + c0; ac.flags = flags + c0;
        ac.ids = ids ? ids + c0 : nullptr; ac.accepted = accepted ? accepted + c0 : nullptr;
        hipLaunchKernelGGL(add_targets_kernel, dim3(1), dim3(1024), 0, ctx->stream, ac);
        MHT_HIP_CHECK(hipGetLastError());
    }
    f->nT_ub = (f->nT_ub + n < f->Tcap) ? f->nT_ub + n : f->Tcap;
    f->L_ub = (f->L_ub + n < f->Ncap) ? f->L_ub + n : f->Ncap;
    f->births_since_step += n;
    f->births_after[f->scan % 64] += n;
    return MHT_OK;
}

extern "C" int mht_forest_add_targets(mht_ctx* ctx, int32_t n, const double* x0, const float* P0, const uint8_t* flags,
                                      const double* pd, const int32_t* meas, int32_t check_neighbours, uint8_t* accepted,
                                      int32_t* ids) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_add_targets: no forest");
    Forest* f = ctx->forest;
    MHT_REQUIRE(n >= 0 && (n == 0 || (x0 && P0 && flags && pd && meas)), "mht_forest_add_targets: null input");
    if (n == 0) return MHT_OK;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    // pack inputs: x0 | pd | P0 | meas | flags    outputs: ids | accepted
    const size_t o_x = 0, o_pd = o_x + (size_t)n * (NX * 8), o_P = o_pd + (size_t)n * 8, o_m = o_P + (size_t)n * (NP * 4),
                 o_f = o_m + (size_t)n * 4, o_id = (o_f + n + 7) & ~(size_t)7, o_acc = o_id + (size_t)n * 4,
                 total = o_acc + n + 16;
    // the staging buffers are reused: anything still in flight from a previous call must have drained
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int rc = stage_host_ensure(f, total);
    if (rc) return rc;
    rc = f->stage_dev.ensure(total);
    if (rc) return rc;
    char* h = static_cast<char*>(f->stage_host);
    memcpy(h + o_x, x0, (size_t)n * (NX * 8)); memcpy(h + o_pd, pd, (size_t)n * 8); memcpy(h + o_P, P0, (size_t)n * (NP * 4));
    memcpy(h + o_m, meas, (size_t)n * 4); memcpy(h + o_f, flags, n);
    char* d = static_cast<char*>(f->stage_dev.ptr);
    MHT_HIP_CHECK(hipMemcpyAsync(d, h, o_id, hipMemcpyHostToDevice, ctx->stream));
    // (ids = -1, accepted = 0 unless the admission says otherwise: the fused commit + admission launch skips the admission on a void scan)
    MHT_HIP_CHECK(hipMemsetAsync(d + o_id, 0xff, (size_t)n * 4, ctx->stream));
    MHT_HIP_CHECK(hipMemsetAsync(d + o_acc, 0, (size_t)n + 16, ctx->stream));
    rc = mht_forest_add_targets_dev(ctx, n, (const double*)(d + o_x), (const float*)(d + o_P), (const uint8_t*)(d + o_f),
                                    (const double*)(d + o_pd), (const int32_t*)(d + o_m), check_neighbours,
                                    (uint8_t*)(d + o_acc), (int32_t*)(d + o_id));
    if (rc) return rc;
    MHT_HIP_CHECK(hipMemcpyAsync(h + o_id, d + o_id, total - o_id, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ids) memcpy(ids, h + o_id, (size_t)n * 4);
    if (accepted) memcpy(accepted, h + o_acc, n);
    return MHT_OK;
}

// ---- argument blocks of a scan's launches ------------------------------------------------------------------------------
// Everything in them depends on the scan number s only through s % R (ring layer) and s & 1 (double-buffered tables): they repeat
// with period 2R.  What really changes per scan -- the scan's measurements, the grid, the scan number in the report -- travels
// separately (FDyn, CommitDyn), so that a group of sectors can keep its blocks in HBM (mht_group_*).
namespace mht {

static void fill_fgrow(const Forest* f, int s, bool fused, FGrowArgs& g) {
    g = FGrowArgs{};
    const int cb = s & 1;
    const mht_nodes& in = f->layer[(s - 1) % f->R];
    const mht_nodes& out = f->layer[s % f->R];
    fill_model_only(g.model, &f->model);
    g.default_pd = f->model.default_pd; g.default_miss_nllr = f->model.default_miss_nllr;
    g.x = in.x; g.cnllr = in.cnllr; g.pd = in.pd; g.cov = in.cov; g.flags = in.flags;
    g.cap = f->Ncap; g.vt = f->vt;
    g.in_path = f->path[(s - 1) & 1]; g.in_apath = f->apath[(s - 1) & 1]; g.pds = f->pds;
    g.Tcap = f->Tcap;
    if (fused) {      // tables of the scan before, still uncommitted
        const int pb = (s - 1) & 1;
        g.nT_dev = &f->cnt->nTv[pb];
        g.p_status = f->t_status; g.p_count = f->t_count; g.p_jdrop = f->t_jdrop; g.p_firstsurv = f->t_firstsurv;
        g.p_depth = f->tab[pb].depth; g.t_root_cnllr = f->w_root_cnllr; g.t_root_f32 = f->w_root_f32;
    } else {
        g.nT_dev = &f->cnt->nT;
        g.t_root_cnllr = f->tab[cb].root_cnllr; g.t_root_f32 = f->tab[cb].root_f32;
    }
    g.t_first = f->tab[cb].first; g.t_leaf_off = f->tab[cb].leaf_off; g.t_depth = f->tab[cb].depth; g.t_shift = f->tab[cb].shift;
    g.nT_new = &f->cnt->nT; g.b_root_cnllr = f->tab[cb].root_cnllr; g.b_root_f32 = f->tab[cb].root_f32;
    g.ox = out.x; g.ocnllr = out.cnllr; g.opd = out.pd; g.oparent = out.parent; g.omeas = out.meas; g.ocov = out.cov;
    g.oflags = out.flags;
    g.out_path = f->path[s & 1]; g.out_apath = f->apath[s & 1]; g.ocost = f->cost2[s & 1];
    g.tchild = f->tchild; g.tcend = f->tcend;
    g.PD = f->PD; g.Nwin = f->cfg.n_scan; g.cur_slot_base = (s % f->R) * f->Mpad; g.AW = f->AW;
    g.alloc = f->alloc2[s & 1]; g.block_cap = f->block_cap; g.over_base = f->over_base; g.region_cap = f->region_cap;
    g.edges = f->edges; g.edge_count = f->edge_count; g.edge_cap = f->SegCap;
    g.used_bytes = f->used_bytes[s & 1];
    g.status = f->status2 + (s & 1); g.prev_status = f->status2 + ((s - 1) & 1); g.sticky_overflow = &f->cnt->overflow;
    g.rec0 = f->rec0; g.new_index = f->new_index; g.ni_flag = &f->cnt->ni_flag;
    g.uf_owner = f->uf_owner; g.uf_parent = f->uf_parent2[s & 1];      // (by scan parity: the grow launch of scan s + 1 links while the ILP launch of scan s still reads)
     g.uf_team_state = f->teams ? f->team_state2[s & 1] : nullptr;
    if (f->ct) { g.ct.on = 1; g.ct.gains = f->ct_gains; g.ct.xbar = f->ct_xbar; g.ct.zhat = f->ct_zhat; g.ct.hw_spill = f->ct_hw_spill; }
    if (f->ais) {
        g.ais.nf = f->ais_nf; g.ais.off = f->ais_off; g.ais.rec = f->ais_rec; g.ais.half = f->ais_half;
        g.ais.hmmsi_in = f->l_hmmsi[(s - 1) % f->R]; g.ais.ommsi = f->l_mmsi[s % f->R]; g.ais.ohmmsi = f->l_hmmsi[s % f->R];
        g.ais.t_window = f->tab[fused ? ((s - 1) & 1) : cb].window;
    }
}

static void fill_cluster(const Forest* f, int s, ClusterArgs& c) {
    c = ClusterArgs{};
    c.assoc = nullptr; c.AW = f->AW; c.nT_dev = &f->cnt->nT; c.Tcap = f->Tcap;
    c.edge_t = f->edge_t; c.edge_m = f->edge_m; c.Ecap = f->Ecap; c.n_mnodes = f->n_mnodes; c.clear_rows = 0;
    c.alloc_reset = f->alloc2[s & 1];
    c.edges_in = f->edges; c.edge_count = f->edge_count; c.ticket_reset = nullptr; c.seg_cap = f->SegCap;
    c.status = f->status2 + (s & 1); c.status_other = f->status2 + ((s - 1) & 1);
    c.dbg = f->debug ? reinterpret_cast<int32_t*>(f->grow_dbg) + 16 : nullptr;
    c.t_label = f->t_label; c.t_cluster = f->t_cluster; c.cl_ptr = f->cl_ptr; c.cl_members = f->cl_members;
    c.multi_list = f->multi_list; c.single_list = f->single_list; c.counts = f->cl_counts;
    c.team_list = f->teams ? f->team_list : nullptr; c.team_state = f->teams ? f->team_state2[s & 1] : nullptr;
    c.gtab = f->cl_gtab;
    cluster_prepare(c);
}

static void fill_blp(const Forest* f, int s, BlpArgs& b) {
    b = BlpArgs{};
    const int cb = s & 1;
    const mht_nodes& out = f->layer[s % f->R];
    b.cl_ptr = f->cl_ptr; b.cl_members = f->cl_members; b.multi_list = f->multi_list; b.single_list = f->single_list;
    b.team_list = f->teams ? f->team_list : nullptr; b.team_state = f->team_state2[s & 1]; b.team_res = f->team_res;
    b.counts = f->cl_counts; b.big_count = f->cl_counts + 4; b.big_list = f->big_list; b.tchild = f->tchild; b.tcend = f->tcend; b.cost = f->cost2[s & 1]; b.cnllr = out.cnllr;
    b.path = f->path[s & 1]; b.cap = f->Ncap; b.PD = f->ais ? f->pds : f->PD; b.pds = f->pds;      // (AIS forest: every entry of a record can be a row)
    b.u = f->u; b.usage = f->usage; b.mark = f->mark; b.n_mnodes = f->n_mnodes;
    if (f->teams) { b.tm_sm = (size_t)f->n_mnodes; b.tm_ss = (size_t)2 * f->Tcap + 2; }
    b.bb_snap = f->bb_snap; b.bb_busy = f->bb_busy; b.bb_snap_rows = f->bb_snap_rows;
    b.best_h = f->best_h; b.best_rc = f->best_rc; b.bb_ch = f->bb_ch; b.bb_best = f->bb_best; b.bb_cost = f->bb_cost;
    b.bb_uused = f->bb_uused; b.bb_last_rc = f->bb_last_rc; b.bb_last_idx = f->bb_last_idx; b.bb_rest = f->bb_rest; b.bb_min = f->bb_min;
    b.sel = f->sel; b.cl_status = f->cl_status; b.cl_iters = f->cl_iters; b.cl_nodes = f->cl_nodes; b.cl_time = f->cl_time;
    b.max_iter = f->cfg.blp_max_iter; b.node_limit = f->cfg.blp_node_limit; b.status = f->status2 + (s & 1);
    b.force_hbm = f->force_hbm ? 1 : 0;
    b.no_enum = f->no_enum ? 1 : 0;
    b.skip_dead = f->prune_thr > 0.f ? 1 : 0;
    b.time_limit = f->blp_time_limit;
    { static int nr = -1; if (nr < 0) { const char* e = getenv("MHT_BLP_NO_REDUCE"); nr = (e && e[0] == '1') ? 1 : 0; } b.no_reduce = nr; }
    b.x = out.x; b.flags = out.flags; b.t_root_cnllr = f->tab[cb].root_cnllr; b.t_root_f32 = f->tab[cb].root_f32;
    b.t_depth = f->tab[cb].depth; b.t_window = f->tab[cb].window;
    b.apath = f->apath[s & 1]; b.R = f->R; b.kc = s % f->R;
    b.ring0 = RingLayer{f->layer[0].x, f->layer[0].cnllr, f->layer[0].meas, f->layer[0].flags};
    b.ring_stride = (size_t)(reinterpret_cast<const char*>(f->layer[1].x) - reinterpret_cast<const char*>(f->layer[0].x));   // layers are laid out identically, back to back
    b.t_id = f->tab[cb].id; b.t_root_scan = f->tab[cb].root_scan; b.t_root_node = f->tab[cb].root_node; b.t_label = f->t_label;
    b.rec = reinterpret_cast<mht_target_report*>(f->report_dev2[s & 1] + f->rec_off);
    b.w_root_scan = f->w_root_scan; b.w_root_node = f->w_root_node; b.w_root_cnllr = f->w_root_cnllr; b.w_root_f32 = f->w_root_f32;
    b.t_alive = f->t_status; b.t_jdrop = f->t_jdrop; b.t_count = f->t_count; b.t_firstsurv = f->t_firstsurv; b.t_score = f->t_score;
    b.Nwin = f->cfg.n_scan; b.score_limit = f->cfg.score_limit; b.cnllr_limit = f->cfg.cnllr_limit;
    b.radar_x = f->cfg.radar_x; b.radar_y = f->cfg.radar_y; b.radar_range = f->cfg.radar_range;
    // (clusters from the grow launch's union-find: switched on per scan by the caller, b.uf_epoch = scan number)
    b.uf_parent = f->uf_parent2[s & 1]; b.nT_dev = &f->cnt->nT; b.uf_cap = f->Tcap; b.status_other = f->status2 + ((s - 1) & 1); b.alloc_reset = f->alloc2[s & 1];
    b.t_cluster = f->t_cluster;
    { static int bs = -1; if (bs < 0) { const char* e = getenv("MHT_BLP_STAMPS"); bs = (e && e[0] == '1') ? 1 : 0; } b.dbg = (bs && f->debug) ? f->grow_dbg : nullptr; }
}

// similar-state pruning of scan s's children (between the cluster and the ILP kernel; tracker.py:230-231)
static void fill_similar(const Forest* f, int s, SimilarArgs& a) {
    a = SimilarArgs{};
    const int cb = s & 1;
    const mht_nodes& out = f->layer[s % f->R];
    a.single_list = f->single_list; a.counts = f->cl_counts; a.tchild = f->tchild; a.tcend = f->tcend;
    a.x = out.x; a.cnllr = out.cnllr; a.pd = out.pd; a.meas = out.meas; a.cov = out.cov; a.flags = out.flags; a.cost = f->cost2[s & 1]; a.cap = f->Ncap;
    a.t_root_cnllr = f->tab[cb].root_cnllr; a.t_root_f32 = f->tab[cb].root_f32; a.Nwin = f->cfg.n_scan;
    a.vt = f->vt;
    fill_model_only(a.model, &f->model);
    a.thr = f->prune_thr;
    a.status = f->status2 + (s & 1);
    a.mmsi = f->ais ? f->l_mmsi[s % f->R] : nullptr;
    if (f->ais) { a.t_window = f->tab[cb].window; a.t_depth = f->tab[cb].depth; }
    if (f->ct) { const int lp = (s - 1 + f->R) % f->R; a.ct_Phat = f->ct_Phat[lp]; a.ct_Pbar = f->ct_Pbar[lp]; }      // (the children's covariances live in the PARENT layer's arrays)
}

// N-scan prune (tracker.py:256-259), target side: surviving leaf ranges -> target table / roots / report, for scan s
static void fill_commit(const Forest* f, int s, CommitArgs& p) {
    p = CommitArgs{};
    const int cb = s & 1, nb = (s + 1) & 1;
    p.cur = f->tab[cb]; p.nxt = f->tab[nb];
    p.sel = f->sel; p.t_status = f->t_status; p.t_jdrop = f->t_jdrop; p.t_count = f->t_count; p.t_firstsurv = f->t_firstsurv;
    p.w_root_scan = f->w_root_scan; p.w_root_node = f->w_root_node; p.w_root_cnllr = f->w_root_cnllr; p.w_root_f32 = f->w_root_f32;
    p.R = f->R; p.cap = f->Ncap; p.Tcap = f->Tcap; p.vnext = nb;
    p.new_index = f->new_index;
    p.cnt = f->cnt; p.status = f->status2 + (s & 1);
    p.cl_counts = f->cl_counts; p.cl_status = f->cl_status; p.cl_iters = f->cl_iters; p.multi_list = f->multi_list;
    p.used_bytes = f->used_bytes[s & 1]; p.used_words = reinterpret_cast<unsigned long long*>(f->report_dev2[s & 1] + f->used_off);
    p.hdr = reinterpret_cast<ReportHeader*>(f->report_dev2[s & 1]); p.hint = f->hint_dev; p.vcount = f->vt.count;
    p.rec = reinterpret_cast<mht_target_report*>(f->report_dev2[s & 1] + f->rec_off);
    p.log = f->commit_log;
}

// Host-side bookkeeping of a step.  begin: every check that can fail comes BEFORE the scan counter moves (a refused step must not
// shift the ring / parity the later steps derive their buffers from); a launch failure after that kills the forest.
// Re-keys the live leaves of the newest layer into the other generation of the value table (see Forest::vts): one thread per leaf;
// leaves that share a key share the new one (remap[], first come first served).  The gains travel with the key (a 64-byte copy).
struct RebuildArgs { mht_nodes layer; TTable tab; const FCounts* cnt; VTab from, to; int32_t* remap; };
__global__ __launch_bounds__(256) void vt_rebuild_kernel(const RebuildArgs a) {
    const int nT = a.cnt->nT, L = a.cnt->L;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
        int lo = 0, hi = nT;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.tab.leaf_off[mid] <= i) lo = mid; else hi = mid; }
        const int nd = a.tab.first[lo] + (i - a.tab.leaf_off[lo]);
        const int k_old = a.layer.cov[nd];
        int k_new = __hip_atomic_load(&a.remap[k_old], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k_new < 0) {
            int mine;
            if (a.layer.flags[nd] & F_COV_F64) {      // a float64 value: two ids, gains in the rows of the key and of its twin (mht_vtab.h)
                double P[NP], row[GKF];
                vt_load64(a.from, a.from.child[k_old], P);
                const int id = vt_find_or_insert64(a.to, P, a.layer.pd[nd]);
                vt_load_gains64(a.from, k_old, row);
                mine = vt_pseudo_key64(a.to, id, row);
                if (*a.to.overflow) continue;
            } else {
            float P[NP];
            vt_load(a.from, a.from.child[k_old], P);
            const int id = vt_find_or_insert(a.to, P, a.layer.pd[nd]);
            const unsigned pid = atomicAdd(a.to.count, 1u);      // a pseudo parent, as for a root: its miss child is the leaf's value
            if (pid >= (unsigned)a.to.vcap) { *a.to.overflow = 1; continue; }
            mine = 2 * (int)pid;
            for (int q = 0; q < GKQ; ++q) a.to.Gk[(size_t)mine * GKQ + q] = a.from.Gk[(size_t)k_old * GKQ + q];
            a.to.child[mine] = id;
            }
            int expected = -1;
            k_new = __hip_atomic_compare_exchange_strong(&a.remap[k_old], &expected, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? mine : expected;
        }
        a.layer.cov[nd] = k_new;
    }
}

struct StepPlan { int s; bool fused; int n_ub; int W; bool rebuilt; };
// Switches the forest to the other generation of its value table in front of scan s (the newest layer is s - 1).
static int vt_switch_generation(mht_ctx* ctx, Forest* f, int s) {
    { const int rc = flush_commit(ctx, f); if (rc) return rc; }      // (the leaf ranges of the committed table are what is re-keyed)
    const int other = 1 - f->vgen;
    VTab& to = f->vts[other];
    MHT_HIP_CHECK(hipMemsetAsync(to.child, 0xff, (size_t)2 * to.vcap * sizeof(int32_t), ctx->stream));
    MHT_HIP_CHECK(hipMemsetAsync(to.slots, 0, ((size_t)to.hmask + 1) * sizeof(unsigned long long), ctx->stream));
    MHT_HIP_CHECK(hipMemsetAsync(to.count, 0, 16 * sizeof(unsigned), ctx->stream));
    MHT_HIP_CHECK(hipMemsetAsync(f->vt_remap, 0xff, (size_t)2 * to.vcap * sizeof(int32_t), ctx->stream));
    RebuildArgs a = {f->layer[(s - 1) % f->R], f->tab[s & 1], f->cnt, f->vts[f->vgen], to, f->vt_remap};
    hipLaunchKernelGGL(vt_rebuild_kernel, dim3(256), dim3(256), 0, ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    f->vgen = other;
    f->vt = f->vts[other];
    f->layer_gen[(s - 1) % f->R] = other;
    f->last_rebuild_scan = s;
    f->rebuilds += 1;
    return MHT_OK;
}
static int forest_begin_step(mht_ctx* ctx, Forest* f, const float* z, int M, const char* who, StepPlan& pl, bool carries_admission = false) {
    MHT_REQUIRE(M >= 0 && M <= f->cfg.max_meas, "%s: M=%d exceeds max_meas=%d", who, M, f->cfg.max_meas);
    MHT_REQUIRE(z || M == 0, "%s: z is null", who);
    if (f->dead) {
        set_error("%s: the forest is dead (a pool overflowed or a launch failed in an earlier scan); create a new one with larger max_nodes / max_targets", who);
        return MHT_E_STATE;
    }
    if (f->timing) MHT_REQUIRE(f->timed_steps < EV_POOL, "%s: %d timed steps pending, read them with mht_forest_stage_times", who, EV_POOL);
    if (f->adm_pending && !carries_admission) { const int rc = flush_commit(ctx, f); if (rc) return rc; }      // (only the one-sector grow launch takes the admission along)
    const int s = ++f->scan;
    pl.rebuilt = false;
    if (f->hint_host) {      // value table three quarters full (as of the last commit the host has seen) and the other generation free again?
        const unsigned long long used = reinterpret_cast<volatile unsigned long long*>(f->hint_host)[1];
        // ids are consumed by new covariance values AND by pseudo parents (roots, births, re-keyed leaves, merged hypotheses of
        // similar-state pruning): the switch comes when the table is three quarters full, or earlier when, at the largest per-scan
        // consumption seen so far, what is left would not last through the R + 2 scans until the NEXT switch is allowed (twice over:
        // the hint lags a scan or two behind the device)
        if (used > f->vt_used_seen) { const unsigned long long dlt = used - f->vt_used_seen; if (dlt > f->vt_rate) f->vt_rate = dlt; }
        if (used) f->vt_used_seen = used;
        const unsigned long long need = 2ull * (unsigned long long)(f->R + 3) * f->vt_rate;
        const bool tight = used > (unsigned long long)f->vt.vcap / 4 * 3 || (used > (unsigned long long)f->vt.vcap / 8 && used + need > (unsigned long long)f->vt.vcap);
        if (tight && s - f->last_rebuild_scan > f->R + 1) {
            const int rc = vt_switch_generation(ctx, f, s);
            if (rc) { f->dead = true; return rc; }
            reinterpret_cast<volatile unsigned long long*>(f->hint_host)[1] = 0;      // (until the next commit reports the new table's fill)
            f->vt_used_seen = 0;
            f->vt_rate /= 2;      // (the re-keying itself is a burst: let the estimate decay)
            pl.rebuilt = true;
        }
    }
    f->layer_gen[s % f->R] = f->vgen;
    f->births_after[s % 64] = 0; f->births_init_ub[s % 64] = 0;
    f->nT_ub_prev = f->nT_ub_step;      // slots of the table the previous scan ran on
    if (s > 1) { const int again = f->targets_ub(s - 1); if (again < f->nT_ub_prev) f->nT_ub_prev = again; }      // (bounded again: the device's hints have moved on since)
    f->nT_ub_step = f->targets_ub(s);
    f->births_since_step = 0;
    f->last_M = M;
    pl.s = s;
    pl.fused = f->commit_pending;      // the previous scan's commit rides in this scan's grow launch
    // one workgroup per slot of the table the scan runs on: the uncommitted one (targets before the last scan's terminations) when
    // the commit rides along, else the committed one
    pl.n_ub = pl.fused ? f->nT_ub_prev : f->nT_ub_step;
    pl.W = (M + 63) / 64;
    return MHT_OK;
}
static void forest_end_step(Forest* f, const StepPlan& pl, int M) {
    // the target-side commit of this scan is deferred: workgroup 0 of the next scan's grow launch runs it, unless somebody needs
    // the committed state before that (flush_commit).  One launch and one kernel boundary less per scan; with timing on, its time
    // shows up in the next scan's "gate" stage and the "prune" stage reads zero.
    fill_commit(f, pl.s, f->pending);
    f->pending_dyn = CommitDyn{pl.s, M, pl.W};
    f->commit_pending = true;
    f->report_pending = true;
    f->L_ub = f->Ncap;        // unknown until the report is fetched
}

}  // namespace mht

namespace mht {
void initiator_scan_args(mht_initiator* in, const float* z, int M, const unsigned long long* used, double now, InitArgs& a);
void initiator_born_ptrs(const mht_initiator* in, const double** x, const float** P, const uint8_t** fl, const double** pd, const int32_t** meas,
                         const int32_t** n, int* cap, mht_ctx** ctx);
int initiator_ais_pending(const mht_initiator* in);
int initiator_mreq(const mht_initiator* in);
void initiator_ais_ptrs(mht_initiator* in, const AisInitMsg** msgs, unsigned char** used);
}

// will a step with this initiator take the union-find path with the initiator as a launch of its own (forest_step_impl: use_uf)?
static bool forest_streams_uf(const Forest* f, const mht_initiator* init) {
    return f->uf_ok && !(f->prune_thr > 0.f) && (!init || (f->adm_fuse && !f->ais && !f->timing && !f->init_side));
}
// a reader of the staged scan other than the grow launch / the initiator's launch is about to be queued on the ctx stream: the event wait
// the step skipped (step_host_impl)
static int flush_z_wait(mht_ctx* ctx, Forest* f) {
    if (f->z_wait_slot >= 0) {      // (a streamed scan records no event of its own: one behind everything the side stream holds now)
        MHT_HIP_CHECK(hipEventRecord(f->z_ev[f->z_wait_slot], f->stage_stream));
        MHT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, f->z_ev[f->z_wait_slot], 0));
        f->z_wait_slot = -1;
    }
    return MHT_OK;
}
// development: where the host time of a streamed scan goes (MHT_HOST_PROF=1: per-section means on stderr when the forest is destroyed)
struct HostProf { bool on = false; bool asked = false; double acc[16] = {}; long n = 0; double t_last = 0.0; };
static HostProf g_hp;
static inline double hp_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
static inline void hp_begin() { if (!g_hp.asked) { g_hp.asked = true; const char* e = getenv("MHT_HOST_PROF"); g_hp.on = e && e[0] == '1'; } if (g_hp.on) { g_hp.t_last = hp_now(); g_hp.n += 1; } }
static inline void hp_mark(int i) { if (g_hp.on) { const double t = hp_now(); g_hp.acc[i] += t - g_hp.t_last; g_hp.t_last = t; } }
static void hp_print() {
    if (!g_hp.on || !g_hp.n) return;
    static const char* nm[16] = {"checks", "stage (memcpy, kernel, events)", "begin_step + deferred init", "fill grow args", "grow launch", "events behind grow", "fill blp", "blp launch", "initiator args / defer", "end_step", "initiate_impl (ride)", "", "", "", "", ""};
    fprintf(stderr, "[mht host prof] %ld scans, us per scan:", g_hp.n);
    for (int i = 0; i < 11; ++i) fprintf(stderr, " %s %.2f |", nm[i], g_hp.acc[i] / g_hp.n);
    fprintf(stderr, "\n");
    g_hp = HostProf();
}
// AIS forest: forest_ais_kernel (mht_ais.hip) over the leaves of the committed table, in front of scan s's grow launch; disarms the messages
static int forest_ais_prepass(mht_ctx* ctx, Forest* f, int s, int n_ub, const float* z, int M) {
    AisForestArgs aa = {};
    const mht_nodes& in = f->layer[(s - 1) % f->R];
    fill_model_only(aa.model, &f->model);
    aa.nT_dev = &f->cnt->nT; aa.t_first = f->tab[s & 1].first; aa.t_leaf_off = f->tab[s & 1].leaf_off;
    aa.x = in.x; aa.pd = in.pd; aa.cov = in.cov; aa.flags = in.flags; aa.hmmsi = f->l_hmmsi[(s - 1) % f->R]; aa.cap = f->Ncap;
    aa.vt = f->vt;
    aa.groups = reinterpret_cast<const AisGroup*>(f->ais_groups_dev); aa.nG = f->ais_nG; aa.msgs = reinterpret_cast<const AisMsg*>(f->ais_msgs_dev);
    aa.eta2_ais = f->ais_eta2; aa.lambda_ais = f->ais_lambda; aa.z = z; aa.M = M;
    aa.nf = f->ais_nf; aa.off = f->ais_off; aa.rec = f->ais_rec; aa.rec_cap = f->ais_rec_cap; aa.rec_count = f->ais_count;
    aa.status = f->status2 + (s & 1);
    if (hipMemsetAsync(f->ais_count, 0, sizeof(unsigned), ctx->stream) != hipSuccess) { set_error("forest_ais_prepass: hipMemsetAsync failed"); return MHT_E_HIP; }
    const int rc = launch_forest_ais(ctx, aa, n_ub);
    f->ais_armed = false;
    return rc;
}
// constant-turn forest: forest_ct_kernel (mht_ais.hip) over the leaves of the committed table, in front of scan s's grow launch
static int forest_ct_prepass(mht_ctx* ctx, Forest* f, int s, int n_ub, bool fused = false) {
    CtForestArgs ca = {};
    const int li = (s - 1) % f->R, lp = (s - 2 + f->R) % f->R;
    const mht_nodes& in = f->layer[li];
    fill_model_only(ca.model, &f->model); ca.T = f->ct_T;
    ca.nT_dev = &f->cnt->nT; ca.t_first = f->tab[s & 1].first; ca.t_leaf_off = f->tab[s & 1].leaf_off;
    if (fused) { ca.fused = 1; ca.nT_dev = &f->cnt->nTv[(s - 1) & 1]; ca.p_status = f->t_status; ca.p_count = f->t_count; ca.p_firstsurv = f->t_firstsurv; }      // (as fill_fgrow)
    ca.x = in.x; ca.pd = in.pd; ca.cov = in.cov; ca.flags = in.flags; ca.cap = f->Ncap;
    ca.Pbar_prev = f->ct_Pbar[lp]; ca.Phat_prev = f->ct_Phat[lp]; ca.Proot = f->ct_Proot[li];
    ca.Pbar = f->ct_Pbar[li]; ca.Phat = f->ct_Phat[li];
    ca.gains = f->ct_gains; ca.xbar = f->ct_xbar; ca.zhat = f->ct_zhat;
    return launch_forest_ct(ctx, ca, n_ub);
}
// init != null (mht_forest_scan): the scan's step 7 rides in the cluster launch (cluster_init_kernel)
static int forest_step_impl(mht_ctx* ctx, const float* z, int32_t M, mht_initiator* init, double now) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_step: no forest");
    Forest* f = ctx->forest;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    const bool ais = f->ais && f->ais_armed;
    // (a refused step must not leave the messages armed: the tracker stays alive after MHT_E_INVALID, and the next accepted scan -- maybe
    // one without messages -- would consume them with this scan's time steps)
    if (ais && !(M + f->ais_nA <= f->Mpad)) f->ais_armed = false;
    if (ais) {      // the fused children are made from the leaves of the COMMITTED table, in front of the grow launch
        MHT_REQUIRE(M + f->ais_nA <= f->Mpad, "mht_forest_step: %d radar measurements + %d AIS messages exceed max_meas=%d (rounded up to %d measurement nodes per scan)",
                    M, f->ais_nA, f->cfg.max_meas, f->Mpad);
        const int rc = flush_commit(ctx, f);
        if (rc) return rc;
    }
    // (constant-turn forest: forest_ct_kernel walks the leaves in front of the grow launch -- of the uncommitted table when the previous scan's
    // commit is still pending (it then rides in the grow launch like everywhere else: one launch and two boundaries less per scan), MHT_CT_FLUSH=1:
    // of the committed table, as until the end of round 5)
    static int ct_flush = -1; if (ct_flush < 0) { const char* e = getenv("MHT_CT_FLUSH"); ct_flush = (e && e[0] == '1') ? 1 : 0; }
    if (f->ct && ct_flush) {
        const int rc = flush_commit(ctx, f);
        if (rc) return rc;
    }
    StepPlan pl;
    { const int rc = forest_begin_step(ctx, f, z, M, "mht_forest_step", pl, !ais); if (rc) { f->ais_armed = false; return rc; } }
    if (ais) pl.W = (M + f->ais_nA + 63) / 64;      // (the messages are measurement nodes M .. M + nA - 1 of this scan)
    hipStream_t st = ctx->stream;
    hipEvent_t* ev = nullptr;
    if (f->timing) {
        ev = f->evp[f->ev_slot];
        f->ev_slot = (f->ev_slot + 1) % EV_POOL;
    }
    // No memsets between scans: the used-measurement bytes are cleared by the commit, the other parity's status word, the edge
    // and child counters and the cluster counters by the cluster kernel.
#define MHT_STEP_CHECK(expr) do { const int rc_ = (expr); if (rc_) { f->dead = true; return rc_; } } while (0)
#define MHT_STEP_HIP(expr) do { if ((expr) != hipSuccess) { f->dead = true; set_error("mht_forest_step: %s failed", #expr); return MHT_E_HIP; } } while (0)
    if (f->timing) MHT_STEP_HIP(hipEventRecord(ev[0], st));
    // ---- 0: AIS-aided children of every leaf (tracker.py:394-396, :417-552), only on scans that carry messages ----------------
    if (ais) MHT_STEP_CHECK(forest_ais_prepass(ctx, f, pl.s, pl.n_ub, z, M));
    if (f->ct) MHT_STEP_CHECK(forest_ct_prepass(ctx, f, pl.s, pl.n_ub, pl.fused));      // ---- 0': the leaves' own transitions, predictions, gains and children covariances
    // Clusters without a clustering launch (mht_kernels.h: FDyn::uf_epoch): the target workgroups of the grow launch hook their targets
    // into a union-find, the workgroups of the ILP launch derive the cluster tables from it.  Similar-state pruning works on the
    // clustering kernel's list of lone targets between the two launches; the streamed path's initiator rides in the cluster launch.
    // (streamed path: the scan's initiator then runs as a launch of its own NEXT to the ILP launch -- launched any-order behind it)
    const bool use_uf = forest_streams_uf(f, init);
    bool grow_ovl = false;      // this scan's grow launch took the previous scan's results target by target (FDyn::ovl)
    // ---- 1: grow every leaf (tracker.py:207-209) ---------------------------------------------------------------
    {
        hp_mark(2);
        FGrowArgs g;
        fill_fgrow(f, pl.s, pl.fused, g);
        FDyn d = {};
        d.z = z; d.M = M; d.W = pl.W; d.c_scan = f->pending_dyn.scan; d.c_M = f->pending_dyn.M; d.c_W = f->pending_dyn.W;
        d.ais_on = ais ? 1 : 0;
        d.maybe_dead = (f->similar_ran_scan == pl.s - 1);
        d.dbg = f->debug ? f->grow_dbg : nullptr;
        d.uf_epoch = use_uf ? 2u * (unsigned)pl.s : 0u;      // (2 x scan: the odd value in between is the epoch of a union-find that had to be redone, mht_fgrow.hip)
        // the previous scan's ILP launch published per-target records and its commit rides here: this launch takes what it needs of that
        // launch target by target -- and may start while it is still running
        d.ovl = (pl.fused && f->pub_scan == pl.s - 1 && f->pending_dyn.scan == pl.s - 1) ? 1 : 0;
        d.c_wait = f->blp_done_total;
        grow_ovl = d.ovl != 0;
        d.ct_spill = f->ct_spill ? 1 : 0;
        d.z_flag = &f->cnt->z_flag; d.z_tag = f->z_tag_step;
        { static int os = -1; if (os < 0) { const char* e = getenv("MHT_OVL_STAMPS"); os = (e && e[0] == '1') ? 1 : 0; } d.stamp_end = os; }
        const bool adm = f->adm_pending && pl.fused;      // (flush_commit clears both)
        static int ovl_force = -1; if (ovl_force < 0) { const char* e = getenv("MHT_OVL_FORCE"); ovl_force = (e && e[0] == '1') ? 1 : 0; }      // (development: any-order launches with the debug stamps on)
        // (the streamed path's launch -- commit, admission of the initiator's births, report push -- overlaps too when that initiator posts its flag)
        const bool adm_ovl = adm && f->init_flag_scan == pl.s - 1;
        d.adm_wait = adm_ovl ? 1 : 0;
        const bool any_order = d.ovl && f->ovl_ok && (!adm || adm_ovl) && (!f->pub_deferred || adm_ovl) && !ais && !f->timing && (!f->debug || ovl_force);
        if (any_order) f->ovl_launches += 1;
        hp_mark(3);
        MHT_STEP_CHECK(launch_deferred_init(ctx, f));      // (the previous scan's initiator: behind this scan's staging kernel, in front of this launch)
        if (adm && f->init_ev_pending) {
            if (adm_ovl && f->init_ev_lazy) { f->init_ev_pending = false; f->init_ev_lazy = false; }      // (the admission waits for the initiator's flag itself)
            else MHT_STEP_CHECK(wait_init_ev(ctx, f));
        }
        const bool pub_flag = f->pub_deferred && adm && f->pub_args.dst && f->rep_flag_ok && !f->serial_prof;      // (fgrow_adm_kernel pushes it: the host polls the pushing workgroups' words)
        if (pub_flag) {
            f->pub_args.done = reinterpret_cast<unsigned long long*>(f->report_host_dev[f->pub_slot] + f->done_off);
            f->pub_args.tag = (unsigned long long)(pl.s - 1) | (1ull << 40);
        } else { f->pub_args.done = nullptr; f->pub_args.tag = 0; }
        MHT_STEP_CHECK(launch_fgrow(ctx, g, d, pl.n_ub, pl.fused ? &f->pending : nullptr, f->pub_deferred ? &f->pub_args : nullptr, adm ? &f->adm : nullptr, any_order));
        hp_mark(4);
        f->adm_pending = false;
        if (f->z_guard_due >= 0) { MHT_STEP_HIP(hipEventRecord(f->z_guard_ev[f->z_guard_due], st)); f->z_guard_due = -1; }      // (step_host_impl: consumer guard of the staging ring)
        if (f->pub_deferred) {      // the previous scan's report went along: the host waits for this launch
            if (pub_flag) { f->rep_by_flag[f->pub_slot] = true; f->rep_tag[f->pub_slot] = f->pub_args.tag; }
            else { MHT_STEP_HIP(hipEventRecord(f->rep_ev[f->pub_slot], st)); f->rep_by_flag[f->pub_slot] = false; }
            f->rep_started[f->pub_slot] = true;
            f->host_block_scan[f->pub_slot] = pl.s - 1;
            f->pub_deferred = false;
        }
    }
    f->commit_pending = false;
    hp_mark(5);
    if (f->timing) MHT_STEP_HIP(hipEventRecord(ev[1], st));
    // ---- 2: cluster (tracker.py:218-221) ---------------------------------------------------------------------------
    if (use_uf) f->uf_scans += 1;
    else {
        ClusterArgs c;
        fill_cluster(f, pl.s, c);
        if (init && !f->cluster_big) {
            InitArgs ia;
            initiator_scan_args(init, z, M, nullptr, now, ia);
            ia.used_b = f->used_bytes[pl.s & 1];      // (written by this scan's grow launch, packed and cleared by its commit later)
            ia.bhint = f->bhint_dev; ia.forest_overflow = &f->cnt->overflow; ia.scan_no = pl.s;
            if (f->init_side && f->stage_stream && f->adm_fuse && !f->ais && !f->timing) {
                MHT_STEP_HIP(hipEventRecord(f->grow_ev, st));                       // behind the grow launch
                MHT_STEP_HIP(hipStreamWaitEvent(f->stage_stream, f->grow_ev, 0));
                hipLaunchKernelGGL(initiator_side_kernel, dim3(1), dim3(INIT_THREADS), 0, f->stage_stream, ia, static_cast<const DevStatus*>(c.status), static_cast<const int32_t*>(&f->cnt->overflow));
                MHT_STEP_HIP(hipGetLastError());
                MHT_STEP_HIP(hipEventRecord(f->init_ev, f->stage_stream));
                f->init_ev_pending = true;
                MHT_STEP_CHECK(launch_cluster(ctx, c));
            } else {
                MHT_STEP_CHECK(launch_cluster(ctx, c, &ia, &f->cnt->overflow));
            }
            f->init_ran_scan = pl.s;
        } else {      // (no initiator, or the HBM-table clustering kernel: the initiator then runs behind the scan, in post_scan_kernel)
            MHT_STEP_CHECK(launch_cluster(ctx, c));
        }
    }
    if (f->timing) MHT_STEP_HIP(hipEventRecord(ev[2], st));
    // ---- 3: global hypothesis per cluster (tracker.py:225-237) + per-target termination / prune decision ---------------
    if (f->prune_thr > 0.f) {
        SimilarArgs sa;
        fill_similar(f, pl.s, sa);
        MHT_STEP_CHECK(launch_prune_similar(ctx, sa, f->nT_ub_step));
        f->similar_ran_scan = pl.s;
    }
    {
        BlpArgs b;
        fill_blp(f, pl.s, b);
        b.uf_epoch = use_uf ? 2u * (unsigned)pl.s : 0u;
        b.ni_flag = &f->cnt->ni_flag; b.uf_ovl = grow_ovl ? 1 : 0;
        int grid = f->nT_ub_step / 2 + 8;
        if (grid > 1024) grid = 1024;
        // Clusters from the union-find: a workgroup per multi-target cluster and a wavefront per single-target one is all the launch needs
        // (the loops of blp_body take more of either) -- every further workgroup runs the prologue for nothing and loads the fabric the
        // others work through: at the headline size 160-192 workgroups instead of 258 are 1 us per scan (profiles/r05_merge_ab.txt, "ILP grid").
        // The commit leaves the last scan's counts in the host-mapped hint block; a scan with a team-sized cluster keeps the workgroups
        // without a cluster (they are the teams).
        // (a wall-clock budget per cluster is set: the width of a team -- and with it which non-proven incumbent a budget-limited search returns -- follows
        // the grid, so the grid stays the full one and does not depend on when the host happened to read the hint word)
        if (use_uf && f->hint_host && f->grid_by_hint && !(f->blp_time_limit > 0 && f->nT_ub_step >= TEAM_MIN_K)) {
            const unsigned long long hh = reinterpret_cast<volatile unsigned long long*>(f->hint_host)[2];
            const int h_scan = (int)(hh >> 48), h_multi = (int)((hh >> 32) & 0xffffu), h_single = (int)((hh >> 8) & 0xffffffu), h_team = (int)(hh & 0xffu);
            const int age = (pl.s - h_scan) & 0xffff;
            // (a host that queues scans far ahead of the device -- the replay -- sees counts that are hundreds of scans old: a stationary
            // stream's statistics; what the launch is too small for, it loops over)
            if (hh != 0ull && age >= 1 && age <= 8192 && h_team == 0) {
                int gh = h_multi + h_multi / 8 + (h_single + 3) / 4 + 8;      // (an eighth more multi-target clusters than last time; the rest loops)
                if (gh < 32) gh = 32;
                // (the hint may be thousands of scans old and targets may have been added since: a workgroup's tables hold 4 multi-target
                // clusters and 32 single-target ones -- UfPersist::own / single -- so THIS scan's target bound keeps the grid from below)
                const int g_min = f->nT_ub_step / 8 + 1;
                if (gh < g_min) gh = g_min;
                if (gh < grid) grid = gh;
            }
        }
        if (use_uf) {
            b.rec0 = f->rec0; b.blp_done = &f->cnt->blp_done; b.pub_scan = (unsigned)pl.s; b.begun = &f->cnt->ilp_begun;
            b.pub_ub = f->nT_ub_step < 1 ? 1 : (f->nT_ub_step < f->Tcap ? f->nT_ub_step : f->Tcap);      // (the next grow launch has one target workgroup at least)
        }
        hp_mark(6);
        MHT_STEP_CHECK(launch_blp(ctx, b, grid));
        hp_mark(7);
        if (use_uf) { f->pub_scan = pl.s; f->blp_done_total += (unsigned long long)grid; }
        if (use_uf && init) {
            // step 7 (tracker.py:264-278) needs the scan and the used-measurement bytes of the grow launch, nothing of the ILPs: one
            // workgroup, launched any-order behind the ILP launch -- it runs next to the ILPs' tail; what it gives birth to is admitted in
            // the next scan's grow launch (which waits for both)
            InitArgs ia;
            initiator_scan_args(init, z, M, nullptr, now, ia);
            ia.used_b = f->used_bytes[pl.s & 1];
            ia.bhint = f->bhint_dev; ia.forest_overflow = &f->cnt->overflow; ia.scan_no = pl.s;
            if (f->init_side_q && f->stage_stream && f->z_tag_step) {      // (launch_deferred_init)
                f->init_deferred = true; f->init_def_args = ia; f->init_def_status = f->status2 + (pl.s & 1); f->init_def_ztag = f->z_tag_step;
                f->init_ev_pending = true; f->init_ev_lazy = true;      // (whoever needs the births without the flag: init_ev, recorded then)
            } else
            hipExtLaunchKernelGGL(initiator_side_kernel, dim3(1), dim3(INIT_THREADS), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, ia,
                                  static_cast<const DevStatus*>(f->status2 + (pl.s & 1)), static_cast<const int32_t*>(&f->cnt->overflow), &f->cnt->init_flag,
                                  static_cast<const unsigned long long*>(f->z_tag_step ? &f->cnt->z_flag : nullptr), f->z_tag_step,
                                  static_cast<unsigned long long*>(nullptr), static_cast<const unsigned long long*>(nullptr), static_cast<unsigned long long*>(nullptr));      // (no ticket: one workgroup)
            MHT_STEP_HIP(hipGetLastError());
            f->init_ran_scan = pl.s; f->init_flag_scan = pl.s;
        }
    }
    hp_mark(8);
    if (f->timing) MHT_STEP_HIP(hipEventRecord(ev[3], st));
    // ---- 4: N-scan prune (tracker.py:256-259), target side: deferred ------------------------------------------------------
    forest_end_step(f, pl, M);
    hp_mark(9);
    if (f->timing) { MHT_STEP_HIP(hipEventRecord(ev[4], st)); f->timed_steps += 1; }
    return MHT_OK;
}
extern "C" int mht_forest_step(mht_ctx* ctx, const float* z, int32_t M) { return forest_step_impl(ctx, z, M, nullptr, 0.0); }

// ---- cluster-sharded step: ONE tracker on several devices (north star: "independent track clusters shard across the GPUs ... when
// the gating graph actually partitions") ---------------------------------------------------------------------------------------
// Every device holds the same forest and is fed the same scans.  Grow and clustering are replicated (17 + 9 us at the headline
// size: less than moving a layer between devices would cost); the 0-1 ILPs -- independent per cluster, tracker.py:228-236 -- are
// spread by size (cluster kernel: cl_owner, longest-processing-time first on the column counts; a single-target cluster by t % n).  The selections
// travel as child ordinals inside each target's block (sel_rel, [max_targets] int32 in caller-owned device memory, -1 = "not mine"):
// one all-reduce(MAX) over them between _begin and _end gives every device every selection; _end then runs the per-target end of
// the scan (termination, N-scan pruning) for all targets, so the forests stay identical.  A gating graph that is ONE component is
// solved by one device while the others wait: the one-GPU fallback the north star asks for.
// A gating graph that is ONE big component (mht_forest_step_sharded_begin2 with an exchange block of mht_forest_sharded_words words): the clusters of
// the team list (>= TEAM_MIN_K targets, at most TEAM_MAX per scan) are searched by ALL devices -- the subtrees of the branch and bound are dealt out over
// the members of every device's team (Team::qg / Wg), every device files its best selection and its value in its own slots of the block, the
// all-reduce(MAX) that merges the selections gathers the files as well (empty slots are -1), and _end2 lets the smallest value win on every device
// alike.  tracker.py:1155-1217 is one CBC call: any exact split is acceptable.
extern "C" int mht_forest_sharded_words(mht_ctx* ctx, int32_t shard_n, int32_t* n_words) {
    MHT_REQUIRE(ctx && ctx->forest && n_words && shard_n >= 1, "mht_forest_sharded_words: bad argument");
    *n_words = ctx->forest->Tcap + shard_n * TEAM_MAX * XT_WORDS;
    return MHT_OK;
}

static int sharded_begin_impl(mht_ctx* ctx, const float* z, int32_t M, int32_t shard_n, int32_t shard_i, int32_t* sel_rel, int32_t n_words);
extern "C" int mht_forest_step_sharded_begin(mht_ctx* ctx, const float* z, int32_t M, int32_t shard_n, int32_t shard_i, int32_t* sel_rel) {
    return sharded_begin_impl(ctx, z, M, shard_n, shard_i, sel_rel, 0);
}
extern "C" int mht_forest_step_sharded_begin2(mht_ctx* ctx, const float* z, int32_t M, int32_t shard_n, int32_t shard_i, int32_t* xch, int32_t n_words) {
    MHT_REQUIRE(ctx && ctx->forest && n_words >= ctx->forest->Tcap + shard_n * TEAM_MAX * XT_WORDS,
                "mht_forest_step_sharded_begin2: the exchange block needs mht_forest_sharded_words() words");
    return sharded_begin_impl(ctx, z, M, shard_n, shard_i, xch, n_words);
}
static int sharded_begin_impl(mht_ctx* ctx, const float* z, int32_t M, int32_t shard_n, int32_t shard_i, int32_t* sel_rel, int32_t n_words) {
    MHT_REQUIRE(ctx && ctx->forest && sel_rel, "mht_forest_step_sharded_begin: null argument");
    MHT_REQUIRE(shard_n >= 1 && shard_i >= 0 && shard_i < shard_n, "mht_forest_step_sharded_begin: bad shard %d of %d", shard_i, shard_n);
    Forest* f = ctx->forest;
    MHT_REQUIRE(!f->timing, "mht_forest_step_sharded_begin: per-stage timing is not available for sharded steps");
    MHT_REQUIRE(!f->shard_open, "mht_forest_step_sharded_begin: the previous sharded step has not been ended");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    { const int rc = flush_publish(ctx, f); if (rc) return rc; }
    StepPlan pl;
    // (AIS messages: the fused children are made on every shard, like grow and clustering -- forest_ais_kernel walks the leaves of the COMMITTED
    // table, as forest_ct_kernel does)
    const bool ais = f->ais && f->ais_armed;
    if (ais && !(M + f->ais_nA <= f->Mpad)) {
        f->ais_armed = false;
        set_error("mht_forest_step_sharded_begin: %d radar measurements + %d AIS messages exceed max_meas=%d", M, f->ais_nA, f->cfg.max_meas);
        return MHT_E_INVALID;
    }
    if (f->ct || ais) { const int rcf = flush_commit(ctx, f); if (rcf) return rcf; }
    { const int rc = forest_begin_step(ctx, f, z, M, "mht_forest_step_sharded_begin", pl); if (rc) { f->ais_armed = false; return rc; } }
    if (ais) pl.W = (M + f->ais_nA + 63) / 64;
    int rc;
    if (ais) { rc = forest_ais_prepass(ctx, f, pl.s, pl.n_ub, z, M); if (rc) { f->dead = true; return rc; } }
    if (f->ct) { rc = forest_ct_prepass(ctx, f, pl.s, pl.n_ub); if (rc) { f->dead = true; return rc; } }
    {
        FGrowArgs g;
        fill_fgrow(f, pl.s, pl.fused, g);
        FDyn d = {};
        d.z = z; d.M = M; d.W = pl.W; d.c_scan = f->pending_dyn.scan; d.c_M = f->pending_dyn.M; d.c_W = f->pending_dyn.W;
        d.ais_on = ais ? 1 : 0;
        d.maybe_dead = (f->similar_ran_scan == pl.s - 1);
        rc = launch_fgrow(ctx, g, d, pl.n_ub, pl.fused ? &f->pending : nullptr);
    }
    f->commit_pending = false;
    if (!rc) {
        ClusterArgs c;
        fill_cluster(f, pl.s, c);
        c.sel_rel_reset = sel_rel;
        c.shard_n = shard_n; c.tchild = f->tchild; c.tcend = f->tcend; c.cl_owner = f->cl_owner;
        rc = launch_cluster(ctx, c);
    }
    if (!rc && f->prune_thr > 0.f) {      // (replicated, like grow and clustering: every device prunes every lone target)
        SimilarArgs sa;
        fill_similar(f, pl.s, sa);
        rc = launch_prune_similar(ctx, sa, f->nT_ub_step);
        f->similar_ran_scan = pl.s;
    }
    if (!rc) {
        BlpArgs b;
        fill_blp(f, pl.s, b);
        b.shard_n = shard_n; b.shard_i = shard_i; b.sel_rel = sel_rel; b.cl_owner = f->cl_owner;
        b.t_alive = nullptr;      // solve only: the per-target end of the scan follows the exchange (mht_forest_step_sharded_end)
        if (n_words > 0 && shard_n > 1) {      // teams across the devices: the files' slots start empty
            b.shard_team = sel_rel + f->Tcap;
            MHT_HIP_CHECK(hipMemsetAsync(b.shard_team, 0xff, (size_t)shard_n * TEAM_MAX * XT_WORDS * sizeof(int32_t), ctx->stream));
        }
        int grid = f->nT_ub_step / 2 + 8;
        if (grid > 1024) grid = 1024;
        rc = launch_blp(ctx, b, grid);
    }
    if (rc) { f->dead = true; return rc; }
    f->shard_open = true; f->shard_plan_s = pl.s; f->shard_plan_W = pl.W; f->shard_M = M; f->shard_xn = (n_words > 0 && shard_n > 1) ? shard_n : 0;
    return MHT_OK;
}

extern "C" int mht_forest_step_sharded_end(mht_ctx* ctx, const int32_t* sel_rel) {
    MHT_REQUIRE(ctx && ctx->forest && sel_rel, "mht_forest_step_sharded_end: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->shard_open, "mht_forest_step_sharded_end: no sharded step is open");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    BlpArgs b;
    fill_blp(f, f->shard_plan_s, b);
    b.sel_rel = const_cast<int32_t*>(sel_rel);
    int rc = MHT_OK;
    if (f->shard_xn > 1) {      // (begun with mht_forest_step_sharded_begin2: the winners of the teams across the devices first)
        b.shard_team = b.sel_rel + f->Tcap;
        rc = launch_shard_team_resolve(ctx, b, f->shard_xn);
    }
    if (!rc) rc = launch_blp_epilogue(ctx, b, &f->cnt->nT, f->nT_ub_step);
    f->shard_open = false;
    if (rc) { f->dead = true; return rc; }
    StepPlan pl = {};
    pl.s = f->shard_plan_s; pl.W = f->shard_plan_W;
    forest_end_step(f, pl, f->shard_M);
    return MHT_OK;
}

// ---- a group of sectors: BASELINE config 4 (independent sensor sectors = independent Tracker instances) on ONE device ----------
// S forests step together with ONE launch per stage (blockIdx.y = sector): the single-sector path is a chain of dependent round
// trips that leaves most of the chip idle, S sectors cost about one such chain.  The argument blocks of every member and every
// phase of the ring are written to HBM once, here.
struct mht_group {
    int n = 0, period = 0;
    mht_ctx* ctx[GROUP_MAX] = {};
    FGrowArgs* ga = nullptr;      // [n][period][2]
    CommitArgs* ca = nullptr;     // [n][period]
    ClusterArgs* cl = nullptr;    // [n][2]
    BlpArgs* bl = nullptr;        // [n][period][2]: LDS tier 1 (small footprint), tier 2 (default footprint, what tier 1 left)
    BlpArgs* bl0 = nullptr;       // [n][period]: one launch, default footprint
    size_t blp_lds[3] = {0, 0, 0};
    bool two_tier = false;        // development: MHT_BLP_TWO_TIER=1
    bool light = false;           // ILPs of a group: wavefront-per-cluster round-0 pass first (mht_blp.hip: blp_light), the full solver for what it leaves
    bool counted = false;         // the members' in_groups counters include this group
    bool wave = true;             // grow launch: wavefront per target (MHT_FG_WAVE=0: the workgroup-per-target kernel of the one-sector launch)
};

extern "C" int mht_group_destroy(mht_group* g) {
    if (!g) return MHT_OK;
    if (g->n > 0) {
        (void)hipSetDevice(g->ctx[0]->device);
        (void)hipStreamSynchronize(g->ctx[0]->stream);
    }
    for (int i = 0; i < g->n; ++i) if (g->counted && g->ctx[i] && g->ctx[i]->forest) g->ctx[i]->forest->in_groups -= 1;
    if (g->ga) (void)hipFree(g->ga);
    if (g->ca) (void)hipFree(g->ca);
    if (g->cl) (void)hipFree(g->cl);
    if (g->bl) (void)hipFree(g->bl);
    if (g->bl0) (void)hipFree(g->bl0);
    delete g;
    return MHT_OK;
}

extern "C" int mht_group_create(mht_group** out, int32_t n, mht_ctx* const* ctxs) {
    MHT_REQUIRE(out && ctxs && n >= 1 && n <= GROUP_MAX, "mht_group_create: need 1 <= n <= %d contexts", GROUP_MAX);
    for (int i = 0; i < n; ++i) {
        MHT_REQUIRE(ctxs[i] && ctxs[i]->forest, "mht_group_create: context %d has no forest", i);
        MHT_REQUIRE(!ctxs[i]->forest->ct, "mht_group_create: context %d's forest is a constant-turn one (MHT_FOREST_CT): not available to groups", i);
        MHT_REQUIRE(!ctxs[i]->forest->cluster_big && !ctxs[i]->forest->ais, "mht_group_create: context %d's forest is too large for the batched clustering kernel "
                    "(its tables live in HBM) or is an AIS forest: step it on its own", i);
        const Forest *a = ctxs[0]->forest, *b = ctxs[i]->forest;
        MHT_REQUIRE(ctxs[i]->device == ctxs[0]->device && ctxs[i]->stream == ctxs[0]->stream,
                    "mht_group_create: the members must share one device and one stream (context %d does not)", i);
        MHT_REQUIRE(a->Tcap == b->Tcap && a->Ncap == b->Ncap && a->Mpad == b->Mpad && a->R == b->R,
                    "mht_group_create: the members must have the same forest configuration (context %d differs)", i);
        for (int j = 0; j < i; ++j) MHT_REQUIRE(ctxs[j] != ctxs[i], "mht_group_create: context %d appears twice", i);
    }
    MHT_HIP_CHECK(hipSetDevice(ctxs[0]->device));
    mht_group* g = new (std::nothrow) mht_group();
    MHT_REQUIRE(g, "mht_group_create: out of host memory");
    g->n = n;
    g->period = 2 * ctxs[0]->forest->R;
    const int P = g->period;
    for (int i = 0; i < n; ++i) g->ctx[i] = ctxs[i];
    FGrowArgs* hga = new FGrowArgs[(size_t)n * P * 2];
    CommitArgs* hca = new CommitArgs[(size_t)n * P];
    ClusterArgs* hcl = new ClusterArgs[(size_t)n * 2];
    BlpArgs* hbl = new BlpArgs[(size_t)n * P * 2];
    BlpArgs* hbl0 = new BlpArgs[(size_t)n * P];
    { const char* e = getenv("MHT_BLP_TWO_TIER"); g->two_tier = e && e[0] == '1'; }
    // (measured, headline config, two groups: 16 sectors 56.7 k -> 72.2 k scans/s with the light pass, 4 sectors 40.6 k -> 36.1 k: the second
    // launch costs more than the narrow one saves while the full solver's workgroups still fit the machine in two rounds)
    { const char* e = getenv("MHT_BLP_LIGHT"); g->light = e ? (e[0] != '0') : (n >= 6); }
    // grow launch of the group: wavefront per target from eight sectors on (measured, headline config: 4 sectors 49 us workgroup-
    // per-target vs 54 us; 16 sectors 136 vs 121 us -- the wavefront variant costs 5.6 us per further sector, the other 7.2)
    { const char* e = getenv("MHT_FG_WAVE"); g->wave = e ? (e[0] != '0') : (n >= 8); }
    for (int i = 0; i < n; ++i) {
        const Forest* f = ctxs[i]->forest;
        for (int v = 0; v < P; ++v) {
            const int s = v == 0 ? P : v;      // any scan number with s % P == v (scans start at 1)
            fill_fgrow(f, s, false, hga[((size_t)i * P + v) * 2]);
            fill_fgrow(f, s, true, hga[((size_t)i * P + v) * 2 + 1]);
            fill_commit(f, s, hca[(size_t)i * P + v]);
            for (int tier = 1; tier <= 2; ++tier) {
                BlpArgs& b = hbl[((size_t)i * P + v) * 2 + tier - 1];
                fill_blp(f, s, b);
                b.skip_dead = 1;      // (the blocks are written once: similar-state pruning may be switched on for any later scan)
                g->blp_lds[tier - 1] = blp_set_tier(b, tier);
            }
            fill_blp(f, s, hbl0[(size_t)i * P + v]);
            hbl0[(size_t)i * P + v].skip_dead = 1;
            g->blp_lds[2] = blp_set_tier(hbl0[(size_t)i * P + v], 0);
        }
        fill_cluster(f, 2, hcl[(size_t)i * 2]);
        fill_cluster(f, 1, hcl[(size_t)i * 2 + 1]);
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&g->ga), sizeof(FGrowArgs) * n * P * 2);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->ca), sizeof(CommitArgs) * n * P);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->cl), sizeof(ClusterArgs) * n * 2);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->bl), sizeof(BlpArgs) * n * P * 2);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g->bl0), sizeof(BlpArgs) * n * P);
    if (e == hipSuccess) e = hipMemcpy(g->ga, hga, sizeof(FGrowArgs) * n * P * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->ca, hca, sizeof(CommitArgs) * n * P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->cl, hcl, sizeof(ClusterArgs) * n * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->bl, hbl, sizeof(BlpArgs) * n * P * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->bl0, hbl0, sizeof(BlpArgs) * n * P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);      // (the members' streams do not wait for the null stream)
    delete[] hga; delete[] hca; delete[] hcl; delete[] hbl; delete[] hbl0;
    if (e != hipSuccess) {
        set_error("mht_group_create: %s", hipGetErrorString(e));
        (void)mht_group_destroy(g);
        return MHT_E_HIP;
    }
    for (int i = 0; i < n; ++i) ctxs[i]->forest->in_groups += 1;
    g->counted = true;
    *out = g;
    return MHT_OK;
}

// a member switched the generation of its value table: its cached grow / commit blocks name the old one
static int group_refresh_member(mht_group* g, int i) {
    const int P = g->period;
    const Forest* f = g->ctx[i]->forest;
    FGrowArgs* hga = new FGrowArgs[(size_t)P * 2];
    CommitArgs* hca = new CommitArgs[(size_t)P];
    for (int v = 0; v < P; ++v) {
        const int s = v == 0 ? P : v;
        fill_fgrow(f, s, false, hga[(size_t)v * 2]);
        fill_fgrow(f, s, true, hga[(size_t)v * 2 + 1]);
        fill_commit(f, s, hca[v]);
    }
    hipError_t e = hipStreamSynchronize(g->ctx[0]->stream);      // (launches still reading the old blocks)
    if (e == hipSuccess) e = hipMemcpy(g->ga + (size_t)i * P * 2, hga, sizeof(FGrowArgs) * P * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->ca + (size_t)i * P, hca, sizeof(CommitArgs) * P, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    delete[] hga; delete[] hca;
    if (e != hipSuccess) { set_error("mht_group_step: %s", hipGetErrorString(e)); return MHT_E_HIP; }
    return MHT_OK;
}

extern "C" int mht_group_step(mht_group* g, const float* const* z, const int32_t* M) {
    MHT_REQUIRE(g && z && M, "mht_group_step: null argument");
    const int n = g->n, P = g->period;
    mht_ctx* c0 = g->ctx[0];
    MHT_HIP_CHECK(hipSetDevice(c0->device));
    for (int i = 0; i < n; ++i) {      // nothing may fail after the first member's scan counter has moved
        const Forest* f = g->ctx[i]->forest;
        MHT_REQUIRE(f, "mht_group_step: member %d lost its forest", i);
        MHT_REQUIRE(M[i] >= 0 && M[i] <= f->cfg.max_meas, "mht_group_step: member %d: M=%d exceeds max_meas=%d", i, M[i], f->cfg.max_meas);
        MHT_REQUIRE(z[i] || M[i] == 0, "mht_group_step: member %d: z is null", i);
        MHT_REQUIRE(!f->timing, "mht_group_step: per-stage timing is per forest (mht_forest_set_timing(ctx, 0) first)");
        if (f->dead) { set_error("mht_group_step: member %d is dead (a pool overflowed in an earlier scan)", i); return MHT_E_STATE; }
    }
    for (int i = 0; i < n; ++i) { const int rc = flush_publish(g->ctx[i], g->ctx[i]->forest); if (rc) return rc; }
    FBatch fb = {};
    PBatch cb = {}, bb = {}, bb2 = {}, bb0 = {};
    StepPlan pl[GROUP_MAX];
    int grid_g = 1, grid_b = 1;
    size_t lds = 256;
    for (int i = 0; i < n; ++i) {
        Forest* f = g->ctx[i]->forest;
        { const int rc = forest_begin_step(g->ctx[i], f, z[i], M[i], "mht_group_step", pl[i]); if (rc) return rc; }
        if (pl[i].rebuilt) { const int rc = group_refresh_member(g, i); if (rc) { f->dead = true; return rc; } }
        const int s = pl[i].s, v = s % P;
        FDyn& d = fb.d[i];
        d.z = z[i]; d.M = M[i]; d.W = pl[i].W;
        d.c_scan = f->pending_dyn.scan; d.c_M = f->pending_dyn.M; d.c_W = f->pending_dyn.W;
        d.maybe_dead = (f->similar_ran_scan == pl[i].s - 1);
        d.dbg = nullptr;
        fgrow_plan(d, pl[i].n_ub, f->Tcap, pl[i].fused, g->wave);
        fb.ga[i] = g->ga + ((size_t)i * P + v) * 2 + (pl[i].fused ? 1 : 0);
        fb.ca[i] = g->ca + (size_t)i * P + (s - 1 + P) % P;      // the commit of the scan before rides along (if fused)
        cb.p[i] = g->cl + (size_t)i * 2 + (s & 1);
        bb.p[i] = g->bl + ((size_t)i * P + v) * 2;
        bb2.p[i] = g->bl + ((size_t)i * P + v) * 2 + 1;
        bb0.p[i] = g->bl0 + (size_t)i * P + v;
        const int gg = fgrow_grid_of(d);
        if (gg > grid_g) grid_g = gg;
        // (a quarter of the targets, not half as in the one-sector launch: the 155 KB workgroups of S sectors share the CUs one at a time, and one
        // without a multi-target cluster still costs its entry; measured, headline config, two groups: S = 4 39.1 -> 42.3 k scans/s, S = 16 74.1 -> 75.2 k;
        // an eighth: 39.7 / 74.8 k.  MHT_GROUP_BLP_DIV overrides)
        static int gdiv = -1; if (gdiv < 0) { const char* e = getenv("MHT_GROUP_BLP_DIV"); gdiv = e ? atoi(e) : 4; if (gdiv < 1) gdiv = 4; }
        int gbl = f->nT_ub_step / gdiv + 8;
        if (gbl > 1024) gbl = 1024;
        if (gbl > grid_b) grid_b = gbl;
        const size_t l = g->wave ? fgrow_wave_lds_bytes(d.W, f->pds, f->AW) : fgrow_lds_bytes(d.W, f->pds, f->AW);
        if (l > lds) lds = l;
        f->commit_pending = false;
    }
    const Forest* f0 = c0->forest;
    int rc = launch_fgrow_batch(c0, fb, n, grid_g, lds, g->ctx[0]->forest->pds, g->wave);
    if (!rc) rc = launch_cluster_batch(c0, cb, n, f0->Tcap, f0->n_mnodes);
    // similar-state pruning of the members that ask for it: a launch of their own each, between clustering and the ILPs
    for (int i = 0; i < n && !rc; ++i) {
        const Forest* f = g->ctx[i]->forest;
        if (f->prune_thr > 0.f) {
            SimilarArgs sa;
            fill_similar(f, pl[i].s, sa);
            rc = launch_prune_similar(c0, sa, f->nT_ub_step);
            g->ctx[i]->forest->similar_ran_scan = pl[i].s;
        }
    }
    // ILPs in two LDS tiers: the small footprint (several workgroups per CU) takes the clusters that fit it and the single-target
    // clusters, a narrow launch with the default footprint takes the few that do not
    if (g->light && !g->two_tier) {
        if (!rc) rc = launch_blp_light_batch(c0, bb, n, grid_b);
        if (!rc) rc = launch_blp_batch(c0, bb2, n, 8 + TEAM_W, g->blp_lds[1]);      // (the workgroups without a cluster join the teams of the giant ones)
    } else if (g->two_tier) {
        if (!rc) rc = launch_blp_batch(c0, bb, n, grid_b, g->blp_lds[0]);
        if (!rc) rc = launch_blp_batch(c0, bb2, n, 24, g->blp_lds[1]);
    } else {
        if (!rc) rc = launch_blp_batch(c0, bb0, n, grid_b, g->blp_lds[2]);
    }
    for (int i = 0; i < n; ++i) {
        Forest* f = g->ctx[i]->forest;
        if (rc) f->dead = true;
        else forest_end_step(f, pl[i], M[i]);
    }
    return rc;
}

static int forest_initiate_impl(mht_ctx* ctx, mht_initiator* in, const float* z, int32_t M, double now, bool defer_publish) {
    MHT_REQUIRE(ctx && ctx->forest && in, "mht_forest_initiate: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->scan > 0, "mht_forest_initiate: no scan processed yet");
    MHT_REQUIRE(M == f->last_M, "mht_forest_initiate: M=%d is not the scan just stepped (M=%d)", M, f->last_M);
    if (!z) z = f->z_cur;      // (the scan mht_forest_step_host staged)
    const double* bx; const float* bP; const uint8_t* bfl; const double* bpd; const int32_t* bme; const int32_t* bn; int cap; mht_ctx* ictx;
    initiator_born_ptrs(in, &bx, &bP, &bfl, &bpd, &bme, &bn, &cap, &ictx);
    MHT_REQUIRE(ictx == ctx, "mht_forest_initiate: the initiator belongs to another context");
    MHT_REQUIRE(cap <= BIRTH_CAP, "mht_forest_initiate: the initiator's max_born=%d exceeds the report's %d", cap, BIRTH_CAP);
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    InitArgs ia = {};
    char* report_dev = f->report_dev2[f->scan & 1];
    const bool init_done = f->init_ran_scan == f->scan;      // (mht_forest_scan: it ran inside this scan's cluster launch)
    AisUsedArgs au = {};
    if (!init_done && f->ais && initiator_ais_pending(in) > 0) {      // messages for the initiator: which of them a track took is decided here, behind the commit
        const int nb_ = (f->scan + 1) & 1;
        au.nA = initiator_ais_pending(in);
        initiator_ais_ptrs(in, &au.msgs, &au.used);
        au.mmsi = f->l_mmsi[f->scan % f->R]; au.first = f->tab[nb_].first; au.leaf_off = f->tab[nb_].leaf_off; au.cnt = f->cnt;
    }
    if (!init_done) { initiator_scan_args(in, z, M, reinterpret_cast<const unsigned long long*>(report_dev + f->used_off), now, ia); ia.bhint = f->bhint_dev; ia.forest_overflow = &f->cnt->overflow; ia.scan_no = f->scan; }
    AddArgs a = {};
    a.n = cap; a.n_dev = bn; a.x0 = bx; a.pd = bpd; a.P0 = bP; a.meas = bme; a.flags = bfl; a.ids = nullptr; a.accepted = nullptr;
    a.check = 1; a.thr = f->cfg.merge_threshold;
    const int nb = (f->scan + 1) & 1;
    a.layer = f->layer[f->scan % f->R];
    a.tab = f->tab[nb]; a.vidx = nb;
    a.path = f->path[f->scan & 1]; a.apath = f->apath[f->scan & 1]; a.PD = f->pds;
    a.cnt = f->cnt; a.scan = f->scan; a.Nwin = f->cfg.n_scan; a.Tcap = f->Tcap;
    a.near = f->near;
    fill_model_only(a.model, &f->model); a.vt = f->vt; a.root_base = f->root_base; a.ct_Proot = f->ct ? f->ct_Proot[f->scan % f->R] : nullptr;
    if (f->ais) { a.mmsi = f->l_mmsi[f->scan % f->R]; a.hmmsi = f->l_hmmsi[f->scan % f->R]; }
    a.hdr = reinterpret_cast<ReportHeader*>(report_dev);
    a.births = reinterpret_cast<mht_birth_report*>(report_dev + f->birth_off);
    // commit (if it is still pending: the used-measurement mask of the scan is part of it) + initiator + admission: one launch
    { const int rc = flush_publish(ctx, f); if (rc) return rc; }      // (an older report still waiting for a ride: its device block is about to be reused)
    // the report goes to the host from this launch, or -- streaming: mht_forest_scan -- with the next scan's grow launch
    PublishArgs pub = publish_args(f);
    if (defer_publish) { f->pub_args = pub; pub.dst = nullptr; }
    // streaming (mht_forest_scan) and nothing left to do here but the commit and the admission: both ride in the next scan's grow launch
    const bool ride = defer_publish && init_done && f->adm_fuse && f->commit_pending && !f->ais && au.nA == 0 && ia.nA == 0;
    if (!ride) { const int rc = wait_init_ev(ctx, f); if (rc) return rc; }
    if (!ride) { const int rc = flush_z_wait(ctx, f); if (rc) return rc; }      // (post_scan_kernel may read the scan)
    if (ride) { f->adm = a; f->adm_pending = true; }
    else if (au.nA > 0 || ia.nA > 0)
        hipLaunchKernelGGL(post_scan_kernel<true>, dim3(1), dim3(1024), 0, ctx->stream, f->pending, f->pending_dyn, ia, a, f->commit_pending ? 1 : 0, pub,
                           init_done ? 0 : 1, au);
    else
        hipLaunchKernelGGL(post_scan_kernel<false>, dim3(1), dim3(1024), 0, ctx->stream, f->pending, f->pending_dyn, ia, a, f->commit_pending ? 1 : 0, pub,
                           init_done ? 0 : 1, au);
    f->published_scan = f->scan;
    MHT_HIP_CHECK(hipGetLastError());
    if (!ride) f->commit_pending = false;
    // the host does not know how many of the candidates exist: every bound moves by the most there can be
    f->nT_ub = (f->nT_ub + cap < f->Tcap) ? f->nT_ub + cap : f->Tcap;
    f->L_ub = (f->L_ub + cap < f->Ncap) ? f->L_ub + cap : f->Ncap;
    f->births_since_step += cap;
    f->births_init_ub[f->scan % 64] = cap;
    f->init_mreq = f->ais ? 0 : initiator_mreq(in);
    f->report_pending = true;      // (the births block of the report changed)
    return MHT_OK;
}

extern "C" int mht_forest_initiate(mht_ctx* ctx, mht_initiator* in, const float* z, int32_t M, double now) {
    MHT_REQUIRE(NX == 4, "mht_forest_initiate: the M-of-N initiator is the reference's 4-state one (m_of_n.py imports models/pv); this is the %d-state build", NX);
    return forest_initiate_impl(ctx, in, z, M, now, false);
}

static int step_host_impl(mht_ctx* ctx, const float* z_host, int32_t M, bool mark_done, mht_initiator* init = nullptr, double now = 0.0) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_step_host: no forest");
    Forest* f = ctx->forest;
    MHT_REQUIRE(M >= 0 && M <= f->cfg.max_meas, "mht_forest_step_host: M=%d exceeds max_meas=%d", M, f->cfg.max_meas);
    if (M > 0) {
        MHT_REQUIRE(z_host, "mht_forest_step_host: z is null");
        MHT_HIP_CHECK(hipSetDevice(ctx->device));
        // a ring of pinned staging buffers: the host only waits if the copy that used this slot Z_RING scans ago is still in flight
        const int slot = f->z_slot;
        f->z_slot = (slot + 1) % Z_RING;
        if (f->z_used[slot]) MHT_HIP_CHECK(hipEventSynchronize(f->z_ev[slot]));      // (recorded for the scans whose staging the ctx stream waited for; the streamed ones are covered by the guard below)
        float* zh = f->z_host + (size_t)slot * 2 * f->Mpad;
        float* zd = f->z_dev + (size_t)slot * 2 * f->Mpad;
        memcpy(zh, z_host, (size_t)M * 2 * sizeof(float));
        const int n16 = (M * 2 + 3) / 4;
        // The scan is pulled out of pinned memory by a one-workgroup kernel on a stream of its own: a host that streams scans in queues
        // scan k + 1 while scan k is still on the device, so the pull overlaps scan k's ILPs instead of standing in front of scan k + 1's
        // grow launch (4 us + a launch gap per scan); the ctx stream waits for the event (already signalled by then).
        if (!f->stage_stream_tried) {
            f->stage_stream_tried = true;
            const char* e = getenv("MHT_STAGE_STREAM");
            if (!(e && e[0] == '0') && hipStreamCreateWithFlags(&f->stage_stream, hipStreamNonBlocking) != hipSuccess) f->stage_stream = nullptr;
        }
        hipStream_t sst = f->stage_stream ? f->stage_stream : ctx->stream;
        // The slot's previous tenant (Z_RING scans ago) must have been CONSUMED -- grow, initiator, post-scan launches on the ctx stream --
        // before the side stream overwrites it: z_ev only says that its pull had finished.  A host that loops over this call without ever
        // reading a report can get that far ahead of the device.  Every Z_GUARD scans an event goes onto the ctx stream (behind everything
        // queued for the scans so far) and the side stream waits for the one recorded Z_GUARD scans earlier: that covers the tenants of the
        // next Z_GUARD slots (Z_RING = 2 x Z_GUARD), at one event operation per two scans.
        const bool by_flag = f->stage_stream && init && forest_streams_uf(f, init) && !(f->ais && f->ais_armed) && !f->serial_prof;
        // (streamed scans: the HOST waits for that event -- long signalled -- instead of the side stream: it then also knows that the pinned slots
        // were pulled, and no per-scan event is needed; and the new event is recorded BEHIND this scan's grow launch (forest_step_impl,
        // z_guard_due), not in front of it: a marker packet in front of an any-order grow launch would end its overlap with the previous
        // scan's ILP launch on every fourth scan)
        if (f->stage_stream && f->z_count % Z_GUARD == 0) {
            const int gi = (int)((f->z_count / Z_GUARD) & 1);
            if (f->z_guard_ev[1 - gi]) {
                if (by_flag) MHT_HIP_CHECK(hipEventSynchronize(f->z_guard_ev[1 - gi]));
                else MHT_HIP_CHECK(hipStreamWaitEvent(sst, f->z_guard_ev[1 - gi], 0));      // (recorded Z_GUARD scans ago: covers every scan up to then)
            }
            if (!f->z_guard_ev[gi]) MHT_HIP_CHECK(hipEventCreateWithFlags(&f->z_guard_ev[gi], hipEventDisableTiming));
            if (by_flag) f->z_guard_due = gi;
            else MHT_HIP_CHECK(hipEventRecord(f->z_guard_ev[gi], ctx->stream));
        }
        f->z_count += 1;
        // Streamed scans (device initiator, clusters from the union-find): the ctx stream does not wait for the staging kernel through an
        // event -- an event wait is a barrier packet in the queue: it costs microseconds and ends the overlap of this scan's grow launch
        // with the previous scan's ILP launch.  The kernel posts a tag behind its (written-through) stores instead, and the scan's first
        // readers -- the grow launch's target workgroups, the initiator's launch -- wait for the tag (it is there long before: the host
        // runs ahead).  Any other reader of this scan on the ctx stream gets the event wait first (z_wait_slot, flush_z_wait).
        f->z_tag_step = by_flag ? (unsigned long long)f->z_count : 0ull;
        hipLaunchKernelGGL(stage_scan_kernel, dim3(1), dim3(256), 0, sst, reinterpret_cast<const float4*>(f->z_host_dev + (size_t)slot * 2 * f->Mpad),
                           reinterpret_cast<float4*>(zd), n16, by_flag ? &f->cnt->z_flag : nullptr, f->z_tag_step);
        MHT_HIP_CHECK(hipGetLastError());
        if (!by_flag) MHT_HIP_CHECK(hipEventRecord(f->z_ev[slot], sst));      // (the host may refill this slot once the kernel has run; streamed scans: the guard)
        if (f->stage_stream && !by_flag) { MHT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, f->z_ev[slot], 0)); if (!f->init_ev_lazy) f->init_ev_pending = false; }      // (behind the side stream's initiator launch as well)
        f->z_wait_slot = by_flag ? slot : -1;
        f->z_used[slot] = !by_flag;
        f->z_cur = zd;
        (void)mark_done;
        hp_mark(1);
        const int rc = forest_step_impl(ctx, zd, M, init, now);
        f->z_tag_step = 0;
        return rc;
    }
    f->z_cur = f->z_dev;
    f->z_tag_step = 0; f->z_wait_slot = -1;
    return forest_step_impl(ctx, f->z_dev, M, init, now);
}
extern "C" int mht_forest_step_host(mht_ctx* ctx, const float* z_host, int32_t M) { return step_host_impl(ctx, z_host, M, true); }

// One radar scan of the drop-in API path in one call: steps 1-6 (mht_forest_step_host), step 7 (mht_forest_initiate, if an
// initiator is given) and the start of the report's way to the host (mht_forest_report_begin).  Nothing here waits for the device.
extern "C" int mht_forest_scan(mht_ctx* ctx, mht_initiator* in, const float* z_host, int32_t M, double now) {
    hp_begin();
    if (in) {      // (checked before the scan is stepped: nothing may fail between the initiator's run and the admission of its births)
        MHT_REQUIRE(NX == 4, "mht_forest_scan: the M-of-N initiator is the reference's 4-state one (m_of_n.py imports models/pv); this is the %d-state build", NX);
        MHT_REQUIRE(ctx && ctx->forest, "mht_forest_scan: no forest");
        const double* bx; const float* bP; const uint8_t* bfl; const double* bpd; const int32_t* bme; const int32_t* bn; int cap; mht_ctx* ictx;
        initiator_born_ptrs(in, &bx, &bP, &bfl, &bpd, &bme, &bn, &cap, &ictx);
        MHT_REQUIRE(ictx == ctx, "mht_forest_scan: the initiator belongs to another context");
        MHT_REQUIRE(cap <= BIRTH_CAP, "mht_forest_scan: the initiator's max_born=%d exceeds the report's %d", cap, BIRTH_CAP);
    }
    // (messages waiting for the initiator: which of them a track took is known behind the scan's pruning only -- the initiator then runs in
    // post_scan_kernel, not next to the clustering)
    const bool ais_init = in && initiator_ais_pending(in) > 0;
    hp_mark(0);
    int rc = step_host_impl(ctx, z_host, M, false, ais_init ? nullptr : in, now);
    if (rc) return rc;
    if (!in) return mht_forest_report_begin(ctx);
    rc = forest_initiate_impl(ctx, in, nullptr, M, now, true);
    hp_mark(10);
    if (rc) return rc;
    // the report is complete in its device block; its push to the host rides in the next scan's grow launch (or in a launch of its
    // own as soon as somebody asks for it: report_expose)
    Forest* f = ctx->forest;
    f->rep_slot = f->scan & 1;
    f->pub_slot = f->rep_slot;
    f->pub_deferred = true;
    f->rep_started[f->rep_slot] = false;
    f->report_pending = false;
    f->rep_inflight = true;
    return MHT_OK;
}

// Starts the transfer of the last scan's report (commit first, if it is still pending) into one of two pinned host buffers and
// returns; mht_forest_report waits for it.  A host that steps scan k+1 before it reads the report of scan k overlaps its own work
// with the device's.
extern "C" int mht_forest_report_begin(mht_ctx* ctx) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_report_begin: no forest");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->scan > 0, "mht_forest_report_begin: no scan processed yet");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    if (!f->report_pending) return flush_publish(ctx, f);      // (streaming: the last scan's report waits for a ride in the next grow launch -- it goes now, in a launch of its own)
    { const int rc = flush_publish(ctx, f); if (rc) return rc; }
    { const int rc = flush_commit(ctx, f, true); if (rc) return rc; }
    f->rep_slot = f->scan & 1;      // host block = device block = scan parity
    if (f->published_scan != f->scan) {      // the commit ran without a host block (inside a grow launch): fetch the device block
        const size_t bytes = f->rec_off + (size_t)f->nT_ub_step * sizeof(mht_target_report);
        MHT_HIP_CHECK(hipMemcpyAsync(f->report_host2[f->rep_slot], f->report_dev2[f->scan & 1], bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    MHT_HIP_CHECK(hipEventRecord(f->rep_ev[f->rep_slot], ctx->stream));
    f->host_block_scan[f->rep_slot] = f->scan;
    f->report_pending = false;
    f->rep_inflight = true;
    f->rep_started[f->rep_slot] = true;
    return MHT_OK;
}

static int report_expose(mht_ctx* ctx, Forest* f, int slot, mht_scan_report* out, bool lagged = false);

extern "C" int mht_forest_report(mht_ctx* ctx, mht_scan_report* out) {
    MHT_REQUIRE(ctx && ctx->forest && out, "mht_forest_report: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->scan > 0, "mht_forest_report: no scan processed yet");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    if (f->report_pending) { const int rc = mht_forest_report_begin(ctx); if (rc) return rc; }
    return report_expose(ctx, f, f->rep_slot, out);
}

// The report whose transfer the last (which = 0) or the last but one (which = 1) mht_forest_report_begin started: a host that
// begins the report of scan k+1 before it reads the one of scan k keeps two scans in flight.
extern "C" int mht_forest_report_get(mht_ctx* ctx, int32_t which, mht_scan_report* out) {
    MHT_REQUIRE(ctx && ctx->forest && out && which >= 0 && which <= 2, "mht_forest_report_get: bad argument");
    Forest* f = ctx->forest;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    if (which == 2) {
        // streaming (mht_forest_scan): the report of the last scan has not started its way to the host (it rides in the NEXT grow launch),
        // and the host block it will go to still holds the report of the scan two before it
        if (!(f->pub_deferred && f->pub_slot == f->rep_slot && f->host_block_scan[f->rep_slot] == f->scan - 2 && f->scan >= 3)) {
            set_error("mht_forest_report_get: the report of scan %d is not in a host block any more (which = 2 reads it between two mht_forest_scan calls)", f->scan - 2);
            return MHT_E_STATE;
        }
        return report_expose(ctx, f, f->rep_slot, out, true);
    }
    return report_expose(ctx, f, f->rep_slot ^ which, out);
}

// the host block `slot` holds its report: the words of the workgroups that pushed it (PublishArgs::done), or the event behind the launch
static int report_wait(Forest* f, int slot) {
    if (!f->rep_by_flag[slot]) { MHT_HIP_CHECK(hipEventSynchronize(f->rep_ev[slot])); return MHT_OK; }
    const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(f->report_host2[slot] + f->done_off);
    const unsigned long long tag = f->rep_tag[slot];
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spins = 0;; ++spins) {
        bool all = true;
        for (int q = 0; q < PUB_DONE_WORDS; ++q) all = all && (w[q] == tag);
        if (all) break;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0xfff) == 0xfff) {
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > 10.0) { set_error("mht_forest_report_get: the report of the host block did not arrive within 10 s"); f->dead = true; return MHT_E_HIP; }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return MHT_OK;
}
static int report_expose(mht_ctx* ctx, Forest* f, int slot, mht_scan_report* out, bool lagged) {
    if (lagged) {
        { const int rc = report_wait(f, slot); if (rc) return rc; }      // (pushed by the grow launch of the scan before the last)
    } else {
        if (f->pub_deferred && slot == f->pub_slot) { const int rc = flush_publish(ctx, f); if (rc) return rc; }
        if (f->rep_started[slot]) {
            { const int rc = report_wait(f, slot); if (rc) return rc; }
            f->rep_started[slot] = false;
        }
        if (slot == f->rep_slot) f->rep_inflight = false;
    }
    f->report_host = f->report_host2[slot];
    const ReportHeader* h = reinterpret_cast<const ReportHeader*>(f->report_host);
    if (lagged && h->scan != f->scan - 2) { set_error("mht_forest_report_get: the host block holds scan %d, not %d", h->scan, f->scan - 2); return MHT_E_STATE; }
    memcpy(out, h, sizeof(ReportHeader));
    out->used = reinterpret_cast<const uint64_t*>(f->report_host + f->used_off);
    out->targets = reinterpret_cast<const mht_target_report*>(f->report_host + f->rec_off);
    out->births = reinterpret_cast<const mht_birth_report*>(f->report_host + f->birth_off);
    // tighten the host-side bounds; targets added since that scan was issued are not in its report (the report may be read after
    // later scans have been issued: only births move the target count up)
    if (f->scan - h->scan < 60) {
        if (f->births_init_ub[h->scan % 64] > h->n_births) f->births_init_ub[h->scan % 64] = h->n_births;      // (now known)
        const long long ub = (long long)h->n_alive + f->births_between(h->scan, f->scan + 1);
        f->nT_ub = ub < f->Tcap ? (int)ub : f->Tcap;
        f->L_ub = f->Ncap;
    }
    if (h->error == MHT_E_HIP) {
        f->dead = true;
        set_error("forest: grow_kernel stalled in scan %d waiting for a tile that was never dispatched (GPU shared with another "
                  "resident grid?); the forest must be recreated", h->scan);
        return MHT_E_HIP;
    }
    if (h->error) {
        f->dead = true;
        set_error("forest: a pool overflowed during scan %d (max_nodes=%d, max_targets=%d, or the initiator's max_born / max_prelim behind it): children=%d", h->scan,
                  f->Ncap, f->Tcap, h->n_children);
        return MHT_E_CAPACITY;
    }
    if (h->n_limit) {
        set_error("forest: %d ILP(s) hit the branch-and-bound node limit in scan %d", h->n_limit, h->scan);
        return MHT_E_LIMIT;
    }
    return MHT_OK;
}

extern "C" int mht_forest_set_blp_time_limit(mht_ctx* ctx, double milliseconds) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_set_blp_time_limit: no forest");
    MHT_REQUIRE(!(milliseconds != milliseconds), "mht_forest_set_blp_time_limit: NaN");
    if (ctx->forest->in_groups > 0) {      // (mht_group_step reads the limit from argument blocks written at mht_group_create: it would be ignored silently)
        set_error("mht_forest_set_blp_time_limit: the forest is a member of a group; set the limit before mht_group_create");
        return MHT_E_STATE;
    }
    ctx->forest->blp_time_limit = milliseconds > 0.0 ? (long long)(milliseconds * 1e5) : 0;
    return MHT_OK;
}

extern "C" int mht_forest_set_prune_similar(mht_ctx* ctx, double threshold) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_set_prune_similar: no forest");
    MHT_REQUIRE(!(threshold != threshold), "mht_forest_set_prune_similar: threshold is NaN");
    Forest* f = ctx->forest;
    MHT_REQUIRE(!f->shard_open, "mht_forest_set_prune_similar: a sharded step is open");
    f->prune_thr = threshold > 0.0 ? (float)threshold : 0.f;
    return MHT_OK;
}

// PB = bytes per covariance entry of the export: 4 (mht_forest_leaves) or 8 (mht_forest_leaves_f64: exact for both kinds of value)
static int forest_leaves_impl(mht_ctx* ctx, int32_t capacity, double* x, void* P, int PB, double* cnllr, int32_t* meas,
                              int32_t* target, int32_t* id, int32_t* node, uint8_t* flags, int32_t* n_out) {
    MHT_REQUIRE(ctx && ctx->forest && n_out, "mht_forest_leaves: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(capacity >= 0, "mht_forest_leaves: negative capacity");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    { const int rc = flush_commit(ctx, f); if (rc) return rc; }
    FCounts c;
    MHT_HIP_CHECK(hipMemcpyAsync(&c, f->cnt, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *n_out = c.L;
    const int n = c.L < capacity ? c.L : capacity;
    if (n == 0) return MHT_OK;
    const size_t o_x = 0, o_c = o_x + (size_t)n * (NX * 8), o_P = o_c + (size_t)n * 8, o_m = o_P + (size_t)n * ((size_t)NP * PB),
                 o_t = o_m + (size_t)n * 4, o_i = o_t + (size_t)n * 4, o_n = o_i + (size_t)n * 4, o_f = o_n + (size_t)n * 4,
                 total = o_f + n + 16;
    int rc = stage_host_ensure(f, total);
    if (rc) return rc;
    rc = f->stage_dev.ensure(total);
    if (rc) return rc;
    char* d = static_cast<char*>(f->stage_dev.ptr);
    char* h = static_cast<char*>(f->stage_host);
    const int nb = (f->scan + 1) & 1;
    LeavesArgs a = {f->layer[f->scan % f->R], f->tab[nb], f->cnt, f->vt, n,
                    (double*)(d + o_x), PB == 4 ? (float*)(d + o_P) : nullptr, (double*)(d + o_c), (int32_t*)(d + o_m), (int32_t*)(d + o_t),
                    (int32_t*)(d + o_i), (int32_t*)(d + o_n), (uint8_t*)(d + o_f), PB == 8 ? (double*)(d + o_P) : nullptr};
    LeavesCt lc = {};
    if (f->ct) {
        lc.c.on = 1; lc.li = f->scan % f->R; lc.lprev = (f->scan + f->R - 1) % f->R;
        for (int k = 0; k < f->R; ++k) { lc.c.Pbar[k] = f->ct_Pbar[k]; lc.c.Phat[k] = f->ct_Phat[k]; lc.c.Proot[k] = f->ct_Proot[k]; }
    }
    hipLaunchKernelGGL(leaves_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a, lc);
    MHT_HIP_CHECK(hipGetLastError());
    MHT_HIP_CHECK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // leaves that similar-state pruning took out of the tree keep their slot in the layer but are no hypotheses: squeezed out here
    const uint8_t* hf = reinterpret_cast<const uint8_t*>(h + o_f);
    int live = 0;
    for (int i = 0; i < n; ++i) {
        if (hf[i] & F_DEAD) continue;
        if (live != i) {
            memmove(h + o_x + (size_t)live * (NX * 8), h + o_x + (size_t)i * (NX * 8), NX * 8);
            memmove(h + o_c + (size_t)live * 8, h + o_c + (size_t)i * 8, 8);
            memmove(h + o_P + (size_t)live * ((size_t)NP * PB), h + o_P + (size_t)i * ((size_t)NP * PB), (size_t)NP * PB);
            memmove(h + o_m + (size_t)live * 4, h + o_m + (size_t)i * 4, 4);
            memmove(h + o_t + (size_t)live * 4, h + o_t + (size_t)i * 4, 4);
            memmove(h + o_i + (size_t)live * 4, h + o_i + (size_t)i * 4, 4);
            memmove(h + o_n + (size_t)live * 4, h + o_n + (size_t)i * 4, 4);
            h[o_f + live] = h[o_f + i];
        }
        ++live;
    }
    *n_out = c.L - (n - live);
    if (x) memcpy(x, h + o_x, (size_t)live * (NX * 8));
    if (cnllr) memcpy(cnllr, h + o_c, (size_t)live * 8);
    if (P) memcpy(P, h + o_P, (size_t)live * ((size_t)NP * PB));
    if (meas) memcpy(meas, h + o_m, (size_t)live * 4);
    if (target) memcpy(target, h + o_t, (size_t)live * 4);
    if (id) memcpy(id, h + o_i, (size_t)live * 4);
    if (node) memcpy(node, h + o_n, (size_t)live * 4);
    if (flags) memcpy(flags, h + o_f, live);
    return MHT_OK;
}

extern "C" int mht_forest_leaves(mht_ctx* ctx, int32_t capacity, double* x, float* P, double* cnllr, int32_t* meas,
                                 int32_t* target, int32_t* id, int32_t* node, uint8_t* flags, int32_t* n_out) {
    return forest_leaves_impl(ctx, capacity, x, P, 4, cnllr, meas, target, id, node, flags, n_out);
}
extern "C" int mht_forest_leaves_f64(mht_ctx* ctx, int32_t capacity, double* x, double* P, double* cnllr, int32_t* meas,
                                     int32_t* target, int32_t* id, int32_t* node, uint8_t* flags, int32_t* n_out) {
    return forest_leaves_impl(ctx, capacity, x, P, 8, cnllr, meas, target, id, node, flags, n_out);
}

static void chain_args_of(Forest* f, ChainArgs& a, CtLayers& cl) {
    for (int k = 0; k < f->R; ++k) { a.layers[k] = f->layer[k]; a.lgen[k] = f->layer_gen[k]; }
    a.vt[0] = f->vts[0]; a.vt[1] = f->vts[1];
    a.R = f->R;
    if (f->ct) {
        cl.on = 1;
        for (int k = 0; k < f->R; ++k) { cl.Pbar[k] = f->ct_Pbar[k]; cl.Phat[k] = f->ct_Phat[k]; cl.Proot[k] = f->ct_Proot[k]; }
    }
}

static int forest_chain_impl(mht_ctx* ctx, int32_t scan, int32_t node, int32_t max_len, int32_t* nodes, int32_t* meas,
                             double* x, double* cnllr, void* P, int PB, uint8_t* flags, int32_t* n_out) {
    MHT_REQUIRE(ctx && ctx->forest && n_out, "mht_forest_chain: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(scan >= 0 && scan <= f->scan && f->scan - scan < f->R, "mht_forest_chain: scan %d is outside the window", scan);
    MHT_REQUIRE(node >= 0 && node < f->Ncap && max_len >= 1, "mht_forest_chain: bad node / max_len");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    { const int rc = flush_commit(ctx, f); if (rc) return rc; }
    int len = max_len;
    const int avail = f->R - (f->scan - scan);   // layers still in the ring going backwards
    if (len > avail) len = avail;
    const size_t o_n = 0, o_m = o_n + (size_t)len * 4, o_x = (o_m + (size_t)len * 4 + 7) & ~(size_t)7, o_c = o_x + (size_t)len * (NX * 8),
                 o_P = o_c + (size_t)len * 8, o_fl = o_P + (size_t)len * ((size_t)NP * PB), o_k = (o_fl + (size_t)len + 7) & ~(size_t)7, total = o_k + 16;
    int rc = stage_host_ensure(f, total);
    if (rc) return rc;
    rc = f->stage_dev.ensure(total);
    if (rc) return rc;
    char* d = static_cast<char*>(f->stage_dev.ptr);
    char* h = static_cast<char*>(f->stage_host);
    ChainArgs a = {};
    CtLayers cl = {};
    chain_args_of(f, a, cl);
    a.scan = scan; a.node = node; a.max_len = len;
    a.nodes = (int32_t*)(d + o_n); a.meas = (int32_t*)(d + o_m); a.x = (double*)(d + o_x); a.cnllr = (double*)(d + o_c);
    a.P = PB == 4 ? (float*)(d + o_P) : nullptr; a.P64 = PB == 8 ? (double*)(d + o_P) : nullptr; a.flags = (uint8_t*)(d + o_fl); a.n_out = (int32_t*)(d + o_k);
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, ctx->stream, a, cl);
    MHT_HIP_CHECK(hipGetLastError());
    MHT_HIP_CHECK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int n = *(int32_t*)(h + o_k);
    *n_out = n;
    if (nodes) memcpy(nodes, h + o_n, (size_t)n * 4);
    if (meas) memcpy(meas, h + o_m, (size_t)n * 4);
    if (x) memcpy(x, h + o_x, (size_t)n * (NX * 8));
    if (cnllr) memcpy(cnllr, h + o_c, (size_t)n * 8);
    if (P) memcpy(P, h + o_P, (size_t)n * ((size_t)NP * PB));
    if (flags) memcpy(flags, h + o_fl, (size_t)n);
    return MHT_OK;
}
extern "C" int mht_forest_chain(mht_ctx* ctx, int32_t scan, int32_t node, int32_t max_len, int32_t* nodes, int32_t* meas,
                                double* x, double* cnllr, float* P, int32_t* n_out) {
    return forest_chain_impl(ctx, scan, node, max_len, nodes, meas, x, cnllr, P, 4, nullptr, n_out);
}
extern "C" int mht_forest_chain_f64(mht_ctx* ctx, int32_t scan, int32_t node, int32_t max_len, int32_t* nodes, int32_t* meas,
                                    double* x, double* cnllr, double* P, uint8_t* flags, int32_t* n_out) {
    return forest_chain_impl(ctx, scan, node, max_len, nodes, meas, x, cnllr, P, 8, flags, n_out);
}

extern "C" int mht_forest_chains_begin(mht_ctx* ctx, int32_t scan, const int32_t* start_nodes, int32_t count, int32_t max_len, int32_t f64, int64_t* ticket) {
    MHT_REQUIRE(ctx && ctx->forest && start_nodes && ticket, "mht_forest_chains_begin: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(scan >= 0 && scan <= f->scan && f->scan - scan < f->R, "mht_forest_chains_begin: scan %d is outside the window", scan);
    MHT_REQUIRE(count >= 1 && count <= f->Tcap && max_len >= 1, "mht_forest_chains_begin: bad count / max_len");
    for (int i = 0; i < count; ++i) MHT_REQUIRE(start_nodes[i] >= 0 && start_nodes[i] < f->Ncap, "mht_forest_chains_begin: bad node %d", start_nodes[i]);
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    // (a layer older than the newest scan's is final: its commit is queued at the latest with the newest scan's grow launch, and this launch runs behind it)
    if (scan == f->scan) { const int rc = flush_commit(ctx, f); if (rc) return rc; }
    int len = max_len;
    const int avail = f->R - (f->scan - scan);   // layers still in the ring going backwards
    if (len > avail) len = avail;
    const int PB = f64 ? 8 : 4;
    const ChainRec r = chain_rec(len, PB);
    const size_t head = ((size_t)count * 4 + 15) & ~(size_t)15, total = head + (size_t)count * r.stride;
    Forest::ChainSlot& cs = f->chain_slots[f->chain_next % Forest::CHAIN_SLOTS];
    if (!cs.ev) MHT_HIP_CHECK(hipEventCreateWithFlags(&cs.ev, hipEventDisableTiming));
    if (cs.pending) { MHT_HIP_CHECK(hipEventSynchronize(cs.ev)); cs.pending = false; }      // (the ticket that owned this block expires)
    if (cs.bytes < total) {
        if (cs.host) MHT_HIP_CHECK(hipHostFree(cs.host));
        cs.host = nullptr; cs.bytes = 0;
        MHT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&cs.host), total + 4096, hipHostMallocMapped));
        MHT_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&cs.dev), cs.host, 0));
        cs.bytes = total + 4096;
    }
    memcpy(cs.host, start_nodes, (size_t)count * 4);
    ChainArgs a = {};
    CtLayers cl = {};
    chain_args_of(f, a, cl);
    a.scan = scan; a.max_len = len;
    hipLaunchKernelGGL(chains_kernel, dim3((count + 63) / 64), dim3(64), 0, ctx->stream, a, cl, reinterpret_cast<const int32_t*>(cs.dev), count, cs.dev + head, PB);
    MHT_HIP_CHECK(hipGetLastError());
    MHT_HIP_CHECK(hipEventRecord(cs.ev, ctx->stream));
    cs.pending = true; cs.count = count; cs.len = len; cs.pb = PB; cs.ticket = f->chain_next;
    *ticket = f->chain_next++;
    return MHT_OK;
}

extern "C" int mht_forest_chains_fetch(mht_ctx* ctx, int64_t ticket, int32_t index, int32_t* nodes, int32_t* meas, double* x, double* cnllr, void* P, uint8_t* flags,
                                       int32_t* n_out) {
    MHT_REQUIRE(ctx && ctx->forest && n_out, "mht_forest_chains_fetch: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(ticket >= 0 && ticket < f->chain_next, "mht_forest_chains_fetch: unknown ticket %lld", (long long)ticket);
    Forest::ChainSlot& cs = f->chain_slots[ticket % Forest::CHAIN_SLOTS];
    MHT_REQUIRE(cs.ticket == ticket, "mht_forest_chains_fetch: ticket %lld has expired (%d later ones were issued)", (long long)ticket, Forest::CHAIN_SLOTS);
    MHT_REQUIRE(index >= 0 && index < cs.count, "mht_forest_chains_fetch: chain %d of %d", index, cs.count);
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    if (cs.pending) { MHT_HIP_CHECK(hipEventSynchronize(cs.ev)); cs.pending = false; }
    const ChainRec r = chain_rec(cs.len, cs.pb);
    const char* o = cs.host + (((size_t)cs.count * 4 + 15) & ~(size_t)15) + (size_t)index * r.stride;
    const int n = *(const int32_t*)o;
    *n_out = n;
    if (nodes) memcpy(nodes, o + 8, (size_t)n * 4);
    if (meas) memcpy(meas, o + r.o_m, (size_t)n * 4);
    if (x) memcpy(x, o + r.o_x, (size_t)n * (NX * 8));
    if (cnllr) memcpy(cnllr, o + r.o_c, (size_t)n * 8);
    if (P) memcpy(P, o + r.o_P, (size_t)n * ((size_t)NP * cs.pb));
    if (flags) memcpy(flags, o + r.o_fl, (size_t)n);
    return MHT_OK;
}

extern "C" int mht_forest_set_timing(mht_ctx* ctx, int32_t enable) {
    MHT_REQUIRE(ctx && ctx->forest, "mht_forest_set_timing: no forest");
    Forest* f = ctx->forest;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    if (enable && !f->evp) {
        f->evp = new hipEvent_t[EV_POOL][5];
        for (int k = 0; k < EV_POOL; ++k)
            for (int i = 0; i < 5; ++i) MHT_HIP_CHECK(hipEventCreate(&f->evp[k][i]));
    }
    f->timing = enable != 0;
    f->timed_steps = 0;
    f->ev_slot = 0;
    return MHT_OK;
}

extern "C" int mht_forest_stage_times(mht_ctx* ctx, float* ms5, int32_t* n_steps) {
    MHT_REQUIRE(ctx && ctx->forest && ms5, "mht_forest_stage_times: null argument");
    Forest* f = ctx->forest;
    MHT_REQUIRE(f->timing && f->timed_steps > 0, "mht_forest_stage_times: no timed step pending");
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 5; ++i) ms5[i] = 0.f;
    const int n = f->timed_steps;
    for (int k = 0; k < n; ++k) {
        hipEvent_t* ev = f->evp[(f->ev_slot - 1 - k + 2 * EV_POOL) % EV_POOL];
        float v;
        for (int i = 0; i < 4; ++i) { MHT_HIP_CHECK(hipEventElapsedTime(&v, ev[i], ev[i + 1])); ms5[i] += v; }
        MHT_HIP_CHECK(hipEventElapsedTime(&v, ev[0], ev[4]));
        ms5[4] += v;
    }
    if (n_steps) *n_steps = n;
    f->timed_steps = 0;
    return MHT_OK;
}

// Development / tooling: copy a named internal array of the forest to the host (synchronises).
extern "C" int mht_forest_debug_read(mht_ctx* ctx, const char* name, void* host, int64_t bytes) {
    MHT_REQUIRE(ctx && ctx->forest && name && host && bytes > 0, "mht_forest_debug_read: bad argument");
    Forest* f = ctx->forest;
    const void* src = nullptr;
    size_t avail = 0, offset = 0;
    const size_t T = f->Tcap;
    char nbuf[32];
    if (const char* at = strchr(name, '@')) {      // "array@byte_offset": a window of a large array
        const size_t n = (size_t)(at - name) < sizeof(nbuf) - 1 ? (size_t)(at - name) : sizeof(nbuf) - 1;
        memcpy(nbuf, name, n);
        nbuf[n] = 0;
        offset = (size_t)strtoull(at + 1, nullptr, 10);
        name = nbuf;
    }
    if (!strcmp(name, "vt_rebuilds")) {      // (host-side counter: generation switches of the covariance-value table so far)
        MHT_REQUIRE(bytes == 4, "mht_forest_debug_read: 'vt_rebuilds' is one int32");
        *static_cast<int32_t*>(host) = f->rebuilds;
        return MHT_OK;
    }
    if (!strcmp(name, "uf_ovl")) {      // (host-side counters: scans clustered by the union-find, grow launches made any-order)
        MHT_REQUIRE(bytes == 8, "mht_forest_debug_read: 'uf_ovl' is two int32");
        static_cast<int32_t*>(host)[0] = f->uf_scans; static_cast<int32_t*>(host)[1] = f->ovl_launches;
        return MHT_OK;
    }
    if (!strcmp(name, "status2")) { src = f->status2; avail = 2 * sizeof(DevStatus); }
    else if (!strcmp(name, "init_dbg")) { src = f->init_dbg; avail = 64; }
    else if (!strcmp(name, "cl_status")) { src = f->cl_status; avail = T * 4; }
    else if (!strcmp(name, "cl_iters")) { src = f->cl_iters; avail = T * 4; }
    else if (!strcmp(name, "cl_nodes")) { src = f->cl_nodes; avail = T * 4; }
    else if (!strcmp(name, "cl_time")) { src = f->cl_time; avail = 8 * T * 4; }
    else if (!strcmp(name, "cl_ptr")) { src = f->cl_ptr; avail = (T + 1) * 4; }
    else if (!strcmp(name, "cl_members")) { src = f->cl_members; avail = T * 4; }
    else if (!strcmp(name, "multi_list")) { src = f->multi_list; avail = T * 4; }
    else if (!strcmp(name, "cl_counts")) { src = f->cl_counts; avail = 8 * 4; }
    else if (!strcmp(name, "cl_owner")) { src = f->cl_owner; avail = T * 4; }
    else if (!strcmp(name, "team_list")) { src = f->team_list; avail = TEAM_MAX * 4; }
    else if (!strcmp(name, "tchild")) { src = f->tchild; avail = (T + 1) * 4; }
    else if (!strcmp(name, "tcend")) { src = f->tcend; avail = (T + 1) * 4; }
    else if (!strcmp(name, "Gk")) { src = f->vt.Gk; avail = (size_t)f->vt.vcap * 2 * GKF * 4; }                  // gains by key
    else if (!strcmp(name, "vchild")) { src = f->vt.child; avail = (size_t)f->vt.vcap * 8; }              // value id by key
    else if (!strcmp(name, "vcount")) { src = f->vt.count; avail = 4; }                                 // value ids handed out (current generation)
    else if (!strcmp(name, "cov")) { src = f->layer[f->scan % f->R].cov; avail = (size_t)f->Ncap * 4; }     // keys of the newest layer's nodes
    else if (!strcmp(name, "grow_dbg")) { src = f->grow_dbg; avail = (32 + 16 * 4000) * 8; }
    else if (!strcmp(name, "commit_log")) { src = f->commit_log; avail = 64 * 16 * 4; }
    else if (!strcmp(name, "cluster_dbg")) { src = reinterpret_cast<int32_t*>(f->grow_dbg) + 16; avail = 8 * 4; }
    else if (!strcmp(name, "path")) { src = f->path[f->scan & 1]; avail = (size_t)f->pds * f->Ncap * 4; }      // records of the newest layer
    else if (!strcmp(name, "cost")) { src = f->cost2[f->scan & 1]; avail = (size_t)f->Ncap * 8; }
    MHT_REQUIRE(src, "mht_forest_debug_read: unknown array '%s'", name);
    MHT_REQUIRE(offset <= avail && (size_t)bytes <= avail - offset, "mht_forest_debug_read: '%s' holds %zu bytes", name, avail);
    src = static_cast<const char*>(src) + offset;
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    { const int rc = flush_commit(ctx, f); if (rc) return rc; }
    MHT_HIP_CHECK(hipMemcpyAsync(host, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHT_OK;
}
