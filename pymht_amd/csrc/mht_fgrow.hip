// fgrow_kernel: the grow stage of the device-resident forest -- gate / update / score of every leaf hypothesis against every
// measurement of a scan and the creation of the child hypotheses (reference: Tracker._growTarget / _processLeafNodes,
// pymht/tracker.py:309-351, :383-398, :804-889; children pyTarget.py:227-258, :319-328), ONE launch per scan.
//
// Round-2 design (the stateless seam mht_gate_scan keeps the tile/look-back kernel of mht_gate.hip):
//   * one WORKGROUP PER TARGET (the leaves of a target are one contiguous node range of the previous layer, 32 of them at
//     N-scan 5): no leaf -> target search, one gate bounding box, every per-target quantity is a scalar, and -- because all
//     leaves of a target are in ONE workgroup -- the target's association set is de-duplicated in an LDS bitset (no global
//     bitset, no returning atomic per child).  Targets with more than FG_CAP leaves are processed in chunks (two passes:
//     count, then emit).
//   * children are NOT placed by a device-wide prefix over all leaves any more (a look-back every tile waited 3.7 us in):
//     the children of a target only have to be contiguous and in DFS order AMONG THEMSELVES (pyTarget.getLeafNodes), so a
//     workgroup takes its block with one returning atomicAdd on the child counter of its XCD's region of the (sparse) node
//     index space -- eight regions, neighbours in memory were written through the same L2.  tchild[t] / tcend[t] give the
//     block; nothing downstream needs a dense numbering.
//   * the measurement-independent covariance chain P -> P_bar, S, S^-1, K, P_hat (kalman.py:62, :90-93; ~700 dependent VALU
//     operations per leaf) is off the critical path AND shared: covariances live in a value table of the forest (mht_vtab.h),
//     "chain" workgroups of the same launch resolve, per leaf and per hit/miss, the child's covariance and ITS gains (S^-1, K,
//     score constant, gate half-axes) one scan ahead -- a look-up unless the transition is new -- so a target workgroup only
//     forms x_bar = A x, z_hat = C x_bar and looks its gains up.
//   * launch = [commit of the previous scan (deferred, workgroup 0)] + one workgroup per target slot + chain workgroups.
// Algorithmic bytes of the stage: 280 B per leaf + 48 B per gated pair + 8 B per measurement.
#include "mht_fgrow_dev.h"

namespace mht {

// one sector per launch: the argument blocks travel by value (FGrowArgs first: the workgroups re-read it through the kernarg pointer)
template <int PQ, int CAP = FG_CAP_SOLO>
__global__ __launch_bounds__(FG_THREADS, 3) void fgrow_kernel(const FGrowArgs a, const CommitArgs cm, const FDyn d, const PublishArgs pub) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_grow = d.fused + d.n_main + d.n_chain;
    if ((int)blockIdx.x >= n_grow) { publish_part(pub, (int)blockIdx.x - n_grow); return; }      // (only launched when pub.dst is set)
    fgrow_body<PQ, CAP>((KArgs)__builtin_amdgcn_kernarg_segment_ptr(), cm, d, smem);
}

// The streaming drop-in path (mht_forest_scan with the device initiator): the previous scan's commit, the admission of what its
// initiator gave birth to (Tracker.initiateTarget, tracker.py:147-160 / :264-278) AND the report's push to the host ride in workgroup 0
// of this launch instead of a launch of their own behind the ILPs (post_scan_kernel: ~10 us + a launch gap on the critical path of
// every scan, for an admission that has nothing to admit on 98 % of the headline stream's scans).  The target workgroups of the
// uncommitted table do not depend on any of it.  What does: the newborn targets (slots born0 .. nT - 1 of the committed table), grown
// by FG_BORN_WGS workgroups at the end of the grid, their covariance transitions resolved by FG_BORN_CHAIN_WGS more.  They wait for
// the word workgroup 0 posts in FCounts::adm_flag (scan, first newborn slot, count; workgroup 0 is the first of the grid and waits
// for nobody) and leave at once when the count is zero.  Only a scan WITH births pays for device-scope fences (on this part a release
// writes the XCD's dirty L2 lines back -- megabytes of children in the middle of a grow launch: 6 us when every scan did it).
constexpr int FG_BORN_WGS = 16, FG_BORN_CHAIN_WGS = 8;
static_assert(offsetof(mht_target_report, new_index) == 16 && offsetof(mht_target_report, n_leaves) == 28,
              "what the commit writes of a report row are the first and the last word of its second 16 bytes (fgrow_adm_kernel forms them for the host block itself)");
template <int PQ, int CAP = FG_CAP_SOLO>
__global__ __launch_bounds__(FG_THREADS, 3) void fgrow_adm_kernel(const FGrowArgs a, const CommitArgs cm, const FDyn d, const PublishArgs pub, const AddArgs ad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const KArgs ap = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int bx = (int)blockIdx.x, n_grow = 1 + d.n_main + d.n_chain;      // (d.fused = 1)
    if (bx == 0) {
        int* sm = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) { DevStatus* st = ap->status; st->t[0] = wall_clock64(); st->t[2] = 0; st->t[3] = 0; st->t[4] = 0; }
        if (d.uf_epoch && ap->uf_team_state && threadIdx.x < TEAM_MAX) { ap->uf_team_state[threadIdx.x].gub = ~0ull; ap->uf_team_state[threadIdx.x].done = 0; }
        // (what the initiator confirmed is known since the previous launch: fetched in front of the commit, off the critical path)
        int n_cand = ad.n;
        if (ad.n_dev && !d.adm_wait) { const int nd = *ad.n_dev; n_cand = nd < n_cand ? nd : n_cand; }
#ifdef MHT_ADM_STAMPS
        const unsigned long long ts0 = wall_clock64();
#endif
        const int born0 = commit_body<FG_THREADS>(cm, CommitDyn{d.c_scan, d.c_M, d.c_W, d.ovl ? d.c_wait : 0ull, d.adm_wait}, sm);      // targets alive behind the scan, -1: void scan
        __threadfence_block();
        __syncthreads();
#ifdef MHT_ADM_STAMPS
        const unsigned long long ts1 = wall_clock64();
#endif
        int n_born = 0;
        if (d.adm_wait) {
            // overlapping launch: the previous scan's initiator (a launch of its own on another stream) may still be running.  It reads the
            // used-measurement bytes of its scan: the commit above left them alone (CommitDyn::keep_used), they are cleared behind the wait.
            // (Not in front of the commit: the target workgroups of this launch hold the CUs until the commit has posted their indices, and
            // the initiator's 1024-thread workgroup needs a CU.)
            unsigned long long v;
            const bool ok = spin_until(&cm.cnt->init_flag, [&](unsigned long long x) { return x == (unsigned long long)(unsigned)d.c_scan; }, v);
            if (!ok && threadIdx.x == 0) { ap->status->overflow = 2; atomicOr(&ap->status->pad[0], 1 << 4); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (ad.n_dev) { const int nd = __hip_atomic_load(ad.n_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); n_cand = nd < n_cand ? nd : n_cand; }
            for (int jm = threadIdx.x; jm < d.c_M; jm += FG_THREADS) cm.used_bytes[jm] = 0;
        }
        if (born0 >= 0 && n_cand > 0) {                      // (void scan: nothing is admitted; no candidates: the usual scan)
            add_targets_body<FG_THREADS>(ad, sm + 64);
            __threadfence_block();
            __syncthreads();
            n_born = cm.cnt->nT - born0;
        }
        if (n_born > 0) __threadfence();      // the newborn targets' workgroups sit on other CUs / XCDs
        if (threadIdx.x == 0)
            __hip_atomic_store(&cm.cnt->adm_flag, ((unsigned long long)(unsigned)d.c_scan << 32) | ((unsigned long long)(born0 < 0 ? 0 : born0) << 16) | (unsigned long long)n_born,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (pub.dst) {
            // the report: its rows were written by the previous launch (blp_kernel: per-target results) except two words (new index,
            // leaves kept), which the commit above filled in -- and which follow from that launch's results as well.  FG_PUB_WGS
            // workgroups push the rows from the start of the launch, each forming those two words for its rows itself; this workgroup
            // pushes the rest of what it wrote: header, used-measurement mask, births.  Disjoint bytes: nobody waits for anybody.
            const uint4* s4 = reinterpret_cast<const uint4*>(pub.src);
            uint4* d4 = reinterpret_cast<uint4*>(pub.dst);
            const int nb_rep = (born0 >= 0 && n_cand > 0) ? reinterpret_cast<const ReportHeader*>(pub.src)->n_births : 0;
            const int head = (pub.birth_off + nb_rep * (int)sizeof(mht_birth_report) + 15) / 16;      // (head <= rec_off / 16)
            for (int i = threadIdx.x; i < head; i += FG_THREADS) d4[i] = s4[i];
            pub_done(pub, 0);
        }
#ifdef MHT_ADM_STAMPS
        __syncthreads();
        if (threadIdx.x == 0 && cm.log) { int32_t* g = cm.log + (d.c_scan & 63) * 16; const unsigned long long ts3 = wall_clock64();
            g[12] = (int)(ts1 - ts0); g[13] = (int)(ts3 - ts1); g[14] = (int)(ts0 - ap->status->t[0]); g[15] = (int)(ts3 & 0x7fffffff); }
#endif
        return;
    }
    if (pub.dst && bx <= FG_PUB_WGS) {      // the rows of the report (see workgroup 0): workgroup w the rows [w, w + 1) * ceil(nT / FG_PUB_WGS)
        constexpr int RQ = (int)sizeof(mht_target_report) / 16, RB = RQ - 1;
        if (d.adm_wait && d.c_wait) {      // overlapping launch: the rows are the previous scan's ILP launch's, which may still be running
            unsigned long long v;
            if (!spin_until(&cm.cnt->blp_done, [&](unsigned long long x) { return x >= d.c_wait; }, v) && threadIdx.x == 0) { ap->status->overflow = 2; atomicOr(&ap->status->pad[0], 1 << 5); }      // (gave up: the rows may be incomplete -- this scan is void and the forest dead, like every other wait that times out)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const int nTr = ap->nT_dev[0];      // (rows of the report = slots of the uncommitted table)
        const uint4* s4 = reinterpret_cast<const uint4*>(pub.src);
        uint4* d4 = reinterpret_cast<uint4*>(pub.dst);
        const int r0 = pub.rec_off / 16;
        const int per = (nTr + FG_PUB_WGS - 1) / FG_PUB_WGS, t0 = (bx - 1) * per, t1 = (t0 + per < nTr) ? t0 + per : nTr;
        // everything but the second 16 bytes of a row, as the previous launch left it
        for (int i = threadIdx.x; i < (t1 - t0) * RB; i += FG_THREADS) { const int c = i % RB, k = r0 + (t0 + i / RB) * RQ + (c ? c + 1 : 0); d4[k] = s4[k]; }
        // the second 16 bytes = {new index, root scan, root node, leaves kept}: what commit_body computes for the device block (compacted
        // index = targets alive before this one, -1 for a terminated target; its surviving leaves), from the same per-target results
        int* sw = reinterpret_cast<int*>(smem);      // [FG_THREADS / 64 + 1]
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        int before = 0;
        for (int i = tid; i < t0; i += FG_THREADS) before += (ap->p_status[i] == 0) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
        if (lane == 0) sw[wv] = before;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < FG_THREADS / 64; ++q) base += sw[q];
        for (int c0 = t0; c0 < t1; c0 += FG_THREADS) {
            const int t = c0 + tid;
            const bool in = t < t1;
            const int al = (in && ap->p_status[in ? t : 0] == 0) ? 1 : 0;
            const int leaves = al ? ap->p_count[t] : 0;
            const unsigned long long bal = __ballot(al);
            __syncthreads();
            if (lane == 0) sw[wv] = __popcll(bal);
            __syncthreads();
            int off = base, tot = 0;
            for (int q = 0; q < FG_THREADS / 64; ++q) { if (q < wv) off += sw[q]; tot += sw[q]; }
            if (in) {
                uint4 v = s4[r0 + t * RQ + 1];
                v.x = (unsigned)(al ? off + __popcll(bal & ((1ull << lane) - 1ull)) : -1);
                v.w = (unsigned)leaves;
                d4[r0 + t * RQ + 1] = v;
            }
            base += tot;
        }
        pub_done(pub, bx);
        return;
    }
    const int bg = bx - (pub.dst ? FG_PUB_WGS : 0);      // index among the grow workgroups
    if (bg < n_grow) { fgrow_body<PQ, CAP>(ap, cm, d, smem, bg); return; }
    const int w = bg - n_grow;
    unsigned long long& s_flag = *reinterpret_cast<unsigned long long*>(smem);      // (no static LDS: it would shift the dynamic base off its alignment)
    if (threadIdx.x == 0) {
        unsigned long long v;
        while (((v = __hip_atomic_load(&cm.cnt->adm_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (unsigned long long)(unsigned)d.c_scan) __builtin_amdgcn_s_sleep(8);
        s_flag = v;
    }
    __syncthreads();
    const unsigned long long fl = s_flag;
    __syncthreads();                       // (the LDS is the target's from here on)
    const int b0 = (int)((fl >> 16) & 0xffffu), nb = (int)(fl & 0xffffu);
    if (nb == 0) return;                   // nothing was born (the usual case)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (every wavefront)
    const int nT = b0 + nb;
    FDyn db = d;
    db.fused = 0;                          // (the newborn targets are slots of the COMMITTED table)
    if (d.ovl && d.uf_epoch) {             // (the old targets redo their union-find under the alternative epoch when a target died: the newborn ones join that one)
        const unsigned long long ni = __hip_atomic_load(ap->ni_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (posted by the commit, in front of the admission's flag)
        if ((unsigned)ni == (unsigned)d.c_scan && ((ni >> 32) & 1ull)) db.uf_epoch = d.uf_epoch | 1u;
    }
    if (w < FG_BORN_WGS) {
        const int n_old = ap->nT_dev[0];   // slots of the uncommitted table: the static blocks of the node index space behind them are free
        for (int q = w; b0 + q < nT; q += FG_BORN_WGS) {
            target_part<PQ, CAP>(ap, db, b0 + q, smem, n_old + q, 1);
            __syncthreads();
        }
    } else {
        for (int c = w - FG_BORN_WGS; b0 + c * FG_CHAIN_TARGETS < nT; c += FG_BORN_CHAIN_WGS) chain_part(*ap, db, c, b0, 1);
    }
}

// AIS forest (mht_forest_create_ex with MHT_FOREST_AIS): split path records, identities per node, fused children on scans with messages
template <int PQ>
__global__ __launch_bounds__(FG_THREADS, 2) void fgrow_ais_kernel(const FGrowArgs a, const CommitArgs cm, const FDyn d, const PublishArgs pub) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_grow = d.fused + d.n_main + d.n_chain;
    if ((int)blockIdx.x >= n_grow) { publish_part(pub, (int)blockIdx.x - n_grow); return; }
    fgrow_body<PQ, FG_CAP, CommitArgs, 1>((KArgs)__builtin_amdgcn_kernarg_segment_ptr(), cm, d, smem);
}

// Constant-turn forest (six-state build, MHT_FOREST_CT): predictions and gains per leaf come from forest_ct_kernel, the children's keys
// name the leaf; no chain workgroups (nothing is shared by value)
#if MHT_NX == 6
template <int PQ>
__global__ __launch_bounds__(FG_THREADS, 3) void fgrow_ct_kernel(const FGrowArgs a, const CommitArgs cm, const FDyn d, const PublishArgs pub) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_grow = d.fused + d.n_main + d.n_chain;
    if ((int)blockIdx.x >= n_grow) { publish_part(pub, (int)blockIdx.x - n_grow); return; }
    fgrow_body<PQ, FG_CAP, CommitArgs, 0, false, 1>((KArgs)__builtin_amdgcn_kernarg_segment_ptr(), cm, d, smem);
}
#endif

// a group of sectors per launch (BASELINE config 4 on one GPU): blockIdx.y = sector, its argument blocks are read from HBM (they
// repeat with period 2 x ring length and are written once, at group creation), only FDyn travels by value
typedef const __attribute__((address_space(4))) CommitArgs* KCommit;
template <int PQ, int CAP = FG_CAP>      // CAP = 0: the wavefront-per-target variant
__global__ __launch_bounds__(FG_THREADS, NX == 4 ? 4 : 3) void fgrow_batch_kernel(const FBatch b) {      // (six states: 165 registers, three workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int y = blockIdx.y;
    const FDyn d = b.d[y];
    if ((int)blockIdx.x >= d.fused + d.n_main + d.n_chain) return;
    fgrow_body_lean<PQ, CAP>((KArgs)b.ga[y], *(KCommit)b.ca[y], d, smem);
}

size_t fgrow_lds_bytes_cap(int W, int pds, int AW, int cap) {
    size_t b = (size_t)2 * W * 64 * 4 + (size_t)cap * sizeof(FLeaf) + (size_t)2 * pds * cap * 4 + (size_t)cap * W * 8 + (size_t)AW * 8 +
               (size_t)(cap + 4) * 4 + 128 + (size_t)W * 64 * 2 + FG_MAP;
    if (b < 256) b = 256;      // the commit workgroup keeps its scan partials here
    return (b + 15) & ~(size_t)15;
}

// fgrow_ct_kernel: hit masks over the target's candidate list (FG_HWC words per leaf) and a conflict list of its own (see target_part)
static size_t fgrow_ct_lds_bytes(int W, int pds, int AW) {
    const int cap = FG_CAP;
    size_t b = (size_t)2 * W * 64 * 4 + (size_t)cap * sizeof(FLeaf) + (size_t)2 * pds * cap * 4 + (size_t)cap * FG_HWC * 8 + (size_t)AW * 8 +
               (size_t)(cap + 4) * 4 + 128 + (size_t)W * 64 * 2 + FG_MAP + (size_t)FG_CONF * 4;
    if (b < 256) b = 256;
    return (b + 15) & ~(size_t)15;
}
size_t fgrow_lds_bytes(int W, int pds, int AW) { return fgrow_lds_bytes_cap(W, pds, AW, FG_CAP); }      // (the batched launch, workgroup per target)
size_t fgrow_wave_lds_bytes(int W, int pds, int AW) {      // (the batched launch, wavefront per target: four slices)
    size_t b = (size_t)(FG_THREADS / 64) * fw_layout(pds, AW, W * 64).total;
    return b < 256 ? 256 : b;
}

static int fgrow_lds_attr(mht_ctx* ctx, size_t lds) {
    if (lds > 150 * 1024) {
        set_error("fgrow: %zu bytes of LDS per workgroup (max_meas / window too large)", lds);
        return MHT_E_CAPACITY;
    }
    size_t& attr_bytes = ctx->lds_attr_fgrow;
    if (lds > 48 * 1024 && lds > attr_bytes) {
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_kernel<2, FG_CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_kernel<4, FG_CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_batch_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_batch_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_batch_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_batch_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_adm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_adm_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_adm_kernel<2, FG_CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_adm_kernel<4, FG_CAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_ais_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_ais_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#if MHT_NX == 6
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_ct_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MHT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fgrow_ct_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
        attr_bytes = lds;
    }
    return MHT_OK;
}

// grid of one sector: [commit] + one workgroup per target slot + chain workgroups
static inline int fgrow_grid(const FDyn& d) { return d.fused + d.n_main + d.n_chain; }

// wave: the wavefront-per-target variant (four target slots per workgroup)
void fgrow_plan(FDyn& d, int n_targets_ub, int Tcap, bool fused, bool wave) {
    d.fused = fused ? 1 : 0;
    int n_tgt = n_targets_ub < 1 ? 1 : n_targets_ub;
    if (n_tgt > Tcap) n_tgt = Tcap;
    d.n_tgt = n_tgt;
    d.n_main = wave ? (n_tgt + FG_THREADS / 64 - 1) / (FG_THREADS / 64) : n_tgt;
#ifdef MHT_FW_NOFOLD
    wave = false;
#endif
    d.n_chain = wave ? 0 : (n_tgt + FG_CHAIN_TARGETS - 1) / FG_CHAIN_TARGETS;      // (the wavefronts of the wave variant resolve their own transitions)
}

// any_order: the launch may start while the launch in front of it in the stream -- the previous scan's ILP launch -- is still running
// (hipExtAnyOrderLaunch; FDyn::ovl: what it needs of that launch it waits for itself, target by target)
int launch_fgrow(mht_ctx* ctx, const FGrowArgs& a, FDyn& d, int n_targets_ub, const CommitArgs* commit, const PublishArgs* publish, const AddArgs* adm, bool any_order) {
    static int wave_solo = -1;      // development: MHT_FG_WAVE_SOLO=1 runs the wavefront-per-target variant in the one-sector launch too
    if (wave_solo < 0) { const char* e = getenv("MHT_FG_WAVE_SOLO"); wave_solo = (e && e[0] == '1') ? 1 : 0; }
    if (adm) {      // the commit and the admission of the initiator's births ride along (fgrow_adm_kernel)
        MHT_REQUIRE(commit && a.ais.half == 0, "launch_fgrow: the admission rides with the commit, in a forest without AIS records");
        fgrow_plan(d, n_targets_ub, a.Tcap, true, false);
        const size_t lds_hi = fgrow_lds_bytes_cap(d.W, a.pds, a.AW, FG_CAP_SOLO), lds_lo = fgrow_lds_bytes_cap(d.W, a.pds, a.AW, FG_CAP);
        auto per_cu = [](size_t b) { const size_t n = (size_t)160 * 1024 / b; return n > 3 ? (size_t)3 : n; };
        const bool wide = per_cu(lds_hi) >= per_cu(lds_lo);
        size_t lds = wide ? lds_hi : lds_lo;
        const size_t need0 = (size_t)(64 + ADM_LDS_INTS) * sizeof(int);      // workgroup 0: commit partials + admission list
        if (lds < need0) lds = need0;
        { const int rc = fgrow_lds_attr(ctx, lds); if (rc) return rc; }
        const bool pub = publish && publish->dst;
        const int grid = fgrow_grid(d) + FG_BORN_WGS + FG_BORN_CHAIN_WGS + (pub ? FG_PUB_WGS : 0);
        const PublishArgs pa = pub ? *publish : PublishArgs{};
        if (any_order && d.ovl && d.adm_wait) {
            const dim3 g(grid), b(FG_THREADS);
            if (a.pds == 8 && wide) hipExtLaunchKernelGGL((fgrow_adm_kernel<2>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, *commit, d, pa, *adm);
            else if (a.pds == 8) hipExtLaunchKernelGGL((fgrow_adm_kernel<2, FG_CAP>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, *commit, d, pa, *adm);
            else if (wide) hipExtLaunchKernelGGL((fgrow_adm_kernel<4>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, *commit, d, pa, *adm);
            else hipExtLaunchKernelGGL((fgrow_adm_kernel<4, FG_CAP>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, *commit, d, pa, *adm);
            MHT_HIP_CHECK(hipGetLastError());
            return MHT_OK;
        }
        if (a.pds == 8 && wide) hipLaunchKernelGGL(fgrow_adm_kernel<2>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, *commit, d, pa, *adm);
        else if (a.pds == 8) hipLaunchKernelGGL((fgrow_adm_kernel<2, FG_CAP>), dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, *commit, d, pa, *adm);
        else if (wide) hipLaunchKernelGGL(fgrow_adm_kernel<4>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, *commit, d, pa, *adm);
        else hipLaunchKernelGGL((fgrow_adm_kernel<4, FG_CAP>), dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, *commit, d, pa, *adm);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
#if MHT_NX == 6
    if (a.ct.on) {      // constant-turn forest: its own kernel, no chain workgroups
        fgrow_plan(d, n_targets_ub, a.Tcap, commit != nullptr, false);
        d.n_chain = 0;
        const size_t lds = fgrow_ct_lds_bytes(d.W, a.pds, a.AW);
        { const int rc = fgrow_lds_attr(ctx, lds); if (rc) return rc; }
        const bool pub = publish && publish->dst;
        const int grid = fgrow_grid(d) + (pub ? FG_PUB_WGS : 0);
        const PublishArgs pa = pub ? *publish : PublishArgs{};
        const CommitArgs cm = commit ? *commit : CommitArgs{};
        if (a.pds == 8) hipLaunchKernelGGL(fgrow_ct_kernel<2>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
        else hipLaunchKernelGGL(fgrow_ct_kernel<4>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
#endif
    if (a.ais.half > 0) {      // AIS forest: its own kernel on every scan (records in two halves, identities per node)
        fgrow_plan(d, n_targets_ub, a.Tcap, commit != nullptr, false);
        const size_t lds = fgrow_lds_bytes_cap(d.W, a.pds, a.AW, FG_CAP) + (size_t)FG_CAP * (16 + sizeof(FLeafX));      // (+ s_ais, lgx)
        { const int rc = fgrow_lds_attr(ctx, lds); if (rc) return rc; }
        const bool pub = publish && publish->dst;
        const int grid = fgrow_grid(d) + (pub ? FG_PUB_WGS : 0);
        const PublishArgs pa = pub ? *publish : PublishArgs{};
        const CommitArgs cm = commit ? *commit : CommitArgs{};
        if (a.pds == 8) hipLaunchKernelGGL(fgrow_ais_kernel<2>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
        else hipLaunchKernelGGL(fgrow_ais_kernel<4>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
    fgrow_plan(d, n_targets_ub, a.Tcap, commit != nullptr, wave_solo != 0);
    if (wave_solo) {
        const size_t lds = fgrow_wave_lds_bytes(d.W, a.pds, a.AW);
        { const int rc = fgrow_lds_attr(ctx, lds); if (rc) return rc; }
        const bool pub = publish && publish->dst;
        const int grid = fgrow_grid(d) + (pub ? FG_PUB_WGS : 0);
        const PublishArgs pa = pub ? *publish : PublishArgs{};
        const CommitArgs cm = commit ? *commit : CommitArgs{};
        if (a.pds == 8) hipLaunchKernelGGL((fgrow_kernel<2, 0>), dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
        else hipLaunchKernelGGL((fgrow_kernel<4, 0>), dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
    // 128 leaves per pass unless the larger tables cost a workgroup per CU (long scans: the hit masks grow with the scan): 3 per CU is
    // all the launch bounds allow, fewer than with 96 leaves per pass is a loss (config-5 size, 2 048 measurements: 1 instead of 2)
    const size_t lds_hi = fgrow_lds_bytes_cap(d.W, a.pds, a.AW, FG_CAP_SOLO), lds_lo = fgrow_lds_bytes_cap(d.W, a.pds, a.AW, FG_CAP);
    auto per_cu = [](size_t b) { const size_t n = (size_t)160 * 1024 / b; return n > 3 ? (size_t)3 : n; };
    const bool wide = per_cu(lds_hi) >= per_cu(lds_lo);
    const size_t lds = wide ? lds_hi : lds_lo;
    { const int rc = fgrow_lds_attr(ctx, lds); if (rc) return rc; }
    const bool pub = publish && publish->dst;
    const int grid = fgrow_grid(d) + (pub ? FG_PUB_WGS : 0);
    const PublishArgs pa = pub ? *publish : PublishArgs{};
    const CommitArgs cm = commit ? *commit : CommitArgs{};
    if (any_order && d.fused && d.ovl && !pub) {
        const dim3 g(grid), b(FG_THREADS);
        if (a.pds == 8 && wide) hipExtLaunchKernelGGL((fgrow_kernel<2>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, cm, d, pa);
        else if (a.pds == 8) hipExtLaunchKernelGGL((fgrow_kernel<2, FG_CAP>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, cm, d, pa);
        else if (wide) hipExtLaunchKernelGGL((fgrow_kernel<4>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, cm, d, pa);
        else hipExtLaunchKernelGGL((fgrow_kernel<4, FG_CAP>), g, b, lds, ctx->stream, nullptr, nullptr, hipExtAnyOrderLaunch, a, cm, d, pa);
        MHT_HIP_CHECK(hipGetLastError());
        return MHT_OK;
    }
    if (a.pds == 8 && wide) hipLaunchKernelGGL(fgrow_kernel<2>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
    else if (a.pds == 8) hipLaunchKernelGGL((fgrow_kernel<2, FG_CAP>), dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
    else if (wide) hipLaunchKernelGGL(fgrow_kernel<4>, dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
    else hipLaunchKernelGGL((fgrow_kernel<4, FG_CAP>), dim3(grid), dim3(FG_THREADS), lds, ctx->stream, a, cm, d, pa);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

int launch_fgrow_batch(mht_ctx* ctx, const FBatch& b, int n_sectors, int grid_x, size_t lds, int pds, bool wave) {
    { const int rc = fgrow_lds_attr(ctx, lds); if (rc) return rc; }
    if (wave && pds == 8) hipLaunchKernelGGL((fgrow_batch_kernel<2, 0>), dim3(grid_x, n_sectors), dim3(FG_THREADS), lds, ctx->stream, b);
    else if (wave) hipLaunchKernelGGL((fgrow_batch_kernel<4, 0>), dim3(grid_x, n_sectors), dim3(FG_THREADS), lds, ctx->stream, b);
    else if (pds == 8) hipLaunchKernelGGL(fgrow_batch_kernel<2>, dim3(grid_x, n_sectors), dim3(FG_THREADS), lds, ctx->stream, b);
    else hipLaunchKernelGGL(fgrow_batch_kernel<4>, dim3(grid_x, n_sectors), dim3(FG_THREADS), lds, ctx->stream, b);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}

int fgrow_grid_of(const FDyn& d) { return fgrow_grid(d); }

}  // namespace mht
