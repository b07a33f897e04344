// Kernel argument blocks and launchers shared by the translation units of libmht_amd.so.
#pragma once
#include "mht_common.h"
#include "mht_vtab.h"

namespace mht {

struct GateArgs {
    Model model;
    double default_pd, default_miss_nllr;
    // input layer
    const double* x; const double* cnllr; const double* pd; const int32_t* cov; const uint8_t* flags; const float* P;
    int cap_in, capc_in;
    const int32_t* leaf_src;      // explicit leaf list (or null = identity)
    int L;                        // number of leaves
    const float* z; int M; int W;
    // cross-workgroup machinery of grow_kernel
    int32_t* ticket;              // tile ticket counter (zero at launch)
    int max_resident;             // workgroups that are co-resident; with more tiles than that they are numbered by the ticket
    unsigned long long* tile_state;   // [ntiles] epoch-tagged child counts of the tiles
    unsigned long long* group_state;  // [ntiles/64] epoch-tagged sums over groups of 64 tiles
    unsigned epoch;
    unsigned long long* dbg;      // development only: [8] summed wall-clock ticks per phase over all tiles, [8..15] max
    int ablate;                   // development only (env MHT_GROW_ABLATE): knock out phases for timing experiments
    // output layer
    double* ox; double* ocnllr; double* opd; int32_t* oparent; int32_t* omeas; int32_t* ocov; uint8_t* oflags; float* oP;
    int cap_out, capc_out;
    int32_t* child_ptr; double* nllr; unsigned long long* used;
    DevStatus* status;
};

struct CommitArgs;
struct FDyn;
struct FBatch;
struct TeamState;

// AIS-aided children in the forest (tracker.py:417-552; mht_ais.hip).  A forest made with MHT_FOREST_AIS keeps two more words per
// node -- the identity of the AIS message the node was updated with (mmsi, 0 = none) and the identity its track is bound to (hmmsi:
// its own or its nearest AIS-updated ancestor's, pyTarget.py:297-302) -- and splits a path record in two halves: entries
// [0, half) are the radar measurement nodes per level, [half, 2 half) the AIS message nodes (a fused child has both at its level;
// to the ILP they are all rows, tracker.py:1046-1090).  On a scan with messages forest_ais_kernel runs in front of the grow launch
// and leaves, per leaf of the newest layer, the number of its fused children and where their records start; the grow kernel appends
// them behind the leaf's radar children (pyTarget.py:260-295).
struct AisRec {
    double x[4];        // state (float64: ais.C is float64)
    double nllr;        // score increment (nllr_ais + nllr_radar) / 2, or nllr_ais alone
    int32_t radar;      // 0-based radar measurement or -1
    int32_t msg;        // index of the message in the scan's (grouped) list: its measurement node is M + msg
    int32_t key;        // covariance key in the value table (a pseudo parent's miss child, like a root's)
    int32_t mmsi;
};
struct AisGrow {
    const int32_t* nf; const int32_t* off; const AisRec* rec;      // [cap] by node of the input layer; record pool
    const int32_t* hmmsi_in;                                       // input layer
    int32_t* ommsi; int32_t* ohmmsi;                               // output layer
    int half;                                                      // levels per half of a path record (0: not an AIS forest)
    // the window column of the target table the scan runs on (fused launch: the uncommitted one, by old slot): bits 8.. = levels of the
    // target's tree its association set has been REBUILT from (mht_commit.h: WIN_REBUILT; pyTarget.py:292-295 vs :414-430)
    const int32_t* t_window;
};
// The window column of the target table: bits 0-7 = the target's N-scan window; bits 8-15 = RL, the number of tree levels below the root
// that the reference's association set of the target has been rebuilt from (Target.getMeasurementSet, tracker.py:1226 / :1238): 0 for a
// target whose set is still the one spawnNewNodes built incrementally -- it lacks the radar measurements of fused children,
// pyTarget.py:292-295 -- WIN_REBUILT_ALL once the root has advanced (rebuilt at the end of every scan from then on), the children's level
// + 1 when similar-state pruning rebuilt it for a target alone in its cluster (tracker.py:1233-1239).  Only AIS forests look at RL.
constexpr int WIN_REBUILT_SHIFT = 8, WIN_REBUILT_ALL = 255;

// Constant-turn forest (six-state build, mht_forest_create_ex with MHT_FOREST_CT; BASELINE config 5's model, pymht_amd/models/ct.py): the
// transition Phi(T, w) depends on the hypothesis' own turn rate, so nothing is shared by value -- the reference's per-hypothesis form,
// kalman.predict_single + kalman.precalc on a batch of one (kalman.py:67-70, :82-101), runs for every leaf in forest_ct_kernel in front
// of the grow launch and leaves, per leaf NODE of the input layer: its prediction, its gains row (the layout of VTab::Gk) and the
// covariances of its children (P_bar for the missed detection, P_hat for every hit).  A node names its covariance by
// key = 2 * (parent node) + hit/miss into the parent LAYER's arrays; a root by -2 - r into its own layer's root array.
struct CtGrow {
    int on;
    const float4* gains;      // [cap][GKQ]
    const double* xbar;       // [NX][cap]
    const double* zhat;       // [2][cap]
    unsigned long long* hw_spill;      // [Tcap][FG_CAP][ceil(max_meas / 64)] full-width hit masks of a target whose candidate list outgrows the LDS words (fgrow_ct_kernel: FG_HWC)
};
struct CtForestArgs {
    Model model; double T;
    const int32_t* nT_dev; const int32_t* t_first; const int32_t* t_leaf_off;      // the committed target table ...
    int fused; const int32_t* p_status; const int32_t* p_count; const int32_t* p_firstsurv;      // ... or (fused) the uncommitted one: the previous scan's per-target results by old slot, as FGrowArgs::p_* (the commit then rides in the grow launch)
    const double* x; const double* pd; const int32_t* cov; const uint8_t* flags; int cap;      // the leaves' layer
    const float* Pbar_prev; const float* Phat_prev; const float* Proot;      // what the leaves' keys resolve against: the layer before / their own layer's roots
    float* Pbar; float* Phat;      // [cap][NP] out: the covariances of the leaves' children
    float4* gains; double* xbar; double* zhat;
};
int launch_forest_ct(mht_ctx* ctx, const CtForestArgs& a, int n_targets_ub);

// fgrow_kernel (mht_fgrow.hip): the grow stage of the forest, one workgroup per target + covariance-chain workgroups
struct FGrowArgs {
    Model model;
    double default_pd, default_miss_nllr;
    // input layer (the previous scan's nodes); cap is the same for every layer of the ring
    const double* x; const double* cnllr; const double* pd; const int32_t* cov; const uint8_t* flags;
    int cap;                       // cov = key into the forest's covariance-value table (mht_vtab.h)
    VTab vt;
    const int32_t* in_path;        // [cap][pds] measurement nodes below the root, one record per node of the input layer
    const int32_t* in_apath;       // [cap][pds] ancestor node per level
    int pds;                       // ints per record: 8 (PD <= 8) or 16
    // the target table this scan runs on.  fused (FDyn) = 1: the commit of the previous scan has not run (it rides in workgroup 0):
    // the per-target results of that scan (p_*), indexed by old slot, stand in for the compacted table
    const int32_t* nT_dev;
    int Tcap;
    const int32_t* p_status; const int32_t* p_count; const int32_t* p_jdrop; const int32_t* p_firstsurv; const int32_t* p_depth;
    const int32_t* t_first; const int32_t* t_leaf_off; const int32_t* t_depth; const int32_t* t_shift;
    const double* t_root_cnllr; const uint8_t* t_root_f32;      // by slot of the table named above
    // targets admitted INSIDE this launch (fgrow_adm_kernel: workgroup 0 runs the previous scan's commit and the admission of what the
    // initiator gave birth to; the newborn targets are slots of the committed table): its count and root columns
    const int32_t* nT_new; const double* b_root_cnllr; const uint8_t* b_root_f32;
    // output layer
    double* ox; double* ocnllr; double* opd; int32_t* oparent; int32_t* omeas; int32_t* ocov; uint8_t* oflags;
    int32_t* out_path; int32_t* out_apath; double* ocost;
    int32_t* tchild; int32_t* tcend;      // [T] children of (compacted) target t: tchild[t] .. tcend[t]-1
    int PD; int Nwin; int cur_slot_base; int AW;
    unsigned* alloc;               // [FG_REGIONS][32] child counter of every region (a cache line each), zero at launch
    int block_cap;                 // node indices of the static block of every target slot: slot t owns [t * block_cap, (t+1) * block_cap)
    int over_base, region_cap;     // overflow area behind the static blocks: region r = over_base + [r * region_cap, (r+1) * region_cap)
    unsigned* edges; int32_t* edge_count; int edge_cap;
    unsigned char* used_bytes;
    DevStatus* status;             // this scan's status word: n_children is accumulated here
    const DevStatus* prev_status; const int32_t* sticky_overflow;
    AisGrow ais;                   // (fgrow_kernel<..., AIS = 1> only)
    CtGrow ct;                     // (fgrow_ct_kernel only)
    // clustering inside the grow launch (FDyn::uf_epoch != 0, see there): owner word per measurement node, parent word per target
    unsigned long long* uf_owner; unsigned long long* uf_parent;
    TeamState* uf_team_state;      // [TEAM_MAX] reset for this scan's ILP launch by the launch's first workgroup (the cluster kernel did it)
    // a grow launch that overlaps the previous scan's ILP launch (FDyn::ovl): that launch's per-target records (TGT_REC_*), the compacted
    // indices its commit -- workgroup 0 of THIS launch -- computes, and the word that commit posts when they are valid
    const unsigned long long* rec0; const int32_t* new_index; const unsigned long long* ni_flag;
};
// The per-target record blp_uf_kernel publishes for the NEXT scan's grow launch the moment a target is finished (one 8-byte store, written
// through; the root's cumulative score goes out in front of it): the next grow launch may be running already (FDyn::ovl) and its
// workgroup for the target waits for nothing else.  tag = scan & 0xff | alive | root score is float32 | layers the root advances | surviving
// leaves | first surviving leaf
constexpr int TGT_REC_TAG = 56, TGT_REC_ALIVE = 55, TGT_REC_RF = 54, TGT_REC_J = 50, TGT_REC_CNT = 27;
__host__ __device__ __forceinline__ unsigned long long tgt_rec(unsigned scan, int alive, int rf, int j, int count, int first) {
    return ((unsigned long long)(scan & 0xffu) << TGT_REC_TAG) | ((unsigned long long)(alive ? 1 : 0) << TGT_REC_ALIVE) | ((unsigned long long)(rf ? 1 : 0) << TGT_REC_RF) |
           ((unsigned long long)(j & 15) << TGT_REC_J) | ((unsigned long long)(count & 0x7fffff) << TGT_REC_CNT) | (unsigned long long)(first & 0x7ffffff);
}
// What changes from scan to scan (everything in FGrowArgs repeats with period 2 x ring length, for fused = 0 and 1): passed by
// value next to the argument block (one launch per sector) or to a pointer to it (one launch for a group of sectors).
struct FDyn {
    const float* z; int M; int W;  // the scan: dev (M,2) float32; W = ceil(M / 64)
    int fused;                     // workgroup 0 runs the previous scan's commit (CommitDyn c)
    int n_main;                    // workgroups [fused, fused + n_main): the targets (one slot each, or four -- one per wavefront); then n_chain
    int n_tgt, n_chain;            // target slots covered; covariance-chain workgroups (two targets each)
    int c_scan, c_M, c_W;          // CommitDyn of the commit that rides along
    int maybe_dead;                // similar-state pruning ran on the previous scan: leaves may carry F_DEAD (a target's LIVE leaf count decides gemm / gemv order)
    int ais_on;                    // AIS forest: this scan carries messages (AisGrow::nf / off / rec are valid)
    unsigned long long* dbg;       // development only (-DMHT_GROW_STAMPS): [workgroup][16] wall-clock ticks at phase boundaries
    int ovl;                       // fused = 1 only: the previous scan's ILP launch may still be running.  Target and chain workgroups take their
                                   // target's results from its record (FGrowArgs::rec0, waiting for it), the commit waits for that launch's
                                   // workgroups (c_wait) and posts the compacted indices the target workgroups end with
    // the scan was staged by a small kernel on the forest's side stream and the ctx stream did NOT wait for it (streamed path): whoever
    // reads the scan first waits until z_flag[0] == z_tag (stage_scan_kernel posts it behind its written-through stores); 0: nothing to wait for
    const unsigned long long* z_flag; unsigned long long z_tag;
    int adm_wait;                  // fgrow_adm_kernel launched any-order: the admission waits for the previous scan's initiator (FCounts::init_flag), the
                                   // report's workgroups for the previous scan's ILP launch (c_wait)
    int stamp_end;                 // development (MHT_OVL_STAMPS=1): the target workgroups leave their end time in DevStatus::t[5] (atomic max)
    int ct_spill;                  // testing (MHT_CT_SPILL=1): fgrow_ct_kernel keeps every target's hit masks in the global spill block
    unsigned long long c_wait;     // FCounts::blp_done the commit waits for (0: the ILP launch has ended, as stream order says)
    unsigned uf_epoch;             // != 0 (2 x the scan number): no edge list -- the target workgroups hook their targets into a device-wide
                                   // union-find over the measurement nodes they use, and the ILP launch derives the clusters from it
};
// Clustering without a clustering launch (tracker.py:961-974).  The connected components of targets <-> measurement nodes are the
// fixed point of "two targets that use the same node belong together".  Every target workgroup of the grow launch already holds its
// target's de-duplicated association set; per node it exchanges ONE 64-bit word {epoch, target} (agent-scope atomic max): whoever finds
// a word of this scan there unites its target with the one named, in a lock-free union-find over the targets (64-bit words {epoch,
// parent}, the larger root hooked under the smaller by compare-and-swap, so that a component's root is its smallest member -- the
// reference's cluster order).  Words of earlier scans are "empty" / "root": nothing is cleared between scans.  The workgroups of the
// ILP launch read the parents (a kernel boundary later) and each derives the cluster tables in LDS for itself (mht_blp.hip:
// uf_prologue): no clustering kernel, no launch boundary, and three dependent look-ups fewer in front of every ILP.
// The scan report on its way to pinned, device-mapped host memory (mht_forest.hip: publish_report): device block -> host block.
struct PublishArgs { const char* src; char* dst; int rec_off, birth_off; unsigned long long* done = nullptr; unsigned long long tag = 0; };      // dst = null: no host block (report fetched by memcpy)
// done != null (fgrow_adm_kernel): every workgroup that pushes a part of the report into the pinned host block posts `tag` in its word of
// done[0 .. PUB_DONE_WORDS) (in the same block) behind a system-scope release of what it wrote: the host polls the words instead of waiting
// for an event behind the whole launch (the report is complete when the ILP launch is, not when the grow launch it rides in is)
constexpr int PUB_DONE_WORDS = 9;      // workgroup 0 (head) + FG_PUB_WGS (rows)
constexpr int GROUP_MAX = 32;      // sectors per batched launch
struct FBatch { const FGrowArgs* ga[GROUP_MAX]; const CommitArgs* ca[GROUP_MAX]; FDyn d[GROUP_MAX]; };
struct PBatch { const void* p[GROUP_MAX]; };      // one argument block (in HBM) per sector

// Teams: a cluster of >= TEAM_MIN_K targets that needs a branch and bound is searched by SEVERAL workgroups of the launch -- the one
// it belongs to (member 0) and up to TEAM_W - 1 of the launch's workgroups that have no cluster of their own.  Every member
// replicates the (deterministic) dual phase on its own LDS copy of the cluster; below the root of the search member q descends
// only into the (level-0 column, level-1 column) pairs whose hash is q mod W; the value of the best selection found anywhere is
// shared through one 64-bit atomic-min word; the member that finishes LAST takes the best selection and runs the cluster's
// epilogue (nobody waits for anybody).  At most TEAM_MAX clusters per scan form teams.
#ifndef MHT_TEAM_W
#define MHT_TEAM_W 32
#endif
constexpr int TEAM_MIN_K = 24, TEAM_MAX = 8, TEAM_W = MHT_TEAM_W, TEAM_SEL = 256;
constexpr int XT_WORDS = 4 + TEAM_SEL;      // a device's file in the exchange block of a cluster-sharded step (BlpArgs::shard_team): value key [3], K, child ordinals [K]
struct TeamResult { double ub; int32_t status, nodes, iters, pad; int32_t sel[TEAM_SEL]; };      // what a member found (global column per target)
struct TeamState { unsigned long long gub; int32_t done, pad[13]; };                    // per team: shared incumbent key, finished members, TeamProblem filed
// (A giant cluster -- more columns than the LDS tables hold -- runs its dual phase on HBM scratch: every member has its own copy of that
// scratch and replicates the phase, BlpArgs::tm_sm.)

struct ClusterArgs {
    const unsigned long long* assoc;   // [T][AW]
    int AW;                            // words per target
    const int32_t* nT_dev;
    int Tcap;
    int32_t* edge_t; int32_t* edge_m; int Ecap;
    int elds;                          // edges kept in LDS (set by launch_cluster); the rest spills to edge_t / edge_m
    int pcap;                          // capacity of the LDS list of target pairs to unite (set by launch_cluster)
    int n_mnodes;                      // R * Mpad
    const DevStatus* status;           // forest mode: per-scan status word (overflow => do nothing)
    DevStatus* status_other;           // forest mode: the other parity's status word, cleared here for the scan after this one
    int32_t* dbg;                      // development only: [8] wall-clock ticks at phase boundaries
    int clear_rows;                    // zero the bitset rows while reading them
    const unsigned* edges_in;          // forest mode: deduplicated edge list written by grow_kernel (skips the sweep)
    int32_t* edge_count;               //   [EDGE_SEGS] segment lengths; reset to zero here
    int seg_cap;                       //   segment stride
    int32_t* ticket_reset;             //   grow_kernel's tile ticket, reset to zero here for the next scan
    unsigned* alloc_reset;             //   fgrow_kernel's child counters [FG_REGIONS][32], reset to zero here for the next scan
    int32_t* sel_rel_reset;            //   cluster-sharded step: [Tcap] set to -1 here (every device then fills in what it solved)
    // outputs
    int32_t* t_label;      // [T] smallest member of the component
    int32_t* t_cluster;    // [T] cluster index
    int32_t* cl_ptr;       // [T+1]
    int32_t* cl_members;   // [T]
    int32_t* multi_list;   // [T] cluster indices with >= 2 members
    int32_t* single_list;  // [T] target indices that are alone in their cluster
    int32_t* counts;       // [4]: nClusters, nMulti, nSingle, edge overflow; [5]: clusters in team_list (forest)
    // cluster-sharded step (one tracker on shard_n devices with identical forests): which device solves which multi-target cluster.
    // Longest-processing-time rule on the clusters' column counts (largest first, each to the least loaded device; ties: lower cluster
    // index / lower device) -- every device computes the same table from the same data.
    int shard_n; const int32_t* tchild; const int32_t* tcend; int32_t* cl_owner;      // [T] by cluster index, or null / shard_n <= 1: off
    int32_t* gtab;         // cluster_big_kernel: the tables in HBM (cluster_big_ints() ints), or null
    int32_t* team_list;    // [TEAM_MAX] or null: clusters of >= TEAM_MIN_K targets (searched by teams of workgroups, see BlpArgs)
    TeamState* team_state; // [TEAM_MAX] reset here for this scan
};

// branch and bound of blp_kernel: the first BB_RE_LEVELS levels re-optimise the prices of their residual problem and keep a
// snapshot of them in HBM; BB_SLOTS snapshot sets are shared by the workgroups of a launch (clusters that branch are rare)
constexpr int BB_RE_LEVELS = 32;
constexpr int BB_SLOTS = 8 + TEAM_W;

struct RingLayer { const double* x; const double* cnllr; const int32_t* meas; const uint8_t* flags; };   // one layer of the node ring

struct BlpArgs {
    const int32_t* cl_ptr; const int32_t* cl_members; const int32_t* multi_list; const int32_t* single_list;
    const int32_t* counts;          // [1] = nMulti, [2] = nSingle
    const int32_t* tchild;          // [T] children of target t are columns tchild[t] .. tcend[t]-1
    const int32_t* tcend;           // [T] (the stateless seam passes group_ptr and group_ptr + 1)
    const double* cost;             // [cap] f_h
    const double* cnllr;            // [cap] cumulativeNLLR of the children (single-target clusters)
    const int32_t* path; int cap; int PD;
    int pds;                        // 0: path = [PD][cap] rows (stateless seam); else: one record of pds ints per column, path[h * pds + d],
                                    // entries beyond PD = -1 (forest; apath has the same layout)
    double* u; int32_t* usage; int32_t* mark; int n_mnodes;       // HBM-path scratch, zero on entry and on exit
    // branch and bound with re-optimised prices: snapshot pool [slots][levels][bb_snap_rows] and its busy flags (zero = free);
    // null = static-bound search
    double* bb_snap; int32_t* bb_busy; int bb_snap_rows;
    // per-member scratch, slot = cl_ptr[c] + c + k  (k = 0..K)
    int32_t* best_h; double* best_rc; int32_t* bb_ch; int32_t* bb_best; double* bb_cost; double* bb_uused;
    double* bb_last_rc; int32_t* bb_last_idx; double* bb_rest; double* bb_min;
    int32_t* sel;                   // [T] out: selected child per target
    int32_t* cl_status; int32_t* cl_iters; int32_t* cl_nodes;   // [T] per cluster (indexed by cluster id)
    int32_t* cl_time;               // [T][2] or null: wall-clock ticks (10 ns) spent in setup / in total, per cluster
    int max_iter; int node_limit;
    long long time_limit;           // wall-clock budget of a cluster's branch and bound in 10 ns ticks (0 = none): like the node limit, the best
                                    // feasible selection found so far is returned with MHT_BLP_NODE_LIMIT
    int force_hbm;                  // testing: run every cluster through the HBM storage policy (as oversized clusters do)
    int no_enum;                    // testing: small uncertified clusters go to the branch and bound instead of the exact search
    int no_reduce;                  // testing: giant clusters stay on the HBM policy (no reduced-cost fixing + LDS re-solve)
    int skip_dead;                  // similar-state pruning ran on this scan's children: F_DEAD ones are no hypotheses any more
    // LDS tier of the launch (blp_set_tier): capacities of the LDS-resident solve (columns, rows, targets, bitset words); clusters
    // beyond them run on HBM scratch.  tier 0: one launch; 1: small footprint, skips clusters with more than t1_h columns / t1_k
    // targets; 2: default footprint, takes exactly those
    int cap_h, cap_r, cap_k, cap_uw, tier, t1_h, t1_k;
    const int32_t* team_list; TeamState* team_state; TeamResult* team_res;      // teams (null: off): [TEAM_MAX], [TEAM_MAX], [TEAM_MAX][TEAM_W]
    // HBM scratch of team members: u / usage / mark hold TEAM_W copies tm_sm elements apart, the per-member tables (best_h ... bb_min) TEAM_W
    // copies tm_ss elements apart (copy 0 = the owner's); 0: no copies, a cluster on HBM scratch is its owner's alone
    size_t tm_sm, tm_ss;
    int32_t* big_list; int32_t* big_count;
    int shard_n, shard_i;           // cluster sharding over devices with identical forests (0 / 1: off)
    const int32_t* cl_owner;        // [T] device of every multi-target cluster (cluster kernel: LPT by column count); null: cluster c on device c % shard_n
    int32_t* shard_team;            // [shard_n][TEAM_MAX][XT_WORDS] or null: the clusters of the team list are searched by ALL devices (Team::qg / Wg), each files its best here
    int32_t* sel_rel;               // [T] or null: selected child relative to the target's block (tchild[t]); -1 = not solved here      // clusters tier 1 left for tier 2 (count reset by the cluster kernel: counts[4])
    const DevStatus* status;        // forest mode: per-scan status word (overflow => do nothing)
    // forest epilogue (null for the stateless seam): track termination + N-scan prune decision per target
    // (tracker.py:891-916, pyTarget.py:343-356), evaluated by whoever selected the target's leaf
    const double* x; const uint8_t* flags;        // newest layer (x: [4][cap])
    const double* t_root_cnllr; const uint8_t* t_root_f32; const int32_t* t_depth; const int32_t* t_window;
    int32_t* t_alive; int32_t* t_jdrop; int32_t* t_count; int32_t* t_firstsurv; double* t_score;
    int Nwin; double score_limit, cnllr_limit, radar_x, radar_y, radar_range;
    // N-scan prune per target (pyTarget.pruneDepth, pyTarget.py:343-356), done by whoever selected the target's leaf:
    // new root (ancestor table look-up), the target's report record, the surviving leaf range (first, count)
    const int32_t* apath; int R; int kc;      // kc = ring slot of this scan's layer (scan % R)
    RingLayer ring0; size_t ring_stride;      // layer k of the ring: every array of ring0 advanced by k * ring_stride BYTES
    const int32_t* t_id; const int32_t* t_root_scan; const int32_t* t_root_node; const int32_t* t_label;
    mht_target_report* rec; int32_t* w_root_scan; int32_t* w_root_node; double* w_root_cnllr; uint8_t* w_root_f32;
    // clusters from the grow launch's union-find (uf_epoch != 0: blp_uf_kernel; see FDyn::uf_epoch): the parents, the target count of the
    // table the scan ran on, and what the cluster kernel used to reset for the next scan
    const unsigned long long* uf_parent; unsigned uf_epoch; const int32_t* nT_dev; int uf_cap;      // uf_cap: entries of uf_parent (max_targets)
    DevStatus* status_other; unsigned* alloc_reset; int32_t* t_cluster;
    // publication for an overlapping grow launch of the next scan (null: off): per-target records, written for slots [0, pub_ub); the
    // counter the workgroups count themselves off on when they leave; the scan's tag
    unsigned long long* rec0; int pub_ub; unsigned long long* blp_done; unsigned pub_scan;
    unsigned long long* begun;             // != null: the launch's first workgroup posts pub_scan here at entry (FCounts::ilp_begun)
    unsigned long long* dbg;       // development only (MHT_BLP_STAMPS=1 with MHT_GROW_DEBUG): [32 + workgroup * 16 + k] wall-clock ticks of blp_uf_kernel's phases
    const unsigned long long* ni_flag; int uf_ovl;      // uf_ovl: the scan's grow launch overlapped the previous ILP launch -- if ni_flag says that a
                                                        // target died in the previous scan, the union-find was redone under epoch | 1
    unsigned uf_lds_off;           // offset of the workgroup's UfPersist block in the dynamic LDS (behind the solver's tables; set by launch_blp)
};

// prune_similar_kernel (mht_similar.hip): similar-state pruning of the targets that are alone in their cluster, between the
// cluster kernel and the ILP kernel of a scan
struct SimilarArgs {
    const int32_t* single_list; const int32_t* counts;      // cluster kernel: targets alone in their cluster, counts[2] of them
    const int32_t* tchild; const int32_t* tcend;            // children of target t
    double* x; double* cnllr; const double* pd; const int32_t* meas; int32_t* cov; uint8_t* flags; double* cost; int cap;   // this scan's layer
    const double* t_root_cnllr; const uint8_t* t_root_f32; int Nwin;
    VTab vt; Model model;
    float thr;                                              // Tracker.pruneThreshold (tracker.py:117), compared in float32
    const DevStatus* status;
    const int32_t* mmsi;                                    // AIS forest: children updated with an AIS message are not merged (pyTarget.py:371-375), else null
    int32_t* t_window; const int32_t* t_depth;              // AIS forest (else null): the table the scan ran on -- the lone targets' association sets are rebuilt from their trees (WIN_REBUILT_*)
    // constant-turn forest (else null): the covariances of this scan's children by PARENT node -- P_hat of every hit child, P_bar of the missed-detection
    // child (CtGrow: key = 2 x parent + hit/miss).  The merged node takes the missed-detection child's slot AND its key: a mean that is not the hit
    // children's covariance itself is written over the parent's P_bar entry (the missed-detection hypothesis it belonged to is gone)
    const float* ct_Phat; float* ct_Pbar;
};

// forest_ais_kernel (mht_ais.hip): the fused children of every leaf of the newest layer, in front of the grow launch of a scan with AIS messages
struct AisGroup; struct AisMsg;
struct AisForestArgs {
    Model model;
    const int32_t* nT_dev; const int32_t* t_first; const int32_t* t_leaf_off;      // the committed target table
    const double* x; const double* pd; const int32_t* cov; const uint8_t* flags; const int32_t* hmmsi; int cap;      // newest layer
    VTab vt;
    const AisGroup* groups; int nG; const AisMsg* msgs;
    double eta2_ais, lambda_ais;
    const float* z; int M;
    int32_t* nf; int32_t* off; AisRec* rec; int rec_cap; unsigned* rec_count;      // rec_count: zero at launch
    DevStatus* status;
};
int launch_forest_ais(mht_ctx* ctx, const AisForestArgs& a, int n_targets_ub);
int launch_gate(mht_ctx* ctx, GateArgs& a, int grid_leaves_hint);
int launch_prune_similar(mht_ctx* ctx, const SimilarArgs& a, int n_targets_ub);
struct AddArgs;      // mht_admit.h
int launch_fgrow(mht_ctx* ctx, const FGrowArgs& a, FDyn& d, int n_targets_ub, const CommitArgs* commit, const PublishArgs* publish = nullptr,
                 const AddArgs* adm = nullptr, bool any_order = false);
int forest_sync_side(mht_ctx* ctx);      // mht_forest.hip: waits for what the forest queued on streams of its own
size_t fgrow_lds_bytes(int W, int pds, int AW);
size_t fgrow_lds_bytes_cap(int W, int pds, int AW, int cap);
void fgrow_plan(FDyn& d, int n_targets_ub, int Tcap, bool fused, bool wave);
size_t fgrow_wave_lds_bytes(int W, int pds, int AW);
int fgrow_grid_of(const FDyn& d);
int launch_fgrow_batch(mht_ctx* ctx, const FBatch& b, int n_sectors, int grid_x, size_t lds, int pds, bool wave);
int launch_cluster_batch(mht_ctx* ctx, const PBatch& av, int n_sectors, int Tcap, int n_mnodes);
int launch_blp_batch(mht_ctx* ctx, const PBatch& av, int n_sectors, int grid_x, size_t lds);
size_t blp_set_tier(BlpArgs& a, int tier);
int launch_blp_light_batch(mht_ctx* ctx, const PBatch& av, int n_sectors, int grid_x);
void fill_model(GateArgs& a, const mht_model* m);
void fill_model_only(Model& o, const mht_model* m);
struct InitArgs;
// init != null: a second workgroup of the launch runs the M-of-N initiator on this scan's used-measurement bytes (streaming API path)
int launch_cluster(mht_ctx* ctx, const ClusterArgs& a, const InitArgs* init = nullptr, const int32_t* sticky_overflow = nullptr);
void cluster_prepare(ClusterArgs& a);
size_t cluster_lds_bytes(int Tcap, int n_mnodes);
int cluster_elds(int Tcap, int n_mnodes);
bool cluster_fits_lds(int Tcap, int n_mnodes);
size_t cluster_big_ints(int Tcap, int n_mnodes);
int launch_blp(mht_ctx* ctx, const BlpArgs& a, int grid);
bool blp_uf_fits(int Tcap, int n_mnodes);
int launch_blp_epilogue(mht_ctx* ctx, const BlpArgs& a, const int32_t* nT_dev, int n_targets_ub);
int launch_shard_team_resolve(mht_ctx* ctx, const BlpArgs& a, int shard_n);
void forest_destroy(mht_ctx* ctx);

}  // namespace mht
