/* mht_amd.h -- C ABI of libmht_amd.so: the MI355X (gfx950) implementation of pyMHT's per-scan hot path.
 *
 * The reference (erikliland/pyMHT) has NO native/FFI boundary: the path sits behind Python methods of
 * pymht/tracker.py.  The entry points below are what a ctypes binding on the reference side would call in
 * place of those methods (INTEGRATION.md shows the stub); each one cites the reference code it replaces
 * (file:line relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / HIP types in the signatures.  `stream` arguments are
 *     a hipStream_t passed as void* (NULL = the default stream).
 *   - "dev" pointers are device (HBM) pointers, e.g. torch.Tensor.data_ptr(); "host" pointers are ordinary
 *     host memory.  The caller owns every buffer it passes; the library owns only its ctx and workspace.
 *   - every function returns MHT_OK (0) or a negative MHT_E_* code and never aborts; mht_last_error() gives
 *     the message of the last failure on the calling thread.  (The reference signals failure with `assert`,
 *     e.g. tracker.py:1212; the Python wrapper turns non-zero codes into AssertionError/RuntimeError.)
 *   - one ctx per Tracker, one HIP stream per ctx, calls on a ctx are serialised by the caller (the reference
 *     is single-threaded and non-reentrant); different ctxs may be used from different threads / GPUs.
 */
#ifndef MHT_AMD_H
#define MHT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHT_ABI_VERSION 6

/* State dimension of the library build the header is used with: 4 (libmht_amd.so: the reference's CV model, models/pv.py) or 6
 * (libmht_amd6.so: the same sources compiled with -DMHT_NX=6 for BASELINE config 5's six-state model).  It sizes the model matrices
 * and the state vectors / covariances of the forest's reports; the stateless seams mht_gate_scan (4 states) and mht_gate_scan_x
 * (4 or 6 at run time) do not depend on it. */
#ifndef MHT_NX
#define MHT_NX 4
#endif

enum {
    MHT_OK = 0,
    MHT_E_INVALID = -1,    /* bad argument */
    MHT_E_HIP = -2,        /* a HIP runtime call failed */
    MHT_E_CAPACITY = -3,   /* an output buffer / pool is too small: grow and retry */
    MHT_E_INFEASIBLE = -4, /* 0-1 ILP without a feasible point (cannot happen for MHT trees) */
    MHT_E_LIMIT = -5,      /* branch-and-bound node limit hit: selection returned is feasible, not proven optimal */
    MHT_E_STATE = -6       /* call sequence error */
};

/* node flag bits (mht_nodes.flags) -- dtype bookkeeping of the reference */
#define MHT_F_STATE_F32 1u /* state chain is float32 (targets born from the initiator, m_of_n.py:353-358) */
#define MHT_F_SCORE_F32 2u /* cumulativeNLLR currently holds a float32 value */
#define MHT_F_DEAD 8u      /* forest only: taken out of the tree by similar-state pruning (never set in what mht_forest_leaves returns) */
#define MHT_F_COV_F64 16u  /* forest with MHT_FOREST_AIS only: the node's covariance is float64 -- an AIS-updated node (models/ais.py:4:
                            * ais.C is float64, so P_hat of tracker.py:451-487 is) or a child of a batch NumPy promoted because one of its
                            * members was (np.array of the leaves' P_0, tracker.py:859-870).  Such a node's state is float64 too. */

/* ILP status (mht_solve_blp / forest step) */
#define MHT_BLP_CERTIFIED 1   /* Lagrangian certificate: conflict-free minimisers + complementary slackness */
#define MHT_BLP_BRANCHED 2    /* proven optimal by branch and bound on the GPU */
#define MHT_BLP_NODE_LIMIT 3  /* node limit: best feasible point returned (MHT_E_LIMIT) */

typedef struct mht_ctx mht_ctx;

/* The linear-Gaussian model: what Tracker.__init__ reads off the model module (tracker.py:54-59, pv.py:7-34)
 * plus the two scalars the gate and the score need (tracker.py:107, :110).  Row-major float32. */
typedef struct mht_model {
    float A[MHT_NX * MHT_NX]; /* Phi(radarPeriod) */
    float Q[MHT_NX * MHT_NX]; /* Q(radarPeriod)   */
    float C[2 * MHT_NX];      /* C_RADAR (2 x nx) */
    float R[4];               /* R_RADAR() (2x2)  */
    double eta2;
    double lambda_ex;
    double default_pd;        /* Tracker.default_P_d; nodes whose pd equals it use default_miss_nllr */
    double default_miss_nllr; /* -log(1 - default_pd) evaluated by the host libm (pyTarget.py:326) */
} mht_model;

/* A layer of track hypotheses in HBM, structure of arrays (one hypothesis = one index i < cap).
 * Field meaning follows pyTarget.Target (pyTarget.py:16-40). */
typedef struct mht_nodes {
    double* x;       /* dev [4][cap]  : x[k*cap+i]  state x_0 (holds f32 values when MHT_F_STATE_F32) */
    double* cnllr;   /* dev [cap]     : cumulativeNLLR */
    double* pd;      /* dev [cap]     : P_d */
    int32_t* parent; /* dev [cap]     : index of the parent in the previous layer, -1 for a fresh root */
    int32_t* meas;   /* dev [cap]     : measurementNumber (1-based), 0 = missed detection */
    int32_t* cov;    /* dev [cap]     : column of P holding P_0 of this node */
    uint8_t* flags;  /* dev [cap]     : MHT_F_* */
    float* P;        /* dev [16][cap_cov] : P[e*cap_cov+j], e = 4*row+col.  Children of one parent share
                        two columns: 2*leaf = P_bar (miss child), 2*leaf+1 = P_hat (all hit children),
                        as the reference shares one ndarray between siblings (pyTarget.py:246) */
    int32_t cap, cap_cov;
} mht_nodes;

/* ---- lifetime ------------------------------------------------------------------------------------------- */
int mht_abi_version(void);
const char* mht_last_error(void);
/* device: HIP device ordinal; stream: hipStream_t as void* (NULL = default stream) */
int mht_create(mht_ctx** out, int device, void* stream);
int mht_destroy(mht_ctx* ctx);
int mht_synchronize(mht_ctx* ctx);

/* ---- seam (i): Tracker._processLeafNodes + Target.spawnNewNodes ------------------------------------------
 * Replaces tracker.py:383-398 (-> :861-889 predict/precalc, :804-859 gate/update/score) and the child
 * construction of pyTarget.py:227-258 / :319-328 for ALL leaves of ALL targets in one call.
 *
 * For leaf i (0 <= i < L, node leaf_src[i] of `in`, or node i if leaf_src == NULL) the children are written
 * to `out` at indices child_ptr[i] .. child_ptr[i+1]-1: first the missed-detection child (x_bar, P_bar,
 * cnllr - log(1-pd)), then one child per gated measurement in ascending measurement index (x_hat, P_hat,
 * cnllr + nllr, meas = index+1) -- the order of np.nonzero (tracker.py:832) and of getLeafNodes (pyTarget.py:461).
 *   z          dev (M,2) float32 row-major, the scan (MeasurementList.measurements)
 *   child_ptr  dev [L+1] int32 (output)
 *   nllr       dev [out->cap] double or NULL: per child score increment (kalman.py:14-22 / -log(1-pd))
 *   used       dev [(M+63)/64] uint64 or NULL: bit j set iff some leaf gated measurement j (tracker.py:331);
 *              OR-ed into, the caller zeroes it
 *   n_children host int* or NULL: total number of children (forces a stream synchronisation when non-NULL)
 * Returns MHT_E_CAPACITY (after synchronising) if out->cap / out->cap_cov are too small when n_children is
 * requested; otherwise capacity overflow is reported by the next synchronising call. */
int mht_gate_scan(mht_ctx* ctx, const mht_model* model, const mht_nodes* in, const int32_t* leaf_src, int32_t L,
                  const float* z, int32_t M, const mht_nodes* out, int32_t* child_ptr, double* nllr,
                  uint64_t* used, int32_t* n_children);

/* ---- seam (i), dimension-generic: a linear-Gaussian model with nx states (4 or 6) and 2 measurements -----------------------
 * BASELINE config 5 names a 6-state model; the reference ships none, but its kalman module is dimension-generic:
 * predict / precalc (pymht/utils/kalman.py:55-101), z_tilde / NIS / gate (kalman.py:25-40, tracker.py:829), numpyFilter
 * (kalman.py:43-52), nllr (kalman.py:14-22).  This is that module for L leaves x M measurements in one call, results in the
 * reference's evaluation order.  All arrays are device memory, structure-of-arrays:
 *   x [nx][L] f64, flags [L] (MHT_F_STATE_F32: the leaf's state chain is float32), P [nx*nx][L] f32, pd [L], z (M,2) f32
 *   x_bar [nx][L] f64, P_bar / P_hat [nx*nx][L], S / S_inv [4][L], K [2*nx][L]  (K row-major nx x 2)
 *   row_ptr [L+1], col_idx [cap] (gated measurement indices, ascending per leaf: np.nonzero, tracker.py:832),
 *   x_hat [nx][cap] f64 and nllr [cap] per gated pair.  n_pairs (host, may be NULL) = row_ptr[L].  Synchronises.
 * Returns MHT_E_CAPACITY if there are more gated pairs than cap. */
typedef struct mht_model_x {
    int32_t nx;             /* 4 or 6 */
    const float* A;         /* host [nx*nx] row-major state transition */
    const float* Q;         /* host [nx*nx] process noise */
    const float* C;         /* host [2*nx] measurement matrix */
    const float* R;         /* host [4] measurement noise */
    double eta2, lambda_ex;
    /* transition = 0: the linear model above (A for every leaf: kalman.predict, kalman.py:55-64).
     * transition = 1 (nx = 6 only; BASELINE config 5's constant-turn model, pymht_amd/models/ct.py): state = [x, y, vx, vy, w, a]; every
     * leaf gets its OWN A = Phi(period, w of the leaf) -- velocity rotated by w * period, the arc integrated, w += period * a; A above is
     * ignored -- and runs the reference's per-hypothesis form: kalman.predict_single (kalman.py:67-70), then kalman.precalc on a batch of
     * one (kalman.py:82-101), i.e. the matrix x vector products in BLAS gemv order. */
    int32_t transition;
    double period;
} mht_model_x;
int mht_gate_scan_x(mht_ctx* ctx, const mht_model_x* model, int32_t L, const double* x, const uint8_t* flags, const float* P,
                    const double* pd, const float* z, int32_t M, double* x_bar, float* P_bar, float* P_hat, float* S, float* S_inv,
                    float* K, int32_t* row_ptr, int32_t* col_idx, double* x_hat, double* nllr, int32_t cap, int32_t* n_pairs);


/* ---- seam (ii): Tracker._findClustersFromSets (tracker.py:961-974) -------------------------------------------
 * assoc  dev [T][words] uint64: bit b of row t set iff target t is associated with measurement node b
 *        (the reference's __associatedMeasurements__ sets; a node is a (scan, measurement) of the window)
 * label  dev [T] int32 out: smallest target index of the connected component of t.  Clusters ordered by label
 *        with ascending members are exactly the reference's cluster list. */
int mht_cluster(mht_ctx* ctx, int32_t T, int32_t words, const uint64_t* assoc, int32_t* label);

/* ---- seam (iii): Tracker._solveBLP_OR_TOOLS(A1, A2, f) (tracker.py:1155-1217) ---------------------------------
 * One 0-1 ILP:  min f.tau  s.t.  A1 tau <= 1, A2 tau = 1, tau binary, in the sparse form the tree gives it:
 *   group_ptr dev [nT+1] int32 : columns group_ptr[t] .. group_ptr[t+1]-1 belong to target t (A2, tracker.py:1115)
 *   rows      dev [depth][nHyp] int32 : rows[d*nHyp+h] = d-th measurement row of column h or -1 (A1 by columns)
 *   cost      dev [nHyp] double  (f = getScore()/N, tracker.py:1124-1136)
 *   selected  dev [nT] int32 out : chosen column per target (ascending = the reference's return list)
 *   objective/status/iterations/nodes : host out (status MHT_BLP_*).  Synchronises the stream. */
int mht_solve_blp(mht_ctx* ctx, int32_t nHyp, int32_t nT, int32_t nRows, int32_t depth, const int32_t* group_ptr,
                  const int32_t* rows, const double* cost, int32_t max_iter, int32_t node_limit, int32_t* selected,
                  double* objective, int32_t* status, int32_t* iterations, int32_t* nodes);

/* ---- seam (iv): Tracker._nScanPruning (tracker.py:1229-1231) -> _pruneTargetIndex (:1219-1227) -> Target.pruneDepth
 * (pyTarget.py:343-356) -> _pruneAllHypothesisExceptThis(backtrack=True) (:330-337), for trees the caller owns ---------------
 * The trees of all targets are one array of parent pointers.
 *   parent    dev [n_nodes] int32 : parent node or -1 (top of a tree)
 *   sel       dev [T] int32       : __trackNodes__[t], the selected leaf of target t
 *   window    dev [T] int32       : __targetWindowSize__[t] (N)
 *   new_root  dev [T] int32 out   : the ancestor `window` levels above sel[t] (the top of the tree if it is closer)
 *   keep      dev [n_nodes] uint8 out : 1 iff the node survives (it is a new root, above one, or below one)
 * Asynchronous on the ctx stream. */
int mht_prune(mht_ctx* ctx, int32_t n_nodes, const int32_t* parent, int32_t T, const int32_t* sel, const int32_t* window,
              int32_t* new_root, uint8_t* keep);

/* ---- AIS-aided children: Tracker.__fuseRadarAndAis (tracker.py:417-552), stateless ------------------------------------------
 * Per leaf and per AIS message (a 4-state report [x, y, vx, vy] of a ship with identity mmsi, made inside the radar period in
 * front of the scan; models/ais.py) that gates with it (eta2_ais, tracker.py:111): the leaf is predicted to the message's time,
 * updated with it, predicted on to the scan's time and gated against the radar measurements; one child per gated radar
 * measurement, score (nllr_ais + nllr_radar) / 2, or ONE child without a radar measurement, score nllr_ais (tracker.py:497-526).
 * The messages come grouped as the reference walks them (tracker.py:447-453: message times in the iteration order of their SET,
 * high accuracy before low, list order inside a group; pymht_amd/ais.py::group_messages builds the arrays):
 *   groups  host [nG]: Phi, Q (models/pv.py:12-24, float32) over dT1 = t_message - t_leaves and dT2 = t_scan - t_message,
 *           sigma^2 of the accuracy class (models/ais.py:9-13), first message and count
 *   msgs    host [nA]: state float64[4], mmsi
 * Leaves as in mht_gate_scan_x (x dev [4][L] float64, flags, P dev [L][16] float32, pd), own dev [L] int32 or null: the identity
 * a leaf's track is bound to, 0 = none -- messages of other ships are skipped (pyTarget.py:269-272).  model: C, R, eta2,
 * lambda_ex are read.  lambda_ais = nTargets P_ais / (pi radarRange^2) (tracker.py:438; needs a finite radarRange).
 * Out, CSR by leaf in the reference's order: child_ptr dev [L+1]; out_x dev [4][cap] float64; out_P dev [cap][16] float64 (the
 * reference's fused covariances are float64: ais.C is); out_radar dev [cap] (0-based radar measurement or -1); out_nllr;
 * out_msg dev [cap] index into msgs.  Synchronous; MHT_E_CAPACITY if cap is too small (*n_children = what is needed).
 * 4-state build only. */
typedef struct mht_ais_group {
    float A1[16], Q1[16], A2[16], Q2[16];
    float r_diag;
    int32_t first, count, pad;
} mht_ais_group;
typedef struct mht_ais_msg {
    double state[4];
    int32_t mmsi;
    int32_t pad;
} mht_ais_msg;
int mht_fuse_ais(mht_ctx* ctx, const mht_model* model, int32_t L, const double* x, const uint8_t* flags, const float* P, const double* pd,
                 const int32_t* own, const mht_ais_group* groups, int32_t nG, const mht_ais_msg* msgs, int32_t nA, double eta2_ais,
                 double lambda_ais, const float* z, int32_t M, int32_t* child_ptr, double* out_x, double* out_P, int32_t* out_radar,
                 double* out_nllr, int32_t* out_msg, int32_t cap, int32_t* n_children);

/* The same for leaves whose covariance the reference carries in float64 (ABI 5): P dev [L][16] float64 -- the leaves with MHT_F_COV_F64 in
 * `flags` are fused from it as it is (kalman.predict_single on a float64 node.P_0, tracker.py:449-450: nodes behind an AIS update and their
 * targets' later leaves), the others from its values rounded to float32 (exact when they ARE float32 values). */
int mht_fuse_ais_f64(mht_ctx* ctx, const mht_model* model, int32_t L, const double* x, const uint8_t* flags, const double* P, const double* pd,
                     const int32_t* own, const mht_ais_group* groups, int32_t nG, const mht_ais_msg* msgs, int32_t nA, double eta2_ais,
                     double lambda_ais, const float* z, int32_t M, int32_t* child_ptr, double* out_x, double* out_P, int32_t* out_radar,
                     double* out_nllr, int32_t* out_msg, int32_t cap, int32_t* n_children);

/* ---- the device-resident hypothesis forest: Tracker.addMeasurementList end to end -----------------------------
 * Replaces steps 1-6 of tracker.py:162-307 (grow :207-209, cluster :220, optimise :228-236, terminate :252-253,
 * N-scan prune :258 = seam (iv) Tracker._nScanPruning, tracker.py:1219-1231 / pyTarget.py:343-356) without a host
 * round trip between the stages.  The tree store of pyTarget.Target objects becomes a ring of mht_nodes layers
 * (one per scan of the window) owned by the ctx. */
typedef struct mht_forest_config {
    int32_t max_targets;  /* capacity of the target list */
    int32_t max_nodes;    /* hypotheses per scan layer (children of one scan + roots born in it) */
    int32_t max_meas;     /* measurements per scan (<= 4096; (n_scan + 4) x max_meas rounded up to 64 <= 65536) */
    int32_t n_scan;       /* Tracker.N: N-scan window (tracker.py:112-114) */
    int32_t blp_max_iter; /* dual-ascent steps before branch and bound (200 when < 0; 0 = branch and bound only) */
    int32_t blp_node_limit; /* branch-and-bound node budget per cluster (default 1<<20 when <= 0) */
    double score_limit;   /* Tracker.scoreUpperLimit  (tracker.py:115) */
    double cnllr_limit;   /* Tracker.clnnrUpperLimit  (tracker.py:116) */
    double radar_x, radar_y, radar_range; /* Tracker.position / radarRange (tracker.py:44-45), range may be +inf */
    double merge_threshold; /* Tracker.mergeThreshold (tracker.py:65) used by initiateTarget */
} mht_forest_config;

/* ---- AIS-aided forest (Tracker.addMeasurementList(scanList, aisList), tracker.py:162, :394-396, :417-552) ---------------------
 * mht_forest_create_ex(..., MHT_FOREST_AIS): a forest whose nodes also carry the identity of the AIS message they were updated
 * with and the identity their track is bound to (pyTarget.py:34, :297-302) and whose ILP rows include the AIS messages
 * (tracker.py:1057-1064, :1083-1090).  4-state build, n_scan <= 7 (n_scan <= 3: 8-entry path records, the ILPs stay in LDS;
 * above: 16-entry records, the ILPs run on the HBM policy).  mht_forest_set_ais hands over the messages of the NEXT scan, grouped as
 * for mht_fuse_ais; the next mht_forest_step / _step_host / _scan consumes them: radar M + nA <= max_meas (rounded up to a multiple
 * of 64).  A scan without messages needs no call.  Messages start tracks through the initiator (mht_initiator_set_ais).
 * Not available to members of a group.  The cluster-sharded step takes the messages (ABI 5): every shard makes the fused children itself.
 * mht_forest_read_mmsi: identities of the nodes [first, first + count) of the layer of `scan` (host arrays out, either may be null):
 * mmsi[i] = the message node first + i was updated with (0: none; with measurement number 0 that is a child WITHOUT a radar
 * measurement, the reference's measurementNumber None), hist[i] = Target._getHistoricalMmsi(). */
#define MHT_FOREST_AIS 1u
/* mht_forest_create_ex(..., MHT_FOREST_CT), six-state build (libmht_amd6.so) only: the constant-turn model BASELINE config 5 names
 * (pymht_amd/models/ct.py; state [x, y, vx, vy, w, a]).  The transition is not model->A but Phi(T, w) rebuilt for every hypothesis from its
 * own turn rate x[4], T = model->A[4][5] (give A = Phi(T, 0)); every leaf runs the reference's per-hypothesis form kalman.predict_single +
 * kalman.precalc on a batch of one (kalman.py:67-70, :82-101) -- what mht_gate_scan_x does with mht_model_x.transition = 1 -- and nothing
 * is shared by value: the forest keeps the children's covariances per node.  No device initiator, no similar-state pruning, no groups. */
#define MHT_FOREST_CT 2u
int mht_forest_create_ex(mht_ctx* ctx, const mht_model* model, const struct mht_forest_config* cfg, uint32_t flags);
int mht_forest_set_ais(mht_ctx* ctx, const mht_ais_group* groups, int32_t nG, const mht_ais_msg* msgs, int32_t nA, double eta2_ais, double lambda_ais);
int mht_forest_read_mmsi(mht_ctx* ctx, int32_t scan, int32_t first, int32_t count, int32_t* mmsi, int32_t* hist);
/* the same for n given nodes of the layer (host array `nodes`; a node outside the layer gives 0): a gather on the device, for callers that
 * need a few scattered nodes -- e.g. the roots that join a track's committed history -- and not a whole layer (ABI 5) */
int mht_forest_read_mmsi_nodes(mht_ctx* ctx, int32_t scan, int32_t n, const int32_t* nodes, int32_t* mmsi, int32_t* hist);

/* per-target record of the scan report (old target-list order) */
typedef struct mht_target_report {
    int32_t id;         /* Target.ID */
    int32_t status;     /* 0 alive, 1 out of range, 2 score too high, 3 cNLLR too high (tracker.py:891-916) */
    int32_t sel_node;   /* index of the selected leaf (__trackNodes__[t]) in the layer of this scan */
    int32_t sel_meas;   /* its measurementNumber */
    int32_t new_index;  /* index in the target list after termination, -1 if terminated */
    int32_t root_scan;  /* scanNumber of the root after N-scan pruning */
    int32_t root_node;  /* node index of that root in its layer */
    int32_t n_leaves;   /* leaves kept for the next scan */
    double sel_x[MHT_NX];    /* state of the selected leaf */
    double sel_cnllr;   /* its cumulativeNLLR */
    double score;       /* getScore() = cNLLR - root.cNLLR before pruning (pyTarget.py:124) */
    double root_cnllr;  /* cumulativeNLLR of the root after pruning */
    double root_x[MHT_NX];   /* state of the root after pruning */
    int32_t root_meas;  /* measurementNumber of the root after pruning */
    int32_t cluster;    /* smallest target index of this target's cluster (tracker.py:961-974) */
} mht_target_report;

/* one target the device initiator gave birth to after the scan (mht_forest_initiate) */
typedef struct mht_birth_report {
    int32_t id;          /* Target.ID, or -1 if Tracker.initiateTarget discarded the candidate (too close to a current track) */
    int32_t meas;        /* measurementNumber: 1-based index among the scan's UNUSED measurements, 0 for a merged target */
    double x0[MHT_NX];   /* float32 values */
    float P0[MHT_NX * MHT_NX];
} mht_birth_report;

typedef struct mht_scan_report {
    int32_t scan;          /* scanNumber just processed */
    int32_t n_targets;     /* targets before termination (= number of records) */
    int32_t n_alive;
    int32_t n_leaves_in;   /* L: leaves gated in this scan */
    int32_t n_children;    /* L + G */
    int32_t n_leaves_out;  /* leaves kept for the next scan */
    int32_t n_clusters, n_ilp; /* clusters, clusters with >= 2 targets (Tracker.nOptimSolved) */
    int32_t n_branched;    /* ILPs that needed branch and bound */
    int32_t n_limit;       /* ILPs that hit the node limit (selection feasible, not proven optimal) */
    int32_t blp_iters_max;
    int32_t error;         /* 0 or MHT_E_CAPACITY if a pool overflowed during the scan */
    int32_t used_words;    /* number of valid words in `used` */
    int32_t n_births;      /* candidates of the device initiator after this scan (0 without mht_forest_initiate) */
    int32_t pad[2];
    /* device time of the scan's stages in 10 ns ticks of the GPU's wall clock, stamped by the kernels themselves (no HIP events, no
     * host cost): t_process = grow launch (tracker.py toc['Process']), t_cluster (toc['Cluster']), t_optim = similar-state pruning +
     * ILPs + termination / N-scan prune decisions (toc['Optim']), t_scan = start of the grow launch .. end of the last ILP workgroup */
    int32_t t_process, t_cluster, t_optim, t_scan;
    const uint64_t* used;              /* host: bit j set iff measurement j was gated by some leaf */
    const mht_target_report* targets;  /* host: n_targets records */
    const mht_birth_report* births;    /* host: n_births records */
} mht_scan_report;

int mht_forest_create(mht_ctx* ctx, const mht_model* model, const mht_forest_config* cfg);
/* Tracker.initiateTarget (tracker.py:147-160) for n candidates in order: a candidate closer than merge_threshold
 * to any current leaf (or to an earlier accepted candidate) is discarded when check_neighbours != 0.
 * x0 host [n][4] double, P0 host [n][16] float, flags host [n] uint8 (MHT_F_*), pd host [n] double,
 * meas host [n] int32 (measurementNumber of the new root), accepted host [n] uint8 out (may be NULL),
 * ids host [n] int32 out (assigned Target.ID or -1; may be NULL).  Synchronises when an output is requested. */
int mht_forest_add_targets(mht_ctx* ctx, int32_t n, const double* x0, const float* P0, const uint8_t* flags,
                           const double* pd, const int32_t* meas, int32_t check_neighbours, uint8_t* accepted,
                           int32_t* ids);
/* Same with every array in device memory (dev pointers, accepted/ids may be NULL); fully asynchronous -- used to
 * replay pre-staged births without a host round trip. */
int mht_forest_add_targets_dev(mht_ctx* ctx, int32_t n, const double* x0, const float* P0, const uint8_t* flags,
                               const double* pd, const int32_t* meas, int32_t check_neighbours, uint8_t* accepted,
                               int32_t* ids);
/* One scan, asynchronous: z dev (M,2) float32.  Two launches (grow with the clustering union-find, ILP + prune decisions; three with the
 * clustering kernel on similar-state pruning scans); the target-side commit
 * of the scan (compacted target table, next leaf ranges, the report) is deferred: it rides in the next step's first launch,
 * or runs as a launch of its own as soon as the report, new targets or an export are asked for.  Either order leaves the same
 * forest (tests/test_forest_edge_gpu.py).  */
int mht_forest_step(mht_ctx* ctx, const float* z, int32_t M);
/* Same with z in host memory (copied through a ring of pinned staging buffers of the ctx: asynchronous). */
int mht_forest_step_host(mht_ctx* ctx, const float* z_host, int32_t M);
/* Start the transfer of the last step's report into pinned host memory (runs the scan's commit first if it is still pending) and
 * return at once.  A host that issues the next step before it calls mht_forest_report overlaps its own work with the device's;
 * two transfers can be in flight.  Behind mht_forest_scan (whose report waits for a ride in the NEXT scan's grow launch) it sends
 * that report on its way now, in a launch of its own: what a host does that has no further scan to queue. */
int mht_forest_report_begin(mht_ctx* ctx);
/* Wait for the last step (or for the transfer mht_forest_report_begin started) and expose its report (pointers stay valid until
 * the next but one mht_forest_report_begin on this ctx). */
int mht_forest_report(mht_ctx* ctx, mht_scan_report* out);
/* The report whose transfer the last (which = 0) or the last but one (which = 1) mht_forest_report_begin started.
 * which = 2 (streaming with mht_forest_scan and an initiator only): the report of the scan TWO before the last one, while the last
 * scan's own report has not left the device yet -- a host that queues scan k before it folds the report of scan k - 2 never waits for
 * the device unless it is two scans ahead (MHT_E_STATE when that report is not in a host block any more). */
int mht_forest_report_get(mht_ctx* ctx, int32_t which, mht_scan_report* out);
/* Snapshot of the current leaves in target-list / DFS order (pyTarget.getLeafNodes order): any pointer may be
 * NULL.  x host [n][4], P host [n][16], cnllr host [n], meas/target/id/node host [n] int32, flags host [n] uint8.
 * capacity = length of the host arrays; *n_out = number of leaves.  Synchronises. */
int mht_forest_leaves(mht_ctx* ctx, int32_t capacity, double* x, float* P, double* cnllr, int32_t* meas,
                      int32_t* target, int32_t* id, int32_t* node, uint8_t* flags, int32_t* n_out);
/* The same with every covariance as float64, P host [n][16] double: the exact value of a float32 covariance, the reference's own float64
 * one where flags carries MHT_F_COV_F64 (a forest with AIS: Target.P_0 keeps the dtype the reference gives it, pyTarget.py:16-40).
 * mht_forest_leaves rounds those to float32. */
int mht_forest_leaves_f64(mht_ctx* ctx, int32_t capacity, double* x, double* P, double* cnllr, int32_t* meas,
                          int32_t* target, int32_t* id, int32_t* node, uint8_t* flags, int32_t* n_out);
/* Similar-state pruning -- Tracker._pruneSimilarState (tracker.py:1233-1239) -> Target.pruneSimilarState (pyTarget.py:358-412),
 * what addMeasurementList(..., pruneSimilar=True) asks for (tracker.py:230-231) -- for the scans stepped from now on: in every target
 * that is alone in its cluster, the hit children of a node that lie within `threshold` metres (Tracker.pruneThreshold,
 * tracker.py:117) of its missed-detection child are replaced, together with that child, by one measurement-less hypothesis carrying
 * their mean state / covariance / cumulativeNLLR (NumPy's float32 / float64 arithmetic and summation order).  One extra launch
 * per scan between clustering and the ILPs (also for a member of a group stepped by mht_group_step).  threshold <= 0 switches it off.
 * With it on, n_leaves / n_leaves_out of the report count the slots of the surviving leaf ranges (emptied ones included);
 * n_leaves_in and mht_forest_leaves count hypotheses. */
int mht_forest_set_prune_similar(mht_ctx* ctx, double threshold);
/* Real-time guard for the global-hypothesis ILPs (_solveBLP_OR_TOOLS, tracker.py:1124-1217, has none: a giant cluster without
 * a dual certificate keeps CBC -- and this library's branch and bound -- busy for as long as it takes): a cluster whose
 * branch and bound has run for `milliseconds` of wall-clock time stops like one that reached blp_node_limit -- the best feasible
 * selection found so far is used, the scan's report counts it in n_limit and the report call returns MHT_E_LIMIT.  0 = no limit
 * (default).  For forests stepped by mht_group_step: set it before mht_group_create. */
int mht_forest_set_blp_time_limit(mht_ctx* ctx, double milliseconds);
/* Per-stage device time in milliseconds, SUMMED over the steps issued since the last call (at most 64 may be
 * pending): [0] grow kernel = the reference's toc['Process'], [1] cluster, [2] optimise (ILP + single-target selection,
 * incl. the per-target termination test / prune decision / surviving leaf ranges), [3] commit (target-table compaction,
 * next leaf ranges, report), [4] whole step.  *n_steps = number of steps summed.
 * Timing is off by default (five hipEventRecord per step); enable != 0 switches it on for subsequent steps.
 * Synchronises the stream. */
int mht_forest_set_timing(mht_ctx* ctx, int32_t enable);
int mht_forest_stage_times(mht_ctx* ctx, float* ms5, int32_t* n_steps);
/* Tooling: copy a named internal per-cluster array of the last step to the host ("cl_status", "cl_iters",
 * "cl_nodes", "cl_time" [2 int32 per cluster: setup / total in 10 ns ticks], "cl_ptr", "cl_members", "multi_list",
 * "cl_counts", "cl_owner", "team_list", "tchild").  Synchronises. */
int mht_forest_debug_read(mht_ctx* ctx, const char* name, void* host, int64_t bytes);
/* Ancestor chain of one node: walks parents from (scan, node) towards the root of time, at most max_len steps
 * (bounded by the ring of n_scan + 4 layers: with k scans queued behind `scan`, n_scan + 4 - k layers are left).  Outputs host arrays
 * nodes/meas [max_len] int32, x [max_len][4], cnllr [max_len], P [max_len][16]; any may be NULL. */
int mht_forest_chain(mht_ctx* ctx, int32_t scan, int32_t node, int32_t max_len, int32_t* nodes, int32_t* meas,
                     double* x, double* cnllr, float* P, int32_t* n_out);
/* The same with float64 covariances (see mht_forest_leaves_f64) and the nodes' flag bytes, flags host [max_len] uint8 or NULL. */
int mht_forest_chain_f64(mht_ctx* ctx, int32_t scan, int32_t node, int32_t max_len, int32_t* nodes, int32_t* meas,
                         double* x, double* cnllr, double* P, uint8_t* flags, int32_t* n_out);

/* The streaming form (ABI 6; Tracker._apply_report: the window ancestors of the tracks a scan terminated, tracker.py:353-381 keeps them): the
 * chains of `count` nodes of layer `scan` are gathered by ONE launch queued behind what is already on the forest's stream, into a pinned
 * block of the library; nothing waits.  *ticket names the block; it stays valid until 8 later tickets have been issued.  f64 != 0:
 * float64 covariances as mht_forest_chain_f64.  mht_forest_chains_fetch waits for that launch only (not for scans queued behind it)
 * and copies chain `index` out: arrays as mht_forest_chain / _f64 (P: float32 or float64 [max_len][16] as asked at begin). */
int mht_forest_chains_begin(mht_ctx* ctx, int32_t scan, const int32_t* start_nodes, int32_t count, int32_t max_len, int32_t f64, int64_t* ticket);
int mht_forest_chains_fetch(mht_ctx* ctx, int64_t ticket, int32_t index, int32_t* nodes, int32_t* meas, double* x, double* cnllr, void* P,
                            uint8_t* flags, int32_t* n_out);

/* ---- step 7 of a scan: M-of-N track initiation on the device (tracker.py:264-278 -> initiators/m_of_n.py:215-478) -------------
 * What Tracker.__init__ hands to m_of_n.Initiator (tracker.py:66-72) plus the model constants the initiator imports (pv.P0, pv.Q's
 * sigmaQ, m_of_n.py:12-16 gamma = chi2(2).ppf(0.99)). */
typedef struct mht_initiator_config {
    int32_t m_required, n_checks;   /* Tracker.M_required / N_checks */
    int32_t max_meas;               /* measurements per scan */
    int32_t max_prelim;             /* preliminary tracks kept */
    int32_t max_born;               /* confirmed tracks per scan, before merging (behind a forest: at most 256, the report's capacity; more in a scan: MHT_E_CAPACITY) */
    double v_max;                   /* Tracker.maxSpeedMS */
    double gamma;                   /* gate of the preliminary tracks */
    double merge_threshold;         /* Tracker.mergeThreshold */
    double default_pd;
    float C[8], R[4], P0[16];       /* C_RADAR, R_RADAR(), pv.P0 */
    float sigma_q;                  /* scale of pv.Q */
} mht_initiator_config;
typedef struct mht_initiator mht_initiator;
int mht_initiator_create(mht_ctx* ctx, mht_initiator** out, const mht_initiator_config* cfg);
int mht_initiator_destroy(mht_initiator* in);
/* Initiator.processMeasurements (m_of_n.py:233-244) for one scan, asynchronous on the ctx stream: z dev (M,2) float32, the scan;
 * used dev [ceil(M/64)] uint64 or NULL: bit j set = measurement j was gated by a track and is not offered to the initiator
 * (tracker.py:266, MeasurementList.filterUnused); now = scan time stamp. */
int mht_initiator_step(mht_initiator* in, const float* z, int32_t M, const uint64_t* used, double now);
/* AIS messages for the initiator (Initiator.processMeasurements(radar, ais), m_of_n.py:233, :262-280): the messages of the scan it runs on
 * next, host array in LIST order (dT = time of the scan - time of the message); the ones no track took start preliminary tracks unless a
 * track with that identity exists or an existing one is too similar.  `used` (host, one byte per message) marks the taken ones for the
 * stand-alone mht_initiator_step; mht_forest_scan works them out itself (tracker.py:267-270: the identities still in an association set
 * behind the scan's pruning) and then runs the initiator behind the scan instead of next to the clustering. */
typedef struct mht_ais_init_msg { double state[4]; double dT; int32_t mmsi; int32_t pad; } mht_ais_init_msg;
int mht_initiator_set_ais(mht_initiator* in, const mht_ais_init_msg* msgs, int32_t nA, const uint8_t* used);
/* The targets the last step gave birth to (host arrays, any may be NULL): x0 [n][4] (float32 values), P0 [n][16],
 * meas [n] measurementNumber (1-based index among the UNUSED measurements, 0 for a merged target) -- and the sizes of the
 * initiator's lists.  Synchronises. */
int mht_initiator_born(mht_initiator* in, int32_t capacity, double* x0, float* P0, int32_t* meas, int32_t* n_born,
                       int32_t* n_prelim, int32_t* n_seeds);

/* Step 7 behind a forest step, on the stream, no host round trip: runs the scan's commit, offers the scan's unused measurements
 * (tracker.py:266) to the initiator and hands its confirmed candidates to Tracker.initiateTarget's device twin
 * (mht_forest_add_targets_dev, neighbour test included).  z / M: the scan just stepped (z = NULL: the copy mht_forest_step_host
 * staged); now: its time stamp.  The candidates and
 * their fate appear in the report of that scan (mht_scan_report::births).  The initiator must have been created on the same ctx. */
int mht_forest_initiate(mht_ctx* ctx, mht_initiator* in, const float* z, int32_t M, double now);

/* ---- one tracker on several devices: the independent per-cluster ILPs (tracker.py:228-236) are spread --------------------------------
 * Every device holds the same forest and is fed the same scans and births; the multi-target clusters are placed by size -- longest
 * processing time first on their column counts, each to the least loaded device; every device computes the same table from the same
 * data (cluster kernel: cl_owner) --, a single-target cluster goes to device (target index % shard_n).  sel_rel: dev [max_targets] int32 owned by the caller; after _begin it holds, for the targets whose
 * cluster this device solved, the selected child's ordinal inside the target's block, -1 elsewhere.  The caller combines the
 * devices' arrays with an element-wise MAX (all-reduce over RCCL) and calls _end, which finishes the scan for all targets.
 * Asynchronous on the ctx stream. */
int mht_forest_step_sharded_begin(mht_ctx* ctx, const float* z, int32_t M, int32_t shard_n, int32_t shard_i, int32_t* sel_rel);
int mht_forest_step_sharded_end(mht_ctx* ctx, const int32_t* sel_rel);
/* (ABI 6) A gating graph that is ONE big component on several devices (tracker.py:1155-1217 is one CBC call; any exact split will do): with an
 * exchange block of mht_forest_sharded_words() int32 -- [max_targets] selections as above, then [shard_n][8][260] files -- the clusters of
 * >= 24 targets (at most 8 per scan) are searched by ALL devices: the subtrees of the branch and bound are dealt out over every device's
 * workgroups, every device files its best selection and its value in its own slots (-1 = empty), the SAME element-wise MAX all-reduce over
 * the whole block gathers the files, and mht_forest_step_sharded_end (given the block) lets the smallest value win -- on every device alike. */
int mht_forest_sharded_words(mht_ctx* ctx, int32_t shard_n, int32_t* n_words);
int mht_forest_step_sharded_begin2(mht_ctx* ctx, const float* z, int32_t M, int32_t shard_n, int32_t shard_i, int32_t* xch, int32_t n_words);

/* One radar scan of Tracker.addMeasurementList (tracker.py:162-307) in one call, nothing waits for the device: steps 1-6
 * (mht_forest_step_host), step 7 (mht_forest_initiate, skipped when `in` is NULL), mht_forest_report_begin.
 * With an initiator the scans are STREAMED: the scan's commit, the admission of what its initiator gave birth to and the report's push ride
 * in the next scan's grow launch (which starts while this scan's ILP launch is still running); the initiator is a one-workgroup launch on a
 * side stream of the forest's; the report is complete in its pinned host block when the words the pushing workgroups post there say so
 * (mht_forest_report_get waits for them, not for an event).  mht_synchronize also waits for the side stream. */
int mht_forest_scan(mht_ctx* ctx, mht_initiator* in, const float* z_host, int32_t M, double now);

/* ---- a group of independent sectors on one device (BASELINE config 4: four sensor sectors = four independent Tracker
 * instances, pymht/tracker.py:39-137; nothing in tracker.py:162-307 couples two Tracker objects) -------------------------------
 * The members' forests step TOGETHER with one launch per stage (grow, cluster, ILP) for the whole group: a single sector is a
 * chain of dependent round trips that leaves most of the GPU idle, S sectors cost about one such chain.  Results are exactly
 * those of stepping every member with mht_forest_step (tests/test_sectors_gpu.py).
 *   ctxs   n contexts (1 <= n <= 32) that own a forest each; same device, same stream, same forest configuration
 *   z      host array of n device pointers, z[i] = member i's scan, dev (M[i],2) float32
 *   M      host array of n measurement counts
 * After a group step every per-forest call (mht_forest_report, _add_targets*, _leaves, ...) works on the members as usual. */
typedef struct mht_group mht_group;
int mht_group_create(mht_group** out, int32_t n, mht_ctx* const* ctxs);
int mht_group_step(mht_group* g, const float* const* z, const int32_t* M);
int mht_group_destroy(mht_group* g);

#ifdef __cplusplus
}
#endif
#endif /* MHT_AMD_H */
