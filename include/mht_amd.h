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

#define MHT_ABI_VERSION 1

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

/* ILP status (mht_solve_blp / forest step) */
#define MHT_BLP_CERTIFIED 1   /* Lagrangian certificate: conflict-free minimisers + complementary slackness */
#define MHT_BLP_BRANCHED 2    /* proven optimal by branch and bound on the GPU */
#define MHT_BLP_NODE_LIMIT 3  /* node limit: best feasible point returned (MHT_E_LIMIT) */

typedef struct mht_ctx mht_ctx;

/* The linear-Gaussian model: what Tracker.__init__ reads off the model module (tracker.py:54-59, pv.py:7-34)
 * plus the two scalars the gate and the score need (tracker.py:107, :110).  Row-major float32. */
typedef struct mht_model {
    float A[16]; /* Phi(radarPeriod) */
    float Q[16]; /* Q(radarPeriod)   */
    float C[8];  /* C_RADAR (2x4)    */
    float R[4];  /* R_RADAR() (2x2)  */
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

#ifdef __cplusplus
}
#endif
#endif /* MHT_AMD_H */
