// M-of-N track initiation on the device: step 7 of a scan (reference: Tracker.addMeasurementList, pymht/tracker.py:264-278 ->
// pymht/initiators/m_of_n.py: Initiator.processMeasurements :233-244, __processPreliminaryTracks :246-378,
// __processInitiators :380-413 / :425-478, _solve_global_nearest_neighbour :24-104, _merge_targets / merge :107-154).
//
// One workgroup per initiator and scan (the problem is tiny: the measurements no track gated, ~50 at the headline config; what
// matters is that it runs ON the stream, behind the scan's commit, so that a scan needs no host round trip any more):
//   (1) preliminary tracks (float32 state / covariance, m-of-n counters): predict, gate (chi2 0.99) against the unused
//       measurements, global-nearest-neighbour assignment, Kalman update, verdict -> confirmed tracks become new targets;
//   (2) what is left is paired with last scan's leftovers ("initiators") under v_max * dt, GNN again -> new preliminary tracks
//       (unless one like it exists already);
//   (3) what is still left becomes next scan's initiators; confirmed candidates closer than the merge threshold are averaged.
// GNN = the reference's Hungarian assignment on a big-M padded square matrix = a minimum-cost maximum-cardinality matching of the
// sparse "allowed" graph; its connected components are tiny (a leftover has at most a couple of partners within v_max * dt), so
// it is solved exactly per component by successive shortest augmenting paths, one thread per component.
// Arithmetic follows the reference's dtypes (float32 states, covariances, distances); its host BLAS orders 4-term dot products
// differently from any fixed order, so states agree to ~1e-6 relative, decisions (births, counters) exactly (tests/golden G8).
#include "mht_kernels.h"
#include "mht_init_dev.h"
#include <new>
#include <string.h>


namespace mht {
__global__ __launch_bounds__(INIT_THREADS) void initiator_kernel(const InitArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_gnn[INIT_GNN_LDS];
    initiator_body(a, s_gnn, INIT_GNN_LDS);
}
}  // namespace mht

// ------------------------------------------------------------------------------------------------------------------------------
struct mht_initiator {
    mht_ctx* ctx = nullptr;
    mht_initiator_config cfg = {};
    char* arena = nullptr; size_t arena_bytes = 0;
    mht::InitArgs args = {};          // pointers into the arena (pstate/pstate2 ... are swapped every step)
    int flip = 0;
    void* host = nullptr; size_t host_bytes = 0;      // pinned staging for mht_initiator_born / _counts
    int ais_pending = 0;              // messages handed in for the next run (mht_initiator_set_ais)
    bool ais_used_valid = false;      // ... with caller-provided used flags (else: the forest computes them, or none was taken)
    bool ais_used_by_forest = false;
};

namespace mht {
struct IArena {
    char* base; size_t off;
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += n * sizeof(T);
        return p;
    }
};
static void init_layout(mht_initiator* in, IArena& ar) {
    InitArgs& a = in->args;
    const size_t Mc = in->cfg.max_meas, Pc = in->cfg.max_prelim, Bc = in->cfg.max_born, V = Mc + (Pc > Mc ? Pc : Mc) + 64;
    a.st = ar.take<InitDev>(1);
    a.seeds = ar.take<float>(2 * Mc);
    a.pstate = ar.take<float>(4 * Pc); a.pcov = ar.take<float>(16 * Pc); a.pn = ar.take<int32_t>(Pc); a.pm = ar.take<int32_t>(Pc);
    a.pstate2 = ar.take<float>(4 * Pc); a.pcov2 = ar.take<float>(16 * Pc); a.pn2 = ar.take<int32_t>(Pc); a.pm2 = ar.take<int32_t>(Pc);
    a.e_row = ar.take<int32_t>(INIT_ECAP); a.e_col = ar.take<int32_t>(INIT_ECAP); a.e_cost = ar.take<double>(INIT_ECAP); a.e_next = ar.take<int32_t>(INIT_ECAP);
    a.node_parent = ar.take<int32_t>(V); a.row_head = ar.take<int32_t>(V); a.row_next = ar.take<int32_t>(V); a.comp_head = ar.take<int32_t>(V);
    a.match_row = ar.take<int32_t>(V); a.match_col = ar.take<int32_t>(V); a.bf_dist = ar.take<double>(2 * V); a.bf_pred = ar.take<int32_t>(V);
    a.comp_nodes = ar.take<int32_t>(Mc); a.upos = ar.take<int32_t>(Mc);
    a.K = ar.take<float>(8 * Pc); a.pred = ar.take<float>(4 * Pc); a.tmeas = ar.take<int32_t>(Pc);
    a.born_x = ar.take<double>(4 * Bc); a.born_P = ar.take<float>(16 * Bc); a.born_flags = ar.take<uint8_t>(Bc); a.born_pd = ar.take<double>(Bc);
    a.born_meas = ar.take<int32_t>(Bc); a.born_n = ar.take<int32_t>(4);
    a.pmmsi = ar.take<int32_t>(Pc); a.pmmsi2 = ar.take<int32_t>(Pc);
    a.Acap = (int)Mc;
    a.ais = ar.take<AisInitMsg>(Mc); a.ais_used = ar.take<unsigned char>(Mc); a.ais_x64 = ar.take<double>(4 * Mc);
}

// launch one scan of the initiator on the ctx stream (used: device bit mask or null)
// the argument block of one scan (the two preliminary-track buffers alternate)
void initiator_scan_args(mht_initiator* in, const float* z, int M, const unsigned long long* used, double now, InitArgs& a) {
    a = in->args;
    if (in->flip) {      // the kernel compacts the surviving preliminary tracks from (pstate ..) into (pstate2 ..): alternate
        float* t; int32_t* u;
        t = a.pstate; a.pstate = a.pstate2; a.pstate2 = t;
        t = a.pcov; a.pcov = a.pcov2; a.pcov2 = t;
        u = a.pn; a.pn = a.pn2; a.pn2 = u;
        u = a.pm; a.pm = a.pm2; a.pm2 = u;
        u = a.pmmsi; a.pmmsi = a.pmmsi2; a.pmmsi2 = u;
    }
    in->flip ^= 1;
    a.z = z; a.M = M; a.used = used; a.now = now;
    a.nA = in->ais_pending;               // (consumed by this run)
    if (!in->ais_used_valid && !in->ais_used_by_forest) a.ais_used = nullptr;
    in->ais_pending = 0; in->ais_used_valid = false; in->ais_used_by_forest = false;
}
int initiator_ais_pending(const mht_initiator* in) { return in->ais_pending; }
int initiator_mreq(const mht_initiator* in) { return in->cfg.m_required; }
void initiator_ais_ptrs(mht_initiator* in, const AisInitMsg** msgs, unsigned char** used) {      // the forest fills the used flags behind its commit
    *msgs = in->args.ais; *used = const_cast<unsigned char*>(in->args.ais_used);
    in->ais_used_by_forest = true;
}

int initiator_launch(mht_initiator* in, const float* z, int M, const unsigned long long* used, double now) {
    InitArgs a;
    initiator_scan_args(in, z, M, used, now, a);
    hipLaunchKernelGGL(initiator_kernel, dim3(1), dim3(INIT_THREADS), 0, in->ctx->stream, a);
    MHT_HIP_CHECK(hipGetLastError());
    return MHT_OK;
}
void initiator_born_ptrs(const mht_initiator* in, const double** x, const float** P, const uint8_t** fl, const double** pd, const int32_t** meas,
                         const int32_t** n, int* cap, mht_ctx** ctx) {
    const InitArgs& a = in->args;
    *x = a.born_x; *P = a.born_P; *fl = a.born_flags; *pd = a.born_pd; *meas = a.born_meas; *n = a.born_n; *cap = in->cfg.max_born; *ctx = in->ctx;
}
}  // namespace mht

using namespace mht;

extern "C" int mht_initiator_destroy(mht_initiator* in) {
    if (!in) return MHT_OK;
    if (in->ctx) { (void)hipSetDevice(in->ctx->device); (void)hipStreamSynchronize(in->ctx->stream); }
    if (in->arena) (void)hipFree(in->arena);
    if (in->host) (void)hipHostFree(in->host);
    delete in;
    return MHT_OK;
}

extern "C" int mht_initiator_create(mht_ctx* ctx, mht_initiator** out, const mht_initiator_config* cfg) {
    MHT_REQUIRE(ctx && out && cfg, "mht_initiator_create: null argument");
    MHT_REQUIRE(cfg->max_meas >= 1 && cfg->max_meas <= 4096 && cfg->max_prelim >= 1 && cfg->max_prelim <= 16384 && cfg->max_born >= 1 && cfg->max_born <= 4096,
                "mht_initiator_create: need 1 <= max_meas <= 4096, 1 <= max_prelim <= 16384, 1 <= max_born <= 4096");
    MHT_REQUIRE(cfg->m_required >= 1 && cfg->n_checks >= cfg->m_required, "mht_initiator_create: need 1 <= m_required <= n_checks");
    MHT_HIP_CHECK(hipSetDevice(ctx->device));
    mht_initiator* in = new (std::nothrow) mht_initiator();
    MHT_REQUIRE(in, "mht_initiator_create: out of host memory");
    in->ctx = ctx;
    in->cfg = *cfg;
    IArena probe{nullptr, 0};
    init_layout(in, probe);
    in->arena_bytes = probe.off + 4096;
    if (hipMalloc(reinterpret_cast<void**>(&in->arena), in->arena_bytes) != hipSuccess) {
        delete in;
        set_error("mht_initiator_create: hipMalloc of %zu bytes failed", in->arena_bytes);
        return MHT_E_HIP;
    }
    IArena ar{in->arena, 0};
    init_layout(in, ar);
    InitArgs& a = in->args;
    a.Mcap = cfg->max_meas; a.Pcap = cfg->max_prelim; a.born_cap = cfg->max_born;
    a.Mreq = cfg->m_required; a.Nreq = cfg->n_checks; a.v_max = cfg->v_max; a.gamma = cfg->gamma; a.merge_threshold = cfg->merge_threshold;
    a.default_pd = cfg->default_pd; a.sigma_q = cfg->sigma_q;
    for (int i = 0; i < 8; ++i) a.C[i] = cfg->C[i];
    for (int i = 0; i < 4; ++i) a.R[i] = cfg->R[i];
    for (int i = 0; i < 16; ++i) a.P0[i] = cfg->P0[i];
    MHT_HIP_CHECK(hipMemsetAsync(in->arena, 0, in->arena_bytes, ctx->stream));
    in->host_bytes = (size_t)cfg->max_born * (32 + 64 + 4 + 8) + 256;
    MHT_HIP_CHECK(hipHostMalloc(&in->host, in->host_bytes, hipHostMallocDefault));
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *out = in;
    return MHT_OK;
}

// The AIS messages of the scan the initiator runs on next (Initiator.processMeasurements(radar, ais), m_of_n.py:233): host array in LIST
// order; `used` (host, one byte per message, or null = none) marks the ones a track took -- the forest (mht_forest_scan) works that out
// itself behind its commit and ignores it.
extern "C" int mht_initiator_set_ais(mht_initiator* in, const mht_ais_init_msg* msgs, int32_t nA, const uint8_t* used) {
    MHT_REQUIRE(in && nA >= 0 && (msgs || nA == 0), "mht_initiator_set_ais: null argument");
    MHT_REQUIRE(nA <= in->args.Acap, "mht_initiator_set_ais: %d messages exceed max_meas=%d", nA, in->args.Acap);
    static_assert(sizeof(mht_ais_init_msg) == sizeof(AisInitMsg), "ABI struct");
    MHT_HIP_CHECK(hipSetDevice(in->ctx->device));
    if (nA) MHT_HIP_CHECK(hipMemcpyAsync(const_cast<AisInitMsg*>(in->args.ais), msgs, (size_t)nA * sizeof(AisInitMsg), hipMemcpyHostToDevice, in->ctx->stream));
    if (nA && used) MHT_HIP_CHECK(hipMemcpyAsync(const_cast<unsigned char*>(in->args.ais_used), used, (size_t)nA, hipMemcpyHostToDevice, in->ctx->stream));
    in->ais_pending = nA;
    in->ais_used_valid = nA > 0 && used != nullptr;
    return MHT_OK;
}

extern "C" int mht_initiator_step(mht_initiator* in, const float* z, int32_t M, const uint64_t* used, double now) {
    MHT_REQUIRE(in && (z || M == 0), "mht_initiator_step: null argument");
    MHT_REQUIRE(M >= 0 && M <= in->cfg.max_meas, "mht_initiator_step: M=%d exceeds max_meas=%d", M, in->cfg.max_meas);
    MHT_HIP_CHECK(hipSetDevice(in->ctx->device));
    return initiator_launch(in, z, M, reinterpret_cast<const unsigned long long*>(used), now);
}

extern "C" int mht_initiator_born(mht_initiator* in, int32_t capacity, double* x0, float* P0, int32_t* meas, int32_t* n_born,
                                  int32_t* n_prelim, int32_t* n_seeds) {
    MHT_REQUIRE(in && n_born, "mht_initiator_born: null argument");
    MHT_HIP_CHECK(hipSetDevice(in->ctx->device));
    hipStream_t st = in->ctx->stream;
    const InitArgs& a = in->args;
    const int Bc = in->cfg.max_born;
    char* h = static_cast<char*>(in->host);
    const size_t o_x = 0, o_P = o_x + (size_t)Bc * 32, o_m = o_P + (size_t)Bc * 64, o_s = o_m + (size_t)Bc * 4 + 8;
    MHT_HIP_CHECK(hipMemcpyAsync(h + o_s, a.st, sizeof(InitDev), hipMemcpyDeviceToHost, st));
    MHT_HIP_CHECK(hipMemcpyAsync(h + o_x, a.born_x, (size_t)Bc * 32, hipMemcpyDeviceToHost, st));
    MHT_HIP_CHECK(hipMemcpyAsync(h + o_P, a.born_P, (size_t)Bc * 64, hipMemcpyDeviceToHost, st));
    MHT_HIP_CHECK(hipMemcpyAsync(h + o_m, a.born_meas, (size_t)Bc * 4, hipMemcpyDeviceToHost, st));
    MHT_HIP_CHECK(hipStreamSynchronize(st));
    const InitDev* d = reinterpret_cast<const InitDev*>(h + o_s);
    *n_born = d->n_born;
    if (n_prelim) *n_prelim = d->n_prelim;
    if (n_seeds) *n_seeds = d->n_seeds;
    const int n = d->n_born < capacity ? d->n_born : capacity;
    if (x0) memcpy(x0, h + o_x, (size_t)n * 32);
    if (P0) memcpy(P0, h + o_P, (size_t)n * 64);
    if (meas) memcpy(meas, h + o_m, (size_t)n * 4);
    if (d->overflow) {
        set_error("initiator: a capacity was exceeded (max_prelim=%d, max_born=%d, %d edges per assignment problem)", in->cfg.max_prelim, in->cfg.max_born, INIT_ECAP);
        return MHT_E_CAPACITY;
    }
    return MHT_OK;
}
