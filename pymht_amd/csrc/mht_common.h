// Internal declarations shared by the translation units of libmht_amd.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mht_amd.h"
#include "mht_math.h"

namespace mht {

void set_error(const char* fmt, ...);

#define MHT_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            mht::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MHT_E_HIP;                                                                    \
        }                                                                                        \
    } while (0)

#define MHT_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            mht::set_error(__VA_ARGS__);       \
            return MHT_E_INVALID;              \
        }                                      \
    } while (0)

// A grow-only device buffer owned by the ctx (workspace).
struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return MHT_OK;
        if (ptr) MHT_HIP_CHECK(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
        size_t want = need + need / 2 + 4096;
        MHT_HIP_CHECK(hipMalloc(&ptr, want));
        // (hipMemset on device memory goes out on the NULL stream and returns before it has run; the caller's stream may be a
        // non-blocking one that does not wait for the null stream -- the zeros must not land behind what that stream writes here)
        MHT_HIP_CHECK(hipMemset(ptr, 0, want));
        MHT_HIP_CHECK(hipStreamSynchronize(nullptr));
        bytes = want;
        return MHT_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};

constexpr int GATE_TILE = 32;    // leaves per workgroup tile of grow_kernel
constexpr int GATE_THREADS = 512;
constexpr int EMIT_THREADS = 64;  // one wavefront, one leaf per lane
constexpr int MAX_MEAS = 4096;    // 64 hit-mask words per leaf, one per lane
constexpr int EDGE_SEGS = 64;      // the (target, measurement) edge list is written in 64 independently counted segments
constexpr int MAXPD = 15;         // longest root->leaf path kept per hypothesis (N-scan window + 1)
#ifndef MHT_FG_THREADS
#define MHT_FG_THREADS 256
#endif
#ifndef MHT_FG_CAP
#define MHT_FG_CAP 96
#endif
constexpr int FG_THREADS = MHT_FG_THREADS;   // fgrow_kernel: one workgroup per target
constexpr int FG_CAP_SOLO = 128;          //   ... in the one-sector launch: every scan in nine of the headline stream has a target with 97..128 leaves, and a
                                          //   second pass over it sets the duration of the whole launch (32 instead of 20 us); 45 KB of LDS, three workgroups per CU
constexpr int FG_CAP = MHT_FG_CAP;        //   leaves of a target handled per pass, one per lane of two wavefronts (more: chunks, two passes)
constexpr int FG_REGIONS = 8;     //   regions of the node index space, one child counter each (one per XCD)
// covariance-value table (mht_vtab.h), sized by the state dimension of the build (mht_math.h: NX, NP, NK)
constexpr int VT_PW = NP / 2;                   // 64-bit words of one covariance value (8 at 4 states, 18 at 6)
constexpr int GKQ = (4 + NK + 3 + 3) / 4;       // float4 per gains row: S^-1 (4), K (NX x 2), score constant, two gate half-axes (4 / 5)
constexpr int GKF = GKQ * 4;                    // floats per gains row

// device-side status word of a ctx (sticky until read)
struct DevStatus {
    int overflow;       // children did not fit `out`
    int n_children;     // children produced by the last gate
    int n_dead;         // forest: leaves the grow kernel skipped because similar-state pruning had taken them out of the tree
    int pad[1];
    // forest: device wall-clock stamps of the scan's stages (s_memrealtime, 10 ns ticks), written by thread 0 of the first workgroup of
    // each launch: [0] grow start, [1] cluster start, [2] first ILP launch start, [3] similar-state pruning start (0: did not run),
    // [4] end of the last ILP workgroup (atomic max).  The commit turns them into the report's per-stage times (mht_scan_report).
    unsigned long long t[6];
};

struct Forest;

// Copy an argument block that lives in HBM (written once by the host) into registers through the CONSTANT address space: scalar
// loads, shared by the whole wavefront (batched launches: blockIdx.y picks the sector's block).
template <typename T>
__device__ __forceinline__ void load_args(T& dst, const T* src) {
    static_assert(sizeof(T) % 4 == 0, "argument blocks are copied dword-wise");
    const __attribute__((address_space(4))) unsigned* s = (const __attribute__((address_space(4))) unsigned*)src;
    unsigned* d = reinterpret_cast<unsigned*>(&dst);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) d[i] = s[i];
}

}  // namespace mht

struct mht_ctx {
    int device = 0;
    int n_cu = 256;          // compute units of the device (queried in mht_create)
    hipStream_t stream = nullptr;
    mht::Scratch hitmask;    // stateless seams: look-back tile states / BLP scratch
    mht::Scratch counts;     // stateless seams: tile ticket / clustering scratch
    unsigned gate_epoch = 0;
    mht::DevStatus* status = nullptr;   // device
    mht::Forest* forest = nullptr;
    // dynamic-LDS limits already raised with hipFuncSetAttribute (per context: the attribute is per device)
    size_t lds_attr_gate = 0, lds_attr_cluster = 0, lds_attr_blp = 0, lds_attr_fgrow = 0, lds_attr_blp_uf = 0;
};
