// Micro-benchmark: does a kernel launched with hipExtAnyOrderLaunch start while its predecessor IN THE SAME STREAM is still running?
// A: 256 workgroups x 150 KB of LDS, each spins for `spin` us; B: 512 small workgroups, launched right behind it.
// hipcc --offload-arch=gfx950 -O2 anyorder.hip -o anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned long long wc() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__global__ void kA(unsigned long long* st, int ticks, unsigned long long* flag, unsigned long long epoch) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = rt();
#ifdef BIGREGS
    asm volatile("v_mov_b32 v254, 1" ::: "v254");
#endif
    if (threadIdx.x == 0) { lds[0] = 1; st[blockIdx.x * 2] = t0; }
    // workgroup b spins b % 4 + 1 quarters of `ticks`
    const unsigned long long until = t0 + (unsigned long long)ticks * (1 + ((blockIdx.x >> SPINSHIFT) & 3)) / 4;
    while (rt() < until) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) { st[blockIdx.x * 2 + 1] = rt(); __hip_atomic_store(&flag[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
struct Big { long pad[150]; };
__global__ void kB(unsigned long long* st, const unsigned long long* flag, unsigned long long epoch, int wait, Big big) {
    const unsigned long long t0 = rt();
    if (threadIdx.x == 0) {
        st[blockIdx.x * 2] = t0 + (big.pad[3] & 0);
        if (wait) while (__hip_atomic_load(&flag[blockIdx.x & 255], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(8);
        st[blockIdx.x * 2 + 1] = rt();
    }
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1; const int nA = argc > 2 ? atoi(argv[2]) : 256;      // 0: plain launches, 1: B with hipExtAnyOrderLaunch, 2: B on a second stream
    unsigned long long *sa, *sb, *flag;
    CK(hipMalloc(&sa, 512 * 16)); CK(hipMalloc(&sb, 1024 * 16)); CK(hipMalloc(&flag, 512 * 8)); CK(hipMemset(flag, 0, 512 * 8));
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)kA, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    Big big = {};
    hipEvent_t ev[8]; for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    const int chain = argc > 3 ? atoi(argv[3]) : 1;      // launches pairs per rep without a host synchronisation in between
    for (int rep = 0; rep < 6; ++rep) {
      unsigned long long epoch = 0;
      const auto h0 = std::chrono::steady_clock::now();
      for (int it = 0; it < chain; ++it) {
        epoch = (unsigned long long)rep * 1000 + it + 1;
        if (mode == 3 && it > 0) CK(hipStreamWaitEvent(s, ev[(it - 1) & 7], 0));      // A(it) behind B(it - 1)
        hipLaunchKernelGGL(kA, dim3(nA), dim3(256), 150 * 1024, s, sa, 3000, flag, epoch);      // 30 us (100 MHz ticks)
        if (mode == 3) { hipLaunchKernelGGL(kB, dim3(768), dim3(256), 40 * 1024, s2, sb, flag, epoch, 1, big); CK(hipEventRecord(ev[it & 7], s2)); continue; }
        if (mode == 1) hipExtLaunchKernelGGL(kB, dim3(768), dim3(256), 40 * 1024, s, nullptr, nullptr, hipExtAnyOrderLaunch, sb, flag, epoch, 1, big);
        else if (mode == 2) hipLaunchKernelGGL(kB, dim3(768), dim3(256), 40 * 1024, s2, sb, flag, epoch, 1, big);
        else hipLaunchKernelGGL(kB, dim3(768), dim3(256), 40 * 1024, s, sb, flag, epoch, 1, big);
      }
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        const double per_it = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() * 1e6 / chain;
        unsigned long long ha[1024], hb[1536];
        CK(hipMemcpy(ha, sa, sizeof(ha), hipMemcpyDeviceToHost)); CK(hipMemcpy(hb, sb, sizeof(hb), hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0, bl = 0;
        for (int i = 0; i < nA; ++i) { a0 = std::min(a0, ha[2 * i]); a1 = std::max(a1, ha[2 * i + 1]); }
        for (int i = 0; i < 768; ++i) { b0 = std::min(b0, hb[2 * i]); bl = std::max(bl, hb[2 * i]); b1 = std::max(b1, hb[2 * i + 1]); }
        printf("mode %d rep %d (chain %d): A runs 0 .. %.1f us; B first start %.1f, last start %.1f, last end %.1f; wall %.1f us per iteration\n", mode, rep, chain, (a1 - a0) / 100.0, ((double)b0 - (double)a0) / 100.0,
               ((double)bl - (double)a0) / 100.0, ((double)b1 - (double)a0) / 100.0, per_it);
    }
    return 0;
}
