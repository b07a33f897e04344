// Development micro-benchmark: cost of k dependent global round trips per workgroup at the grow_kernel's grid shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void chase(const int* __restrict__ table, int* out, int rounds, int stage_words) {
    extern __shared__ int lds[];
    // stage a small shared table (every workgroup reads the same words, like the scan / offsets staging)
    for (int j = threadIdx.x; j < stage_words; j += blockDim.x) lds[j] = table[j];
    __syncthreads();
    int idx = (blockIdx.x * 37 + threadIdx.x) & 4095;
    for (int r = 0; r < rounds; ++r) idx = table[4096 + ((idx + lds[idx % (stage_words > 0 ? stage_words : 1)]) & 0xfffff)] & 4095;   // dependent loads, scattered
    if (idx == 123456) out[0] = idx;
    out[1 + blockIdx.x] = idx;
}
int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 845;
    int* table; int* out;
    hipMalloc(&table, (4096 + (1 << 20)) * 4); hipMalloc(&out, (blocks + 8) * 4);
    int* h = (int*)malloc((4096 + (1 << 20)) * 4);
    for (int i = 0; i < 4096 + (1 << 20); ++i) h[i] = rand() & 0xfffff;
    hipMemcpy(table, h, (4096 + (1 << 20)) * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int cfgs[][2] = {{0, 0}, {0, 1536}, {1, 1536}, {2, 1536}, {4, 1536}, {8, 1536}, {16, 1536}};
    for (auto& c : cfgs) {
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 16384, 0, table, out, c[0], c[1]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int K = 200;
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 16384, 0, table, out, c[0], c[1]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("blocks=%d stage_words=%d dependent_rounds=%d : %.2f us per launch (back-to-back)\n", blocks, c[1], c[0], 1e3 * ms / K);
    }
    return 0;
}
