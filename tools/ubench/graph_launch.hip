// Micro-benchmark: host and device cost of 4 dependent tiny kernels launched directly vs as one hipGraph
// (with and without per-launch kernel-node parameter updates).  hipcc --offload-arch=gfx950 -O2 graph_launch.hip -o graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Args { long pad[60]; int* p; int v; };      // ~500 B by-value argument, like the tracker's kernels
__global__ void k(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += a.v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    Args a = {}; a.p = d; a.v = 1;
    const int N = 2000;
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, a);
    CK(hipStreamSynchronize(s));
    double t0 = now();
    for (int i = 0; i < N; ++i) for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, a);
    double t1 = now(); CK(hipStreamSynchronize(s)); double t2 = now();
    printf("direct : host %.2f us per 4-kernel step, wall %.2f us per step\n", 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, s, a);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = now();
    for (int i = 0; i < N; ++i) CK(hipGraphLaunch(ge, s));
    t1 = now(); CK(hipStreamSynchronize(s)); t2 = now();
    printf("graph  : host %.2f us per step, wall %.2f us per step\n", 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    t0 = now();
    for (int i = 0; i < N; ++i) {
        a.v = i & 1;
        void* kp[] = {&a};
        for (size_t j = 0; j < nn; ++j) {
            hipKernelNodeParams p = {}; p.func = (void*)k; p.gridDim = dim3(64); p.blockDim = dim3(256); p.kernelParams = kp;
            CK(hipGraphExecKernelNodeSetParams(ge, nodes[j], &p));
        }
        CK(hipGraphLaunch(ge, s));
    }
    t1 = now(); CK(hipStreamSynchronize(s)); t2 = now();
    printf("graph + 4 node updates: host %.2f us per step, wall %.2f us per step\n", 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N);
    return 0;
}
