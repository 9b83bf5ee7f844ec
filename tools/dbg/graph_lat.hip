// per-kernel cost of back-to-back tiny launches: direct vs hipGraph replay
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void tiny(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main() {
    int *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t st; hipStreamCreate(&st);
    const int N = 128, REP = 20;
    for (int w = 0; w < 2; w++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; r++) for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(128), dim3(256), 0, st, d);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        auto t2 = std::chrono::steady_clock::now();
        if (w) printf("direct: enqueue %.2f us/kernel, total %.2f us/kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * REP), std::chrono::duration<double, std::micro>(t2 - t0).count() / (N * REP));
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(128), dim3(256), 0, st, d);
    hipStreamEndCapture(st, &g);
    auto i0 = std::chrono::steady_clock::now();
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    auto i1 = std::chrono::steady_clock::now();
    printf("instantiate %d nodes: %s, %.1f us\n", N, hipGetErrorString(e), std::chrono::duration<double, std::micro>(i1 - i0).count());
    for (int w = 0; w < 2; w++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < REP; r++) hipGraphLaunch(ge, st);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        auto t2 = std::chrono::steady_clock::now();
        if (w) printf("graph:  enqueue %.2f us/kernel, total %.2f us/kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (N * REP), std::chrono::duration<double, std::micro>(t2 - t0).count() / (N * REP));
    }
    int h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("count %d\n", h);
    return 0;
}
