// latency of dependent global loads by memory scope (one wave, pointer chase over an L2-resident array)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
template <int SCOPE> __global__ void chase(const unsigned *p, int steps, long long *out, unsigned *sink) {
    unsigned i = threadIdx.x;
    long long t0 = wall_clock64();
    for (int s = 0; s < steps; s++) {
        if (SCOPE == 0) i = p[i];
        else if (SCOPE == 1) i = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 2) i = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else i = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = i;
}
__global__ void chase_atomic(unsigned long long *p, int steps, long long *out, unsigned *sink) {
    unsigned i = threadIdx.x;
    long long t0 = wall_clock64();
    for (int s = 0; s < steps; s++) { unsigned long long r = atomicMin(p + i, ~0ull); i = (unsigned)r; }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = i;
}
int main() {
    for (int n : {40000, 4000000}) {
        std::vector<unsigned> h(n); std::iota(h.begin(), h.end(), 0u);
        std::mt19937 g(1); std::shuffle(h.begin(), h.end(), g);
        unsigned *d, *sink; long long *out; unsigned long long *d64;
        hipMalloc(&d, n * 4); hipMalloc(&sink, 256); hipMalloc(&out, 8); hipMalloc(&d64, (size_t)n * 8);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<unsigned long long> h64(n); for (int i = 0; i < n; i++) h64[i] = h[i];
        hipMemcpy(d64, h64.data(), (size_t)n * 8, hipMemcpyHostToDevice);
        const int steps = 2000;
        long long t;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(chase<0>, 1, 64, 0, 0, d, steps, out, sink); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); if (rep) printf("n=%d plain      %.0f ns/load\n", n, t * 10.0 / steps);
            hipLaunchKernelGGL(chase<1>, 1, 64, 0, 0, d, steps, out, sink); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); if (rep) printf("n=%d workgroup  %.0f ns/load\n", n, t * 10.0 / steps);
            hipLaunchKernelGGL(chase<2>, 1, 64, 0, 0, d, steps, out, sink); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); if (rep) printf("n=%d agent      %.0f ns/load\n", n, t * 10.0 / steps);
            hipLaunchKernelGGL(chase<3>, 1, 64, 0, 0, d, steps, out, sink); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); if (rep) printf("n=%d system     %.0f ns/load\n", n, t * 10.0 / steps);
            hipLaunchKernelGGL(chase_atomic, 1, 64, 0, 0, d64, steps, out, sink); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); if (rep) printf("n=%d atomic-min %.0f ns/op\n", n, t * 10.0 / steps);
        }
        hipFree(d); hipFree(sink); hipFree(out); hipFree(d64);
    }
    return 0;
}
