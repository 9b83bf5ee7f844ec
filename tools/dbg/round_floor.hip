// What bounds a round of the row reduction (developer microbenchmark): back-to-back dependent launches by grid size, and a grid barrier
// among G workgroups of one XCD (blocks 0, 8, 16, ...) or of all XCDs, with one dependent agent-scope load chain in between.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_k(int *p, int L) { if (p && blockIdx.x == 0 && threadIdx.x == 0 && L < 0) p[0] = L; }
__global__ void touch_k(int *p, int L) { if (blockIdx.x * 256 + threadIdx.x < 64) p[threadIdx.x] = p[threadIdx.x + 64] + L; }
struct Ctl { unsigned arrive, gen; int err; };
__device__ void gbar(Ctl *c, int G, unsigned &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned a = atomicAdd(&c->arrive, 1u) + 1u;
        if (a == (gen + 1u) * (unsigned)G) __hip_atomic_store(&c->gen, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else { long long s = 0; while (__hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= gen) { __builtin_amdgcn_s_sleep(1); if (++s > (1ll << 24)) { c->err = 1; break; } } }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    gen++;
    __syncthreads();
}
// R rounds: barrier + `chain` dependent agent-scope loads per round
template <int STRIDE> __global__ void persist_k(Ctl *c, int G, int R, int chain, const unsigned *p, long long *out, unsigned *sink) {
    if (blockIdx.x % STRIDE) return;
    unsigned gen = 0, i = threadIdx.x + (blockIdx.x / STRIDE) * 256;
    const long long t0 = wall_clock64();
    for (int r = 0; r < R; r++) {
        for (int k = 0; k < chain; k++) i = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gbar(c, G, gen);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = wall_clock64() - t0;
    sink[threadIdx.x] = i;
}
int main() {
    int *d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t st; hipStreamCreate(&st);
    for (int grid : {1, 64, 256, 512, 1024, 2048}) {
        for (int which = 0; which < 2; which++) {
            for (int rep = 0; rep < 2; rep++) {
                hipStreamSynchronize(st);
                auto t0 = std::chrono::steady_clock::now();
                const int N = 2000;
                for (int L = 0; L < N; L++) { if (which) hipLaunchKernelGGL(touch_k, dim3(grid), dim3(256), 0, st, d, L); else hipLaunchKernelGGL(empty_k, dim3(grid), dim3(256), 0, st, d, L); }
                hipStreamSynchronize(st);
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (rep) printf("launch chain grid %4d x 256 (%s): %.2f us per launch\n", grid, which ? "touch" : "empty", us);
            }
        }
    }
    const int n = 1 << 16;
    unsigned *h = new unsigned[n]; for (int i = 0; i < n; i++) h[i] = (i * 9973u + 12345u) % n;
    unsigned *p, *sink; long long *out; Ctl *c;
    hipMalloc(&p, n * 4); hipMalloc(&sink, 1024); hipMalloc(&out, 8); hipMalloc(&c, sizeof(Ctl));
    hipMemcpy(p, h, n * 4, hipMemcpyHostToDevice);
    for (int G : {16, 32, 64, 128}) for (int chain : {0, 4}) {
        long long t;
        hipMemset(c, 0, sizeof(Ctl));
        hipLaunchKernelGGL(persist_k<8>, dim3(8 * G), dim3(256), 0, st, c, G, 2000, chain, p, out, sink);
        hipStreamSynchronize(st); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
        Ctl hc; hipMemcpy(&hc, c, sizeof hc, hipMemcpyDeviceToHost);
        printf("persistent, %3d workgroups of ONE XCD, chain %d: %.3f us per round (err %d)\n", G, chain, t / 100.0 / 2000, hc.err);
        hipMemset(c, 0, sizeof(Ctl));
        hipLaunchKernelGGL(persist_k<1>, dim3(G), dim3(256), 0, st, c, G, 2000, chain, p, out, sink);
        hipStreamSynchronize(st); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hc, c, sizeof hc, hipMemcpyDeviceToHost);
        printf("persistent, %3d workgroups over ALL XCDs, chain %d: %.3f us per round (err %d)\n", G, chain, t / 100.0 / 2000, hc.err);
    }
    return 0;
}
