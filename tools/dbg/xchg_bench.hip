// Microbenchmark: all-to-all exchange latency between W single-wave workgroups through L2
// (data-tagged 16-byte records, double-buffered by round parity).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st64(uint64_t *p, uint64_t x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// slots: [2][64][4] uint64 (32 B per worker)
__global__ __launch_bounds__(64) void xchg(uint64_t *slots, int W, int stride, int rounds, int *xcc_out, long long *cycles_out, int *abort_flag,
                                           const float *junk, int junk_stride_floats, float *sink) {
    if (blockIdx.x % stride) return;
    const int w = blockIdx.x / stride;
    if (w >= W) return;
    const int lane = threadIdx.x;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 0) xcc_out[w] = (int)(xcc & 0xF);
    uint64_t acc = 0;
    float facc = 0.f;
    const long long t0 = wall_clock64();
    for (int t = 0; t < rounds; t++) {
        const uint32_t tag = (uint32_t)t & 0xFF;
        uint64_t *my = slots + ((size_t)(t & 1) * 64 + w) * 4;
        // optional dependent HBM/L2 fetch, emulating the row slice read (address depends on previous winner)
        if (junk) {
            const size_t row = (size_t)((acc >> 8) % 4096);
            facc += junk[row * junk_stride_floats + w * 256 + lane * 4];
        }
        const uint64_t a = ((uint64_t)(uint32_t)((w * 2654435761u + t * 40503u) >> 4) << 32) | ((uint32_t)w << 8) | tag;
        if (lane == 0) { st64(my, a); st64(my + 1, ((uint64_t)t << 32) | tag); }
        const uint64_t *peer = slots + ((size_t)(t & 1) * 64 + lane) * 4;
        uint64_t pa = ~0ull, pb = 0;
        long long spin0 = wall_clock64();
        for (;;) {
            bool ok = true;
            if (lane < W) {
                pa = ld64(peer); pb = ld64(peer + 1);
                ok = ((uint32_t)pa & 0xFF) == tag && ((uint32_t)pb & 0xFF) == tag;
            }
            if (__ballot(ok) == ~0ull) break;
            if (wall_clock64() - spin0 > 100000000LL) { *abort_flag = 1; return; }   // 1 s at 100 MHz
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        }
        // wave min over the keys
        uint64_t m = pa;
        for (int off = 32; off >= 1; off >>= 1) { const uint64_t o = __shfl_xor(m, off); m = o < m ? o : m; }
        acc += m;
    }
    const long long t1 = wall_clock64();
    if (lane == 0) { cycles_out[w] = t1 - t0; sink[w] = facc + (float)acc; }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    uint64_t *slots; int *xcc, *abortf; long long *cyc; float *junk, *sink;
    CK(hipMalloc(&slots, 2 * 64 * 32));
    CK(hipMalloc(&xcc, 64 * 4)); CK(hipMalloc(&abortf, 4)); CK(hipMalloc(&cyc, 64 * 8)); CK(hipMalloc(&sink, 64 * 4));
    const size_t junk_floats = (size_t)4096 * 20480;
    CK(hipMalloc(&junk, junk_floats * 4)); CK(hipMemset(junk, 0, junk_floats * 4));
    for (int mode = 0; mode < 2; mode++)
    for (int stride : {8, 1})
        for (int W : {2, 8, 16, 32, 64}) {
            CK(hipMemset(slots, 0xFF, 2 * 64 * 32)); CK(hipMemset(abortf, 0, 4));
            hipLaunchKernelGGL(xchg, dim3(W * stride), dim3(64), 0, 0, slots, W, stride, rounds, xcc, cyc, abortf, mode ? junk : nullptr, 20480, sink);
            CK(hipDeviceSynchronize());
            std::vector<int> hx(64); std::vector<long long> hc(64); int ha = 0;
            CK(hipMemcpy(hx.data(), xcc, 64 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), cyc, 64 * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&ha, abortf, 4, hipMemcpyDeviceToHost));
            int xm = 0; for (int i = 0; i < W; i++) xm |= 1 << hx[i];
            printf("mode=%s stride=%d W=%2d: %.3f us/round  (xcc mask 0x%02x, abort=%d)\n", mode ? "xchg+dependent-load" : "xchg-only", stride, W,
                   (double)hc[0] / 100.0 / rounds, xm, ha);
        }
    return 0;
}
