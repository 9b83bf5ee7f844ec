// Micro-benchmark: how long does ONE workgroup need to fetch one (dependent) cost row?
// One 512-thread workgroup reads rows of `n` floats from an n x n matrix; the next row index depends on the data just read
// (a pointer chase through rows, like the augmentation's picks).  K rows are in flight per step (K = 1, 2, 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int CH, int K>
__global__ __launch_bounds__(512) void chase(const float* cost, int n, int steps, int* out, long long* cyc) {
    __shared__ int s_next[8];
    const int tid = threadIdx.x;
    int row[K];
    for (int k = 0; k < K; k++) row[k] = (k * 7919 + 13) % n;
    float acc = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s++) {
        float4 x[K][CH];
        for (int k = 0; k < K; k++) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cost + (long long)row[k] * n), 0, n * 4, 0x00020000);
#pragma unroll
            for (int m = 0; m < CH; m++) {
                const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, m * 512 * 16, 0);
                x[k][m] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
            }
        }
        float m0 = 0.f;
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int m = 0; m < CH; m++) m0 += x[k][m].x + x[k][m].y + x[k][m].z + x[k][m].w;
        acc += m0;
        // next rows: depend on the loaded data (all waves agree via LDS + barrier, like the pick)
        if ((tid & 63) == 0) s_next[tid >> 6] = (int)(__float_as_uint(m0) & 0xFFFF);
        __syncthreads();
        int nx = 0;
        for (int w = 0; w < 8; w++) nx += s_next[w];
        __syncthreads();
        for (int k = 0; k < K; k++) row[k] = (unsigned)(nx * 2654435761u + row[k] * 40503u + k * 977u + s) % (unsigned)n;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) { out[0] = (int)acc; cyc[0] = t1 - t0; }
}
template <int CH, int K> void run(const float* d, int n, int steps, int* dout, long long* dcyc, const char* tag) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chase<CH, K>), dim3(1), dim3(512), 0, 0, d, n, 200, dout, dcyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chase<CH, K>), dim3(1), dim3(512), 0, 0, d, n, steps, dout, dcyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s n=%d rows in flight %d: %.3f us per step, %.3f us per row, %.1f GB/s\n", tag, n, K, ms * 1e3 / steps, ms * 1e3 / steps / K, (double)K * n * 4 / (ms * 1e3 / steps) / 1e3);
}
int main() {
    int* dout; long long* dcyc; hipMalloc(&dout, 64); hipMalloc(&dcyc, 64);
    for (int n : {1024, 5000, 10000, 20000}) {
        float* d; size_t bytes = (size_t)n * n * 4; hipMalloc(&d, bytes);
        std::vector<float> h((size_t)n * n); for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8) * 1e-9f;
        hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        const int steps = 20000;
        if (n <= 2048) { run<1, 1>(d, n, steps, dout, dcyc, "L2-resident"); run<1, 4>(d, n, steps, dout, dcyc, "L2-resident"); }
        else if (n <= 5120) { run<3, 1>(d, n, steps, dout, dcyc, "HBM"); run<3, 2>(d, n, steps, dout, dcyc, "HBM"); run<3, 4>(d, n, steps, dout, dcyc, "HBM"); }
        else if (n <= 10240) { run<5, 1>(d, n, steps, dout, dcyc, "HBM"); run<5, 2>(d, n, steps, dout, dcyc, "HBM"); run<5, 4>(d, n, steps, dout, dcyc, "HBM"); }
        else { run<10, 1>(d, n, steps, dout, dcyc, "HBM"); run<10, 2>(d, n, steps, dout, dcyc, "HBM"); }
        hipFree(d);
    }
    return 0;
}
