// Microbenchmark: what the fp32 matrix pipe delivers for the cost GEMM's inner-loop shape (v_mfma_f32_32x32x2_f32, four
// independent accumulators per wave, 256 threads x 2 blocks per CU), adding the kernel's per-tile ingredients one at a time:
// F = two-level fold, B = workgroup barrier per tile, G = global loads (8 float4 per thread per tile), S = LDS staging stores.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/dbg/mfma_peak tools/dbg/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, int B, int G, int S>
__global__ __launch_bounds__(256, 2) void k(int iters, float *out, const float *__restrict__ src, int64_t ld) {
    __shared__ __attribute__((aligned(16))) float As[2][32][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 128; i += 256) { (&As[0][0][0])[i] = 1.0f; (&Bs[0][0][0])[i] = 0.5f; }
    __syncthreads();
    f32x16 acc[2][2], sum[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) { acc[a][b][r] = 0.f; sum[a][b][r] = 0.f; }
    const int li = lane & 31, lk = lane >> 5;
    const int ao = (wave >> 1) * 64 + li, bo = (wave & 1) * 64 + li;
    const int st_k = tid >> 5, st_c = (tid & 31) * 4;
    const float *g = src + (int64_t)st_k * ld + (blockIdx.x % 300) * 128 + st_c;
    float4 r0, r1, r2, r3, r4, r5, r6, r7;
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = make_float4(1.f, 1.f, 1.f, 1.f);
    for (int it = 0; it < iters; it++) {
        const int buf = it & 1;
        if (G) {
            const float *p = g + (int64_t)(it % 600) * 32 * ld;
            r0 = *(const float4 *)(p); r1 = *(const float4 *)(p + 8 * ld); r2 = *(const float4 *)(p + 16 * ld); r3 = *(const float4 *)(p + 24 * ld);
            r4 = *(const float4 *)(p + 64); r5 = *(const float4 *)(p + 8 * ld + 64); r6 = *(const float4 *)(p + 16 * ld + 64); r7 = *(const float4 *)(p + 24 * ld + 64);
        }
        float a0 = As[buf][lk][ao], a1 = As[buf][lk][ao + 32], b0 = Bs[buf][lk][bo], b1 = Bs[buf][lk][bo + 32];
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (kk + 2 < 32) { na0 = As[buf][kk + 2 + lk][ao]; na1 = As[buf][kk + 2 + lk][ao + 32]; nb0 = Bs[buf][kk + 2 + lk][bo]; nb1 = Bs[buf][kk + 2 + lk][bo + 32]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        if (F) {
            for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { sum[a][b] += acc[a][b]; for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f; }
        }
        if (S) {
            *(float4 *)&As[buf ^ 1][st_k][st_c] = r0; *(float4 *)&As[buf ^ 1][st_k + 8][st_c] = r1; *(float4 *)&As[buf ^ 1][st_k + 16][st_c] = r2; *(float4 *)&As[buf ^ 1][st_k + 24][st_c] = r3;
            *(float4 *)&Bs[buf ^ 1][st_k][st_c] = r4; *(float4 *)&Bs[buf ^ 1][st_k + 8][st_c] = r5; *(float4 *)&Bs[buf ^ 1][st_k + 16][st_c] = r6; *(float4 *)&Bs[buf ^ 1][st_k + 24][st_c] = r7;
        }
        if (B) __syncthreads();
    }
    float s = r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) s += acc[a][b][r] + sum[a][b][r];
    if (s == 123.456f) out[0] = s;
}


// the restructured tile: fold of the previous tile interleaved with the first MFMA step (C = 0), LDS staging stores spread
// over the second half of the tile's MFMA steps
template <int DEEP>
__global__ __launch_bounds__(256, 2) void k2(int iters, float *out, const float *__restrict__ src, int64_t ld) {
    __shared__ __attribute__((aligned(16))) float As[2][32][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 128; i += 256) { (&As[0][0][0])[i] = 1.0f; (&Bs[0][0][0])[i] = 0.5f; }
    __syncthreads();
    f32x16 acc[2][2], sum[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) { acc[a][b][r] = 0.f; sum[a][b][r] = 0.f; }
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int li = lane & 31, lk = lane >> 5;
    const int ao = (wave >> 1) * 64 + li, bo = (wave & 1) * 64 + li;
    const int st_k = tid >> 5, st_c = (tid & 31) * 4;
    const float *g = src + (int64_t)st_k * ld + (blockIdx.x % 300) * 128 + st_c;
    float4 r0, r1, r2, r3, r4, r5, r6, r7;
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = make_float4(1.f, 1.f, 1.f, 1.f);
    for (int it = 0; it < iters; it++) {
        const int buf = it & 1;
        float a0 = As[buf][lk][ao], a1 = As[buf][lk][ao + 32], b0 = Bs[buf][lk][bo], b1 = Bs[buf][lk][bo + 32];
        float na0 = As[buf][2 + lk][ao], na1 = As[buf][2 + lk][ao + 32], nb0 = Bs[buf][2 + lk][bo], nb1 = Bs[buf][2 + lk][bo + 32];
        if (!(DEEP & 4) && it + 1 < iters) {      // (under a run-time condition like the product kernel: otherwise the loads are sunk to their uses)
            const float *p = g + (int64_t)(it % 600) * 32 * ld;
            r0 = *(const float4 *)(p); r1 = *(const float4 *)(p + 8 * ld); r2 = *(const float4 *)(p + 16 * ld); r3 = *(const float4 *)(p + 24 * ld);
            r4 = *(const float4 *)(p + 64); r5 = *(const float4 *)(p + 8 * ld + 64); r6 = *(const float4 *)(p + 16 * ld + 64); r7 = *(const float4 *)(p + 24 * ld + 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(DEEP & 2)) sum[0][0] += acc[0][0];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DEEP & 2)) sum[0][1] += acc[0][1];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DEEP & 2)) sum[1][0] += acc[1][0];
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DEEP & 2)) sum[1][1] += acc[1][1];
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, z, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
#pragma unroll
        for (int kk = 2; kk < 32; kk += 2) {
            na0 = 0.f; na1 = 0.f; nb0 = 0.f; nb1 = 0.f;
            if (kk + 2 < 32) { na0 = As[buf][kk + 2 + lk][ao]; na1 = As[buf][kk + 2 + lk][ao + 32]; nb0 = Bs[buf][kk + 2 + lk][bo]; nb1 = Bs[buf][kk + 2 + lk][bo + 32]; }
            if (!(DEEP & 1)) {
            if (kk == 14) *(float4 *)&As[buf ^ 1][st_k][st_c] = r0;
            if (kk == 16) *(float4 *)&As[buf ^ 1][st_k + 8][st_c] = r1;
            if (kk == 18) *(float4 *)&As[buf ^ 1][st_k + 16][st_c] = r2;
            if (kk == 20) *(float4 *)&As[buf ^ 1][st_k + 24][st_c] = r3;
            if (kk == 22) *(float4 *)&Bs[buf ^ 1][st_k][st_c] = r4;
            if (kk == 24) *(float4 *)&Bs[buf ^ 1][st_k + 8][st_c] = r5;
            if (kk == 26) *(float4 *)&Bs[buf ^ 1][st_k + 16][st_c] = r6;
            if (kk == 28) *(float4 *)&Bs[buf ^ 1][st_k + 24][st_c] = r7;
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) s += acc[a][b][r] + sum[a][b][r];
    if (s == 123.456f) out[0] = s;
}

template <int DEEP>
void run2(const float *src, int64_t ld, const char *what) {
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = (DEEP & 8) ? 512 : 512 * 30, iters = (DEEP & 8) ? 625 * 30 : 625;
    hipLaunchKernelGGL((k2<DEEP>), dim3(512), dim3(256), 0, 0, 10, out, src, ld);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k2<DEEP>), dim3(grid), dim3(256), 0, 0, iters, out, src, ld);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 64.0 * (32 * 32 * 2 * 2);
    printf("%-34s %.2f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    (void)hipFree(out);
}

template <int XT>
__global__ __launch_bounds__(256, 2) void k3(int iters, float *out) {
    __shared__ __attribute__((aligned(16))) float As[2][32][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 128; i += 256) { (&As[0][0][0])[i] = 1.0f; (&Bs[0][0][0])[i] = 0.5f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    const int ao = (wave >> 1) * 64 + li, bo = (wave & 1) * 64 + li;
    float a0 = As[0][lk][ao], a1 = As[0][lk][ao + 32], b0 = Bs[0][lk][bo], b1 = Bs[0][lk][bo + 32];
    float na0 = As[0][2 + lk][ao], na1 = As[0][2 + lk][ao + 32], nb0 = Bs[0][2 + lk][bo], nb1 = Bs[0][2 + lk][bo + 32];
    for (int it = 0; it < iters; it++) {
        const int buf = it & 1;
        if (!XT) {
            a0 = As[buf][lk][ao]; a1 = As[buf][lk][ao + 32]; b0 = Bs[buf][lk][bo]; b1 = Bs[buf][lk][bo + 32];
            na0 = As[buf][2 + lk][ao]; na1 = As[buf][2 + lk][ao + 32]; nb0 = Bs[buf][2 + lk][bo]; nb1 = Bs[buf][2 + lk][bo + 32];
        }
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            float ma0, ma1, mb0, mb1;
            if (kk + 4 < 32) { ma0 = As[buf][kk + 4 + lk][ao]; ma1 = As[buf][kk + 4 + lk][ao + 32]; mb0 = Bs[buf][kk + 4 + lk][bo]; mb1 = Bs[buf][kk + 4 + lk][bo + 32]; }
            else if (XT) { ma0 = As[buf ^ 1][kk + 4 - 32 + lk][ao]; ma1 = As[buf ^ 1][kk + 4 - 32 + lk][ao + 32]; mb0 = Bs[buf ^ 1][kk + 4 - 32 + lk][bo]; mb1 = Bs[buf ^ 1][kk + 4 - 32 + lk][bo + 32]; }
            else { ma0 = ma1 = mb0 = mb1 = 0.f; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            na0 = ma0; na1 = ma1; nb0 = mb0; nb1 = mb1;
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) s += acc[a][b][r];
    if (s == 123.456f) out[0] = s;
}
template <int XT>
void run3(const char *what) {
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 512 * 30, iters = 625;
    hipLaunchKernelGGL((k3<XT>), dim3(512), dim3(256), 0, 0, 10, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k3<XT>), dim3(grid), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 64.0 * (32 * 32 * 2 * 2);
    printf("%-34s %.2f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    (void)hipFree(out);
}

// v1-style steady loop (no tiles, no barrier): MODE 0 = all waves read the same LDS addresses; 1 = per-wave fragments (ao / bo);
// 2 = per-wave + alternating buffer per iteration
template <int MODE>
__global__ __launch_bounds__(256, 2) void k4(int iters, float *out) {
    __shared__ __attribute__((aligned(16))) float As[2][32][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 128; i += 256) { (&As[0][0][0])[i] = 1.0f; (&Bs[0][0][0])[i] = 0.5f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    const int ao = MODE ? (wave >> 1) * 64 + li : li, bo = MODE ? (wave & 1) * 64 + li : li;
    float a0 = 1.f, a1 = 1.f, b0 = .5f, b1 = .5f;
    for (int it = 0; it < iters; it++) {
        const int buf = MODE == 2 ? (it & 1) : 0;
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            float na0 = As[buf][(kk + lk) & 31][ao], na1 = As[buf][(kk + lk) & 31][ao + 32], nb0 = Bs[buf][(kk + lk) & 31][bo], nb1 = Bs[buf][(kk + lk) & 31][bo + 32];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 16; r++) s += acc[a][b][r];
    if (s == 123.456f) out[0] = s;
}
template <int MODE>
void run4(const char *what) {
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 512 * 30, iters = 625;
    hipLaunchKernelGGL((k4<MODE>), dim3(512), dim3(256), 0, 0, 10, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k4<MODE>), dim3(grid), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 64.0 * (32 * 32 * 2 * 2);
    printf("%-34s %.2f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    (void)hipFree(out);
}

template <int F, int B, int G, int S>
void run(const float *src, int64_t ld, const char *what) {
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 512 * 30, iters = 625;                   // the c3 GEMM: 15 640 workgroups, 625 tiles each
    hipLaunchKernelGGL((k<F, B, G, S>), dim3(512), dim3(256), 0, 0, 10, out, src, ld);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<F, B, G, S>), dim3(grid), dim3(256), 0, 0, iters, out, src, ld);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 64.0 * (32 * 32 * 2 * 2);
    printf("%-34s %.2f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    (void)hipFree(out);
}

int main() {
    const int64_t ld = 300 * 128;
    float *src; (void)hipMalloc(&src, ld * 600 * 32 * 4 + (1 << 20)); (void)hipMemset(src, 0, ld * 600 * 32 * 4);
    run<0, 0, 0, 0>(src, ld, "MFMA + LDS fragment reads");
    run<1, 0, 0, 0>(src, ld, "+ fold");
    run<0, 1, 0, 0>(src, ld, "+ barrier");
    run<1, 1, 0, 0>(src, ld, "+ fold + barrier");
    run<0, 1, 0, 1>(src, ld, "+ barrier + LDS stores");
    run<0, 0, 1, 0>(src, ld, "+ global loads");
    run<0, 1, 1, 1>(src, ld, "+ barrier + loads + stores");
    run<1, 1, 1, 1>(src, ld, "everything");
    run4<0>("steady loop, shared fragments");
    run4<1>("steady loop, per-wave fragments");
    run4<2>("steady loop, per-wave, 2 buffers");
    run3<0>("tiles + barrier, reads 2 steps ahead");
    run3<1>("  same, reads cross the tile end");
    run2<0>(src, ld, "everything, restructured");
    run2<1>(src, ld, "restructured, no LDS stores");
    run2<2>(src, ld, "restructured, no fold");
    run2<4>(src, ld, "restructured, no global loads");
    run2<7>(src, ld, "restructured, none of the three");
    run2<8>(src, ld, "restructured, 512 persistent blocks");
    run2<15>(src, ld, "none of the three, persistent");
    return 0;
}
