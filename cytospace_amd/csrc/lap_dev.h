// lap_dev.h -- device helpers shared by the LAP kernels (lap_jv.hip: the chain solver; lap_wide.hip: the wide solver):
// order-preserving float keys, DPP wave reductions, the LDS-only barrier, the row-cache constants.
#pragma once
#include "cyto_common.h"
#include <math.h>

namespace cyto {

constexpr int KC = 64;             // cache slots per row (one per lane of a wave)
constexpr int KCU = 63;            // usable entries; slot 63 = { COLSENT, floor }
constexpr uint64_t KEYMAX = ~0ull;
constexpr uint32_t COLSENT = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t f2ord(float h) {
    const uint32_t b = __float_as_uint(h + 0.0f);  // +0.0f: -0 -> +0 so that equal floats get equal keys
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}
__device__ __forceinline__ uint64_t mkkey(float h, uint32_t lowbits) { return ((uint64_t)f2ord(h) << 32) | lowbits; }
__device__ __forceinline__ float key_val(uint64_t k) { return ord2f((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a < b ? b : a; }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

struct K2 { uint64_t m1, m2; };  // two smallest keys of a set of DISTINCT keys (or KEYMAX)
__device__ __forceinline__ void k2_push(K2 &t, uint64_t k) {
    const uint64_t lo = umin64(t.m1, k), hi = umax64(t.m1, k);
    t.m1 = lo; t.m2 = umin64(t.m2, hi);
}
__device__ __forceinline__ void k2_merge(K2 &a, const K2 &b) {
    const uint64_t lo = umin64(a.m1, b.m1), hi = umax64(a.m1, b.m1);
    a.m2 = umin64(hi, umin64(a.m2, b.m2)); a.m1 = lo;
}

// DPP controls quad_perm[1,0,3,2] (0xB1), quad_perm[2,3,0,1] (0x4E), row_half_mirror (0x141),
// row_mirror (0x140): a butterfly inside each 16-lane row (merged sets are disjoint at every step).
template <int CTRL> __device__ __forceinline__ uint32_t dpp32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ uint64_t dpp64(uint64_t x) {
    return ((uint64_t)dpp32<CTRL>((uint32_t)(x >> 32)) << 32) | dpp32<CTRL>((uint32_t)x);
}
__device__ __forceinline__ uint32_t readlane32(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l) {
    return ((uint64_t)readlane32((uint32_t)(x >> 32), l) << 32) | readlane32((uint32_t)x, l);
}
__device__ __forceinline__ uint32_t row_min_u32(uint32_t x) {
    x = umin32(x, dpp32<0xB1>(x)); x = umin32(x, dpp32<0x4E>(x));
    x = umin32(x, dpp32<0x141>(x)); x = umin32(x, dpp32<0x140>(x));
    return x;
}
// all lanes active; result is wave-uniform (scalar).  (Combining the four row minima with the gfx9 cross-row DPP controls
// row_bcast:15 / row_bcast:31 and one read-lane instead of four read-lanes + scalar mins was measured: 2 % slower.)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
    x = row_min_u32(x);
    return umin32(umin32(readlane32(x, 0), readlane32(x, 16)), umin32(readlane32(x, 32), readlane32(x, 48)));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
    x += dpp32<0xB1>(x); x += dpp32<0x4E>(x); x += dpp32<0x141>(x); x += dpp32<0x140>(x);
    return readlane32(x, 0) + readlane32(x, 16) + readlane32(x, 32) + readlane32(x, 48);
}
template <int CTRL> __device__ __forceinline__ void k2_step(K2 &t) {
    K2 o; o.m1 = dpp64<CTRL>(t.m1); o.m2 = dpp64<CTRL>(t.m2);
    k2_merge(t, o);
}
__device__ __forceinline__ void k2_row_allreduce(K2 &t) {
    k2_step<0xB1>(t); k2_step<0x4E>(t); k2_step<0x141>(t); k2_step<0x140>(t);
}
__device__ __forceinline__ K2 k2_wave_allreduce(K2 t) {
    k2_row_allreduce(t);
    K2 r; r.m1 = readlane64(t.m1, 0); r.m2 = readlane64(t.m2, 0);
#pragma unroll
    for (int row = 1; row < 4; row++) {
        K2 o; o.m1 = readlane64(t.m1, row * 16); o.m2 = readlane64(t.m2, row * 16);
        k2_merge(r, o);
    }
    return r;
}
__device__ __forceinline__ uint64_t min64_row_allreduce(uint64_t x) {
    x = umin64(x, dpp64<0xB1>(x)); x = umin64(x, dpp64<0x4E>(x));
    x = umin64(x, dpp64<0x141>(x)); x = umin64(x, dpp64<0x140>(x));
    return x;
}
__device__ __forceinline__ uint64_t min64_wave_allreduce(uint64_t x) {
    x = min64_row_allreduce(x);
    uint64_t r = readlane64(x, 0);
#pragma unroll
    for (int row = 1; row < 4; row++) r = umin64(r, readlane64(x, row * 16));
    return r;
}

// Barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for its outstanding global
// stores/loads (a plain __syncthreads() carries s_waitcnt vmcnt(0): in the augmentation that made every
// step wait for the owner lane's global store to be acknowledged by HBM).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace cyto
