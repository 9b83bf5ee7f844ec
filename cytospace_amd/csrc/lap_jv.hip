// lap_jv.hip -- Jonker-Volgenant LAP solver for gfx950 (MI355X), hand-written HIP.
//
// Replaces the reference's single third-party call
//     `_, y, _ = lapjv.lapjv(cost_scaled)`
// (/root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:34-40).
// Same four phases, same operand order and the same lowest-index tie-breaking as the CPU
// oracle (oracle/jv_oracle_impl.h), so rowsol/colsol/u/v are bit-identical to it.
//
// Kernel plan (float32 path; DESIGN.md section 4.1 has the exactness arguments and the measurements)
//   colred_partial / colred_finish / colred_assign : COLUMN REDUCTION.  The only phase with N^2 independent
//       work: a full-chip streaming pass (float4 per lane, >= 2k workgroups).
//   rows_same_as_prev / rows_group_ids : runs of bitwise identical rows (CytoSPACE repeats every spot row
//       slots[s] times); used to elide provably useless scans in the augmentation.
//   build_row_caches<CH> / build_row_caches_stream : per row, the <= 63 columns with the smallest reduced cost
//       plus a floor that bounds every other column (a certificate while prices only decrease).
//   jv_chain2<CH, LDS_STATE> : REDUCTION TRANSFER + AUGMENTING ROW REDUCTION, a chain of dependent row scans.
//       One persistent 512-thread workgroup; a scan is served from the row's cache by wave 0 alone whenever the
//       cache certifies its top-2, otherwise the workgroup re-scans the row (refresh_row / refresh_row_stream).
//   jv_aug_lazy<LDS_STATE> (n > 5120) : AUGMENTATION with cache-certified scans and sparse search initialisation.
//   jv_aug2<CH, LDS_STATE> (n <= 5120, and taking over from jv_aug_lazy on deep searches) : dense augmentation,
//       per-column search state in registers.
//   jv_chain<T, CH> : the generic dense chain (1024 threads) -- float64 only.
//
// Compile with -ffp-contract=off (build.py does): the arithmetic is subtract/compare only,
// but nothing may be re-associated.
#include "cyto_common.h"
#include "lap_dev.h"
#include "lap_wide.h"
#include <math.h>
#include <type_traits>
#include <vector>
#include <algorithm>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <atomic>

namespace cyto {

constexpr int BLOCK = 1024;
constexpr int NW = BLOCK / 64;

// counters written by jv_chain (index into ChainState::counters)
enum { C_RT = 0, C_ARR, C_AUG_INIT, C_AUG_RELAX, C_AUGS, C_HOPS, C_FREE_CR, C_FREE_A1, C_FREE_A2, C_ROWS_READ, C_NCOUNTERS };

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; static constexpr int W = 4; };
template <> struct VecOf<double> { using type = double2; static constexpr int W = 2; };

// Row indirection (first half): CytoSPACE repeats every spot row slots[s] times
// (linear_assignment_solvers.py:63-66).  With a row map the cost holds every DISTINCT row once (S_u x C) and LAP row i reads
// row rowmap[i]; without one (nullptr) row i is row i.  RBASE gives callees that compute `base + i * ld` themselves a base that
// lands on the mapped row.
__device__ __forceinline__ int64_t row_off(const int32_t *__restrict__ rowmap, int i, int64_t ld) {
    return (int64_t)(rowmap ? rowmap[i] : i) * ld;
}
#define RBASE(cost_, rowmap_, i_, ld_) ((cost_) + (row_off((rowmap_), (i_), (ld_)) - (int64_t)(i_) * (ld_)))

template <typename T> __device__ __forceinline__ T vec_get(const typename VecOf<T>::type &x, int e);
template <> __device__ __forceinline__ float vec_get<float>(const float4 &x, int e) {
    return e == 0 ? x.x : e == 1 ? x.y : e == 2 ? x.z : x.w;
}
template <> __device__ __forceinline__ double vec_get<double>(const double2 &x, int e) { return e == 0 ? x.x : x.y; }

__device__ __forceinline__ bool memcmp_neq(float a, float b) { return __float_as_uint(a) != __float_as_uint(b); }
__device__ __forceinline__ bool memcmp_neq(double a, double b) { return __double_as_longlong(a) != __double_as_longlong(b); }

// ------------------------------------------------------------------------------------------
// COLUMN REDUCTION
// ------------------------------------------------------------------------------------------
// Partial column minima over a block of rows.  Lane owns VW consecutive columns; each row is one
// 16-byte load per lane (a wave reads 1 KiB contiguous).  Strict '<' while rows ascend keeps the
// lowest row index on ties, exactly like the oracle's row-wise sweep.
template <typename T>
__global__ __launch_bounds__(256) void colred_partial(int n, int64_t ld, const T *__restrict__ cost,
                                                      int rows_per_block, T *__restrict__ pmin,
                                                      int32_t *__restrict__ parg, int *__restrict__ nonfinite,
                                                      int nrows, const int32_t *__restrict__ ulist,
                                                      const int32_t *__restrict__ ufirst) {
    // nrows rows are swept (n without a row map).  With one: the k-th DISTINCT row that is in use is row ulist[k] of the
    // cost and stands for the LAP rows starting at ufirst[k] (ascending in k) -- the lowest of them is what a tie keeps.
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    const int q = blockIdx.x * 256 + threadIdx.x;  // vector-column index
    const int col0 = q * VW;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(nrows, r0 + rows_per_block);
    if (col0 >= n) return;
    T mn[VW];
    int32_t arg[VW];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < VW; e++) { mn[e] = (T)INFINITY; arg[e] = ufirst ? ufirst[min(r0, nrows - 1)] : r0; }
    const int64_t stride = ld / VW;
#pragma unroll 4
    for (int r = r0; r < r1; r++) {
        const V x = *(reinterpret_cast<const V *>(cost + (int64_t)(ulist ? ulist[r] : r) * ld) + q);
        const int32_t rid = ufirst ? ufirst[r] : r;
#pragma unroll
        for (int e = 0; e < VW; e++) {
            const T xe = vec_get<T>(x, e);
            if (col0 + e < n) bad |= !__builtin_isfinite(xe);
            if (xe < mn[e]) { mn[e] = xe; arg[e] = rid; }
        }
    }
    (void)stride;
#pragma unroll
    for (int e = 0; e < VW; e++) {
        if (col0 + e < n) {
            pmin[(int64_t)blockIdx.y * n + col0 + e] = mn[e];
            parg[(int64_t)blockIdx.y * n + col0 + e] = arg[e];
        }
    }
    if (bad) atomicOr(nonfinite, 1);
}

// Combine the row blocks in ascending order; v[j] = column minimum, imin[j] = its lowest row.
// "Columns are claimed from the last to the first, the first claim of a row wins":
// rowsol[i] = max{ j : imin[j] == i }, matches[i] = #{ j : imin[j] == i }.
template <typename T>
__global__ void colred_finish(int n, int nblocks, const T *__restrict__ pmin, const int32_t *__restrict__ parg,
                              T *__restrict__ v, int32_t *__restrict__ imin, int32_t *rowsol, int32_t *matches) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    T mn = pmin[j];
    int32_t arg = parg[j];
    for (int b = 1; b < nblocks; b++) {
        const T x = pmin[(int64_t)b * n + j];
        if (x < mn) { mn = x; arg = parg[(int64_t)b * n + j]; }
    }
    v[j] = mn;
    imin[j] = arg;
    atomicMax(&rowsol[arg], j);
    atomicAdd(&matches[arg], 1);
}

__global__ void colred_assign(int n, const int32_t *__restrict__ imin, const int32_t *__restrict__ rowsol,
                              int32_t *__restrict__ colsol) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = imin[j];
    colsol[j] = (rowsol[i] == j) ? i : -1;
}

// ------------------------------------------------------------------------------------------
// Duplicate-row groups.  CytoSPACE repeats every spot row slots[s] times
// (linear_assignment_solvers.py:63-66), so the LAP often has runs of bitwise identical rows.
// same_prev[i] = 1 iff row i equals row i-1 bit for bit (early exit on the first differing 4 KiB:
// distinct rows cost one chunk, identical rows one full read).  Used by the augmentation to skip
// scans that provably change nothing (see chain_augment).
// ------------------------------------------------------------------------------------------
// with a row map: consecutive LAP rows that read the same stored row
__global__ __launch_bounds__(256) void rows_same_from_map(int n, const int32_t *__restrict__ rowmap, int32_t *__restrict__ same_prev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) same_prev[i] = (i > 0 && rowmap[i] == rowmap[i - 1]) ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void rows_same_as_prev(int n, int64_t ld, const T *__restrict__ cost, int32_t *__restrict__ same_prev) {
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    __shared__ int s_diff;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        if (i == 0) { if (threadIdx.x == 0) same_prev[0] = 0; continue; }
        const V *a = reinterpret_cast<const V *>(cost + (int64_t)i * ld);
        const V *b = reinterpret_cast<const V *>(cost + (int64_t)(i - 1) * ld);
        const int nv = (n + VW - 1) / VW;
        int diff = 0;
        for (int q0 = 0; q0 < nv; q0 += 256) {
            const int q = q0 + threadIdx.x;
            bool d = false;
            if (q < nv) {
                const V x = a[q], y = b[q];
#pragma unroll
                for (int e = 0; e < VW; e++)
                    if (q * VW + e < n) d |= memcmp_neq(vec_get<T>(x, e), vec_get<T>(y, e));
            }
            if (threadIdx.x == 0) s_diff = 0;
            __syncthreads();
            if (d) s_diff = 1;
            __syncthreads();
            diff = s_diff;
            __syncthreads();
            if (diff) break;
        }
        if (threadIdx.x == 0) same_prev[i] = diff ? 0 : 1;
    }
}

// gid[i] = number of group starts in rows 0..i, minus 1; *ngroups = number of groups (one workgroup)
__global__ __launch_bounds__(1024) void rows_group_ids(int n, const int32_t *__restrict__ same_prev, int32_t *__restrict__ gid,
                                                       int *__restrict__ ngroups) {
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int R = (n + 1023) / 1024;
    const int r0 = min(n, tid * R), r1 = min(n, r0 + R);
    int c = 0;
    for (int i = r0; i < r1; i++) c += same_prev[i] ? 0 : 1;
    s_part[tid] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int y = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += y;
        __syncthreads();
    }
    int run = s_part[tid] - c;
    for (int i = r0; i < r1; i++) { run += same_prev[i] ? 0 : 1; gid[i] = run - 1; }
    if (tid == 1023) *ngroups = s_part[1023];
}

// ------------------------------------------------------------------------------------------
// Workgroup-wide reductions.  Values are reduced with wave64 shuffles, then the 16 wave
// partials go through LDS once; every lane ends up with the result, so all threads can run
// the (identical) scalar control logic without a broadcast.  `par` alternates 0/1 between
// consecutive reductions so one barrier per reduction suffices.
// ------------------------------------------------------------------------------------------
template <typename T> struct Top2 {
    T u1; uint32_t k1; T a1;  // lexicographic minimum (value, key) and an attached payload
    T u2; uint32_t k2;        // lexicographic minimum excluding k1
};

template <typename T> __device__ __forceinline__ bool lex_less(T a, uint32_t ka, T b, uint32_t kb) {
    return (a < b) || (a == b && ka < kb);
}

template <typename T> __device__ __forceinline__ void top2_push(Top2<T> &t, T h, uint32_t key, T aux) {
    if (lex_less(h, key, t.u1, t.k1)) {
        t.u2 = t.u1; t.k2 = t.k1;
        t.u1 = h; t.k1 = key; t.a1 = aux;
    } else if (lex_less(h, key, t.u2, t.k2)) {
        t.u2 = h; t.k2 = key;
    }
}

template <typename T> __device__ __forceinline__ void top2_merge(Top2<T> &a, const Top2<T> &b) {
    if (lex_less(b.u1, b.k1, a.u1, a.k1)) {
        T nu2; uint32_t nk2;
        if (lex_less(a.u1, a.k1, b.u2, b.k2)) { nu2 = a.u1; nk2 = a.k1; } else { nu2 = b.u2; nk2 = b.k2; }
        a.u1 = b.u1; a.k1 = b.k1; a.a1 = b.a1; a.u2 = nu2; a.k2 = nk2;
    } else if (lex_less(b.u1, b.k1, a.u2, a.k2)) {
        a.u2 = b.u1; a.k2 = b.k1;
    }
}

template <typename T> struct RedScratch {
    T u1[2][NW]; T a1[2][NW]; T u2[2][NW];
    uint32_t k1[2][NW]; uint32_t k2[2][NW];
    int scan[2][NW];
};

template <typename T> __device__ __forceinline__ Top2<T> top2_shfl_xor(const Top2<T> &t, int off) {
    Top2<T> o;
    o.u1 = __shfl_xor(t.u1, off); o.k1 = __shfl_xor(t.k1, off); o.a1 = __shfl_xor(t.a1, off);
    o.u2 = __shfl_xor(t.u2, off); o.k2 = __shfl_xor(t.k2, off);
    return o;
}

template <typename T> __device__ __forceinline__ Top2<T> wg_top2(Top2<T> t, RedScratch<T> &s, int &par) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) top2_merge(t, top2_shfl_xor(t, off));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { s.u1[par][w] = t.u1; s.k1[par][w] = t.k1; s.a1[par][w] = t.a1; s.u2[par][w] = t.u2; s.k2[par][w] = t.k2; }
    __syncthreads();
    const int l = lane & (NW - 1);
    Top2<T> r;
    r.u1 = s.u1[par][l]; r.k1 = s.k1[par][l]; r.a1 = s.a1[par][l]; r.u2 = s.u2[par][l]; r.k2 = s.k2[par][l];
#pragma unroll
    for (int off = NW / 2; off >= 1; off >>= 1) top2_merge(r, top2_shfl_xor(r, off));
    par ^= 1;
    return r;
}

// lexicographic (val, key) minimum with payload; cheaper than the full top-2
template <typename T> struct Min1 { T u; uint32_t k; T a; };

template <typename T> __device__ __forceinline__ Min1<T> wg_min1(Min1<T> t, RedScratch<T> &s, int &par) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const T ou = __shfl_xor(t.u, off); const uint32_t ok = __shfl_xor(t.k, off); const T oa = __shfl_xor(t.a, off);
        if (lex_less(ou, ok, t.u, t.k)) { t.u = ou; t.k = ok; t.a = oa; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { s.u1[par][w] = t.u; s.k1[par][w] = t.k; s.a1[par][w] = t.a; }
    __syncthreads();
    const int l = lane & (NW - 1);
    Min1<T> r; r.u = s.u1[par][l]; r.k = s.k1[par][l]; r.a = s.a1[par][l];
#pragma unroll
    for (int off = NW / 2; off >= 1; off >>= 1) {
        const T ou = __shfl_xor(r.u, off); const uint32_t ok = __shfl_xor(r.k, off); const T oa = __shfl_xor(r.a, off);
        if (lex_less(ou, ok, r.u, r.k)) { r.u = ou; r.k = ok; r.a = oa; }
    }
    par ^= 1;
    return r;
}

// exclusive prefix sum over the workgroup (thread order); returns offset, *total = sum
template <typename T> __device__ __forceinline__ int wg_exscan(int x, RedScratch<T> &s, int &par, int *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(inc, off);
        if (lane >= off) inc += y;
    }
    if (lane == 63) s.scan[par][w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int c = s.scan[par][i];
        if (i < w) base += c;
        tot += c;
    }
    par ^= 1;
    *total = tot;
    return base + inc - x;
}

// ------------------------------------------------------------------------------------------
// The sequential chain: REDUCTION TRANSFER -> AUGMENTING ROW REDUCTION x2 -> AUGMENTATION.
// ------------------------------------------------------------------------------------------
template <typename T> struct ChainArgs {
    int n;
    int64_t ld;
    const T *cost;
    T *v;                // [n] in: column minima; out: final prices
    T *u;                // [n] out
    int32_t *rowsol;     // [n] in/out
    int32_t *colsol;     // [n] in/out
    int32_t *matches;    // [n] in
    int32_t *freerows;   // [n] scratch
    int32_t *rtrows;     // [n] scratch
    int32_t *pred;       // [n] scratch
    double *total;       // [1] out
    long long *counters; // [C_NCOUNTERS] out
    int *status;         // [1] out (0 ok)
    T *dwork;            // [n] scratch (jv_chain_stream: distances)
    int32_t *lvl;        // [n] scratch (jv_chain_stream: level at which a column was scanned; 0 = not scanned)
    const uint32_t *cache_col;   // [n x 64] jv_chain_stream: row caches (build_row_caches_wide), or nullptr
    const T *cache_val;          // [n x 64]
    int cs_lds;                  // jv_chain_stream: colsol as u16 in LDS during RT / ARR (n <= 65 535)
    int v_lds;                   // jv_chain_stream: the prices in LDS too during RT / ARR (with cs_lds, where both fit)
};

// The persistent chain kernels run one workgroup per PROBLEM: a batch of independent chunk LAPs is one launch with
// grid = batch size (consecutive workgroups land on consecutive XCDs, so the chains spread over the whole chip), each
// workgroup reading its own argument block.  One stream per chunk instead tops out at ~32 concurrent kernels
// (measured: 64 chunks on 64 streams took 3x as long as 32) and every 1-workgroup grid starts on XCD 0.

// agent-scope relaxed accesses: served by L2, never by the scalar cache or a stale L1 line
__device__ __forceinline__ int32_t ld_i32(const int32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_i32(int32_t *p, int32_t x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename T, int CH, bool COLSOL_LDS>
__global__ __launch_bounds__(BLOCK) void jv_chain(ChainArgs<T> a) {
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    constexpr int NC = CH * VW;
    static_assert(NC <= 64, "slot masks are 64-bit");
    extern __shared__ __attribute__((aligned(16))) int32_t s_colsol[];
    __shared__ RedScratch<T> red;

    const int tid = threadIdx.x;
    const int n = a.n;
    const int64_t ld = a.ld;
    const T *__restrict__ cost = a.cost;
    int par = 0;
    const T INF = (T)INFINITY;

    // slot s = m*VW + e  <->  column (m*BLOCK + tid)*VW + e
    auto slot_col = [&](int s) -> int { return ((s / VW) * BLOCK + tid) * VW + (s % VW); };
    uint64_t validm = 0;
#pragma unroll
    for (int s = 0; s < NC; s++) if (slot_col(s) < n) validm |= (1ull << s);

    auto colsol_get = [&](int j) -> int32_t { if constexpr (COLSOL_LDS) return s_colsol[j]; else return ld_i32(a.colsol + j); };
    auto colsol_set = [&](int j, int32_t i) { if constexpr (COLSOL_LDS) s_colsol[j] = i; else st_i32(a.colsol + j, i); };

    // ---- load prices and the assignment produced by the column reduction ----
    T vreg[NC];
    uint64_t assignedm = 0;
#pragma unroll
    for (int s = 0; s < NC; s++) {
        const int c = slot_col(s);
        vreg[s] = 0;
        if (c < n) {
            vreg[s] = a.v[c];
            const int32_t cs = a.colsol[c];
            if (cs >= 0) assignedm |= (1ull << s);
            if constexpr (COLSOL_LDS) s_colsol[c] = cs;
        }
    }

    auto load_row = [&](int i, V (&x)[CH]) {
        const V *rp = reinterpret_cast<const V *>(cost + (int64_t)i * ld);
#pragma unroll
        for (int m = 0; m < CH; m++) {
            const int q = m * BLOCK + tid;
            if (q * VW < n) x[m] = rp[q];
        }
    };
    // set vreg of column j (only its owner thread does anything)
    auto set_v = [&](int j, T val) {
        const int q = j / VW;
        if ((q % BLOCK) == tid) {
            const int s = (q / BLOCK) * VW + (j % VW);
#pragma unroll
            for (int t = 0; t < NC; t++) if (t == s) vreg[t] = val;
        }
    };
    auto set_assigned = [&](int j) {
        const int q = j / VW;
        if ((q % BLOCK) == tid) assignedm |= (1ull << ((q / BLOCK) * VW + (j % VW)));
    };

    long long c_rt = 0, c_arr = 0, c_auginit = 0, c_augrelax = 0, c_augs = 0, c_hops = 0;
    long long c_free_a1 = 0;

    // ---- free-row list (matches == 0) and reduction-transfer list (matches == 1), ascending ----
    int numfree = 0, nrt = 0;
    {
        const int R = (n + BLOCK - 1) / BLOCK;
        const int r0 = min(n, tid * R), r1 = min(n, r0 + R);
        int f = 0, g = 0;
        for (int i = r0; i < r1; i++) { const int mt = a.matches[i]; f += (mt == 0); g += (mt == 1); }
        int of = wg_exscan(f, red, par, &numfree);
        int og = wg_exscan(g, red, par, &nrt);
        for (int i = r0; i < r1; i++) {
            const int mt = a.matches[i];
            if (mt == 0) st_i32(a.freerows + of++, i);
            else if (mt == 1) st_i32(a.rtrows + og++, i);
        }
    }
    __syncthreads();
    const long long c_free_cr = numfree;

    // ---- REDUCTION TRANSFER ----
    if (n > 1) {
        for (int k = 0; k < nrt; k++) {
            const int i = ld_i32(a.rtrows + k);
            const int j1 = ld_i32(a.rowsol + i);
            V x[CH];
            load_row(i, x);
            Min1<T> loc; loc.u = INF; loc.k = 0xFFFFFFFFu; loc.a = 0;
#pragma unroll
            for (int s = 0; s < NC; s++) {
                const int c = slot_col(s);
                if (((validm >> s) & 1) && c != j1) {
                    const T h = vec_get<T>(x[s / VW], s % VW) - vreg[s];
                    if (h < loc.u) loc.u = h;
                }
            }
            loc.k = 0;  // value-only minimum
            const Min1<T> g = wg_min1(loc, red, par);
            // v[j1] = v[j1] - min  (owner only)
            {
                const int q = j1 / VW;
                if ((q % BLOCK) == tid) {
                    const int s = (q / BLOCK) * VW + (j1 % VW);
#pragma unroll
                    for (int t = 0; t < NC; t++) if (t == s) vreg[t] = vreg[t] - g.u;
                }
            }
            c_rt++;
        }
    }

    // ---- AUGMENTING ROW REDUCTION, two sweeps ----
    const long long arr_budget = 1000ll * n + 1000000ll;   // == JV_ARR_BUDGET(n) of the oracle
    for (int sweep = 0; sweep < 2; sweep++) {
        int k = 0;
        const int prev = numfree;
        numfree = 0;
        int carry = -1;
        while (carry >= 0 || k < prev) {
            if (c_arr >= arr_budget) {
                // step budget exhausted (see oracle/jv_oracle_impl.h): hand the waiting rows to the
                // augmentation phase, the displaced row first, then the rest in list order
                if (carry >= 0) { if (tid == 0) st_i32(a.freerows + numfree, carry); numfree++; carry = -1; }
                while (k < prev) { const int r = ld_i32(a.freerows + k); k++; if (tid == 0) st_i32(a.freerows + numfree, r); numfree++; }
                break;
            }
            int i;
            if (carry >= 0) { i = carry; carry = -1; }
            else { i = ld_i32(a.freerows + k); k++; }
            V x[CH];
            load_row(i, x);
            Top2<T> loc; loc.u1 = INF; loc.k1 = 0xFFFFFFFFu; loc.a1 = 0; loc.u2 = INF; loc.k2 = 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s < NC; s++) {
                if ((validm >> s) & 1) {
                    const T h = vec_get<T>(x[s / VW], s % VW) - vreg[s];
                    top2_push(loc, h, (uint32_t)slot_col(s), vreg[s]);
                }
            }
            const Top2<T> g = wg_top2(loc, red, par);
            c_arr++;
            int j1 = (int)g.k1;
            const int j2 = (int)g.k2;
            int i0 = colsol_get(j1);
            const T vj1 = g.a1;
            const T vnew = vj1 - (g.u2 - g.u1);
            const bool lowers = vnew < vj1;
            if (lowers) set_v(j1, vnew);
            else if (i0 >= 0) { j1 = j2; i0 = colsol_get(j2); }
            __syncthreads();  // every wave has read colsol for this step before it changes
            if (tid == 0) { st_i32(a.rowsol + i, j1); colsol_set(j1, i); }
            set_assigned(j1);
            if (i0 >= 0) {
                if (lowers) carry = i0;
                else { if (tid == 0) st_i32(a.freerows + numfree, i0); numfree++; }
            }
        }
        __syncthreads();
        if (sweep == 0) c_free_a1 = numfree;
    }
    const long long c_free_a2 = numfree;

    // ---- AUGMENTATION ----
    int err = 0;
    for (int f = 0; f < numfree && !err; f++) {
        const int freerow = ld_i32(a.freerows + f);
        T dreg[NC];
        uint64_t scannedm = 0, readym = 0;
        {
            V x[CH];
            load_row(freerow, x);
#pragma unroll
            for (int s = 0; s < NC; s++) {
                dreg[s] = INF;
                if ((validm >> s) & 1) {
                    dreg[s] = vec_get<T>(x[s / VW], s % VW) - vreg[s];
                    a.pred[slot_col(s)] = freerow;
                }
            }
            c_auginit++;
        }
        bool have = false;
        T curmin = 0;
        int endofpath = -1;
        for (;;) {
            // pick: lexicographic min of (d, assigned?, column) over unscanned columns
            Min1<T> loc; loc.u = INF; loc.k = 0xFFFFFFFFu; loc.a = 0;
#pragma unroll
            for (int s = 0; s < NC; s++) {
                if (((validm & ~scannedm) >> s) & 1) {
                    const uint32_t key = (uint32_t)slot_col(s) | (((assignedm >> s) & 1) ? 0x80000000u : 0u);
                    if (lex_less(dreg[s], key, loc.u, loc.k)) { loc.u = dreg[s]; loc.k = key; loc.a = vreg[s]; }
                }
            }
            const Min1<T> g = wg_min1(loc, red, par);
            if (g.k == 0xFFFFFFFFu) { err = CYTO_ERR_INTERNAL; break; }
            const int jp = (int)(g.k & 0x7FFFFFFFu);
            if (!have || g.u != curmin) { readym |= scannedm; curmin = g.u; have = true; }
            if (!(g.k & 0x80000000u)) { endofpath = jp; break; }
            // scan column jp through its row
            {
                const int q = jp / VW;
                if ((q % BLOCK) == tid) scannedm |= (1ull << ((q / BLOCK) * VW + (jp % VW)));
            }
            const int i = colsol_get(jp);
            const T cip = cost[(int64_t)i * ld + jp];
            V x[CH];
            load_row(i, x);
            const T h = (cip - g.a) - curmin;
#pragma unroll
            for (int s = 0; s < NC; s++) {
                if (((validm & ~scannedm) >> s) & 1) {
                    const T v2 = (vec_get<T>(x[s / VW], s % VW) - vreg[s]) - h;
                    if (v2 < dreg[s]) { dreg[s] = v2; a.pred[slot_col(s)] = i; }
                }
            }
            c_augrelax++;
        }
        if (err) break;
        // price update: columns scanned at an earlier level than the final one
#pragma unroll
        for (int s = 0; s < NC; s++)
            if ((readym >> s) & 1) vreg[s] = (vreg[s] + dreg[s]) - curmin;
        set_assigned(endofpath);
        __syncthreads();  // pred stores of all waves are complete (and, via L2, visible)
        if (tid == 0) {
            int ep = endofpath, i;
            do {
                i = ld_i32(a.pred + ep);
                colsol_set(ep, i);
                const int j1 = ep;
                ep = ld_i32(a.rowsol + i);
                st_i32(a.rowsol + i, j1);
                c_hops++;
            } while (i != freerow);
        }
        c_augs++;
        __syncthreads();
    }

    // ---- write back prices and colsol, then duals u and the total ----
#pragma unroll
    for (int s = 0; s < NC; s++) {
        const int c = slot_col(s);
        if (c < n) {
            a.v[c] = vreg[s];
            if constexpr (COLSOL_LDS) a.colsol[c] = s_colsol[c];
        }
    }
    __threadfence_block();
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < n; i += BLOCK) {
        const int j = ld_i32(a.rowsol + i);
        const T cij = cost[(int64_t)i * ld + j];
        const T vj = __hip_atomic_load(a.v + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.u[i] = cij - vj;
        part += (double)cij;
    }
    // deterministic tree sum
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    __shared__ double s_sum[NW];
    if ((tid & 63) == 0) s_sum[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; w++) t += s_sum[w];
        *a.total = t;
        a.counters[C_RT] = c_rt; a.counters[C_ARR] = c_arr; a.counters[C_AUG_INIT] = c_auginit;
        a.counters[C_AUG_RELAX] = c_augrelax; a.counters[C_AUGS] = c_augs; a.counters[C_HOPS] = c_hops;
        a.counters[C_FREE_CR] = c_free_cr; a.counters[C_FREE_A1] = c_free_a1; a.counters[C_FREE_A2] = c_free_a2;
        a.counters[C_ROWS_READ] = c_rt + c_arr + c_auginit + c_augrelax;
        *a.status = err;
    }
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x);       // (defined with the float32 fast path below)
__device__ __forceinline__ uint64_t wave_lexmin_u64(uint64_t k);

// ------------------------------------------------------------------------------------------
// Row caches for jv_chain_stream (the float64 chain beyond n = 4 096): the idea of the float32 fast path below -- prices
// only decrease during REDUCTION TRANSFER and AUGMENTING ROW REDUCTION, so the (at most) 63 columns of a row with the
// smallest reduced costs at build time, together with the 64th smallest reduced cost F (the floor), stay a certificate:
// a later scan whose second-smallest recomputed cached value is < F has found the row's exact lexicographic top-2 --
// in its simplest form.  One workgroup per row; the 64th smallest of the n order-preserving 64-bit keys is found bit by
// bit (64 counting passes over the row, which stays in L2), then the columns below it are written in column order with
// their raw costs.  Slot 63 holds the floor.
// ------------------------------------------------------------------------------------------
constexpr int WC_KC = 64;                          // slots per row; slot 63 = { sentinel, floor }
constexpr uint32_t WC_SENT = 0xFFFFFFFFu;
template <typename T> __device__ __forceinline__ uint64_t wide_ord(T h);
template <> __device__ __forceinline__ uint64_t wide_ord<double>(double h) {
    const uint64_t b = (uint64_t)__double_as_longlong(h + 0.0);
    return b ^ ((uint64_t)((int64_t)b >> 63) | 0x8000000000000000ull);
}
template <> __device__ __forceinline__ uint64_t wide_ord<float>(float h) {
    const uint32_t b = __float_as_uint(h + 0.0f);
    return (uint64_t)(b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u)) << 32;
}
template <typename T>
__global__ __launch_bounds__(256) void build_row_caches_wide(int n, int64_t ld, const T *__restrict__ cost, const T *__restrict__ v,
                                                             uint32_t *__restrict__ cache_col, T *__restrict__ cache_val) {
    __shared__ int s_cnt[2][4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = blockIdx.x;
    const T *row = cost + (int64_t)i * ld;
    uint64_t K = 0;
    int par = 0;
    if (n >= WC_KC) {
        for (int b = 63; b >= 0; b--) {
            const uint64_t cand = K | (1ull << b);
            int c = 0;
            for (int j = tid; j < n; j += 256) c += wide_ord<T>(row[j] - v[j]) < cand;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
            if (lane == 0) s_cnt[par][w] = c;
            __syncthreads();
            const int tot = s_cnt[par][0] + s_cnt[par][1] + s_cnt[par][2] + s_cnt[par][3];
            par ^= 1;
            if (tot < WC_KC) K = cand;             // fewer than 64 keys below cand: the 64th smallest is >= cand
        }
    } else K = ~0ull;                              // fewer than 64 columns: all of them are cached, the floor is +inf
    // K is now the 64th smallest key: at most 63 keys are strictly below it, and its value is the floor
    if (tid == 0) { s_base = 0; if (n < WC_KC) cache_val[(int64_t)i * WC_KC + WC_KC - 1] = (T)INFINITY; }
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + tid;
        const T c = j < n ? row[j] : (T)0;
        const T h = j < n ? c - v[j] : (T)0;
        const uint64_t key = j < n ? wide_ord<T>(h) : ~0ull;
        const bool sel = j < n && key < K;
        if (j < n && n >= WC_KC && key == K) cache_val[(int64_t)i * WC_KC + WC_KC - 1] = h;   // (equal keys = equal values)
        const uint64_t m = __ballot(sel);
        if (lane == 0) s_cnt[par][w] = __builtin_popcountll(m);
        __syncthreads();
        int off = s_base;
        for (int ww = 0; ww < w; ww++) off += s_cnt[par][ww];
        const int tot = s_cnt[par][0] + s_cnt[par][1] + s_cnt[par][2] + s_cnt[par][3];
        if (sel) {
            const int pos = off + __builtin_popcountll(m & ((1ull << lane) - 1));
            cache_col[(int64_t)i * WC_KC + pos] = (uint32_t)j;
            cache_val[(int64_t)i * WC_KC + pos] = c;
        }
        __syncthreads();
        if (tid == 0) s_base = off + tot;          // (thread 0 is in wave 0: its off is the running base)
        par ^= 1;
        __syncthreads();
    }
    const int used = s_base;
    for (int p = used + tid; p < WC_KC; p += 256) {
        cache_col[(int64_t)i * WC_KC + p] = WC_SENT;
        if (p < WC_KC - 1) cache_val[(int64_t)i * WC_KC + p] = (T)0;
    }
}

// ------------------------------------------------------------------------------------------
// The same chain with every per-column quantity (prices, distances, predecessors, scan levels) in L2-resident
// global memory instead of VGPRs: jv_chain<double, CH> spills badly beyond CH = 2 (1024 threads leave 128 VGPRs
// per lane), this variant has no per-lane arrays at all and takes any n.  Column c is only ever touched by thread
// (c / VW) % BLOCK, so plain loads and stores in program order are all the ordering the per-column arrays need.
// A row scan = one coalesced sweep of the cost row (HBM) and of the price vector (L2).  Same arithmetic, same
// tie-breaking, same step budget as jv_chain / the oracle.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void jv_chain_stream(ChainArgs<T> a) {
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    __shared__ RedScratch<T> red;
    __shared__ int s_run[4];
    __shared__ long long s_run_arr;
    // with the row caches (n <= 65 535): colsol lives in LDS as u16 during REDUCTION TRANSFER / AUGMENTING ROW REDUCTION -- the cached
    // step then gathers the owners from LDS and stores nothing but a lowered price to global memory (a load is only returned
    // after the stores issued before it are acknowledged)
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_cs[];
    // ... and, where they fit beside it (a.v_lds: n <= ~15 800 in float64), the prices too: the cached step then touches global
    // memory for the row's cache only
    const bool vl = a.v_lds != 0;
    T *const s_vT = reinterpret_cast<T *>(dyn_cs);
    uint16_t *const s_cs16 = reinterpret_cast<uint16_t *>(dyn_cs + (vl ? (((size_t)a.n * sizeof(T) + 15) & ~(size_t)15) : 0));
    const bool csl = a.cs_lds != 0;
#define V_GET(j) (vl ? s_vT[j] : ld_agent(v + (j)))
#define CS_GET(j) (csl ? (s_cs16[j] == 0xFFFFu ? -1 : (int32_t)s_cs16[j]) : ld_i32(a.colsol + (j)))
    const int tid = threadIdx.x;
    const int n = a.n;
    const int64_t ld = a.ld;
    const T *__restrict__ cost = a.cost;
    T *v = a.v, *d = a.dwork;
    int32_t *lvl = a.lvl, *pred = a.pred;
    int par = 0;
    const T INF = (T)INFINITY;
    const int nq = (n + VW - 1) / VW;                  // vector chunks per row (rows and v are VW-aligned and padded)
    const V *vp = reinterpret_cast<const V *>(v);

    long long c_rt = 0, c_arr = 0, c_auginit = 0, c_augrelax = 0, c_augs = 0, c_hops = 0, c_free_a1 = 0, c_dense = 0;
    const int lane = tid & 63;
    int numfree = 0, nrt = 0;
    {
        const int R = (n + BLOCK - 1) / BLOCK;
        const int r0 = min(n, tid * R), r1 = min(n, r0 + R);
        int f = 0, g = 0;
        for (int i = r0; i < r1; i++) { const int mt = a.matches[i]; f += (mt == 0); g += (mt == 1); }
        int of = wg_exscan(f, red, par, &numfree);
        int og = wg_exscan(g, red, par, &nrt);
        for (int i = r0; i < r1; i++) {
            const int mt = a.matches[i];
            if (mt == 0) st_i32(a.freerows + of++, i);
            else if (mt == 1) st_i32(a.rtrows + og++, i);
        }
    }
    if (csl) for (int c = tid; c < n; c += BLOCK) { const int32_t r = a.colsol[c]; s_cs16[c] = r < 0 ? (uint16_t)0xFFFFu : (uint16_t)r; }
    if (vl) for (int c = tid; c < n; c += BLOCK) s_vT[c] = v[c];
    __syncthreads();
    const long long c_free_cr = numfree;

    // ---- REDUCTION TRANSFER ----
    if (n > 1) {
        for (int k = 0; k < nrt; k++) {
            const int i = ld_i32(a.rtrows + k);
            const int j1 = ld_i32(a.rowsol + i);
            T mn = INF;
            bool certified = false;
            if (a.cache_col) {
                // cached scan, by every wave for itself (same loads, same result: no exchange needed)
                const uint32_t col = a.cache_col[(int64_t)i * WC_KC + lane];
                const T cv = a.cache_val[(int64_t)i * WC_KC + lane];
                const T F = __shfl(cv, WC_KC - 1);
                T h = INF;
                if (col != WC_SENT && (int)col != j1) h = cv - V_GET(col);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { const T o = __shfl_xor(h, off); h = o < h ? o : h; }
                mn = h;
                certified = mn < F;                    // every column outside the cache is still >= F
            }
            if (!certified) {
                const V *rp = reinterpret_cast<const V *>(cost + (int64_t)i * ld);
                Min1<T> loc; loc.u = INF; loc.k = 0; loc.a = 0;
                for (int q = tid; q < nq; q += BLOCK) {
                    const V x = rp[q], vv = vl ? reinterpret_cast<const V *>(s_vT)[q] : vp[q];
#pragma unroll
                    for (int e = 0; e < VW; e++) {
                        const int c = q * VW + e;
                        if (c < n && c != j1) { const T h = vec_get<T>(x, e) - vec_get<T>(vv, e); if (h < loc.u) loc.u = h; }
                    }
                }
                mn = wg_min1(loc, red, par).u;
                c_dense++;
            }
            if (((j1 / VW) % BLOCK) == tid) { if (vl) s_vT[j1] = s_vT[j1] - mn; else v[j1] = v[j1] - mn; }
            if (a.cache_col) __syncthreads();          // the new price has reached L2 (the barrier waits for the store) before any wave gathers it
            c_rt++;
        }
    }

    // ---- AUGMENTING ROW REDUCTION, two sweeps ----
    const long long arr_budget = 1000ll * n + 1000000ll;   // == JV_ARR_BUDGET(n) of the oracle
    for (int sweep = 0; sweep < 2; sweep++) {
        int k = 0;
        const int prev = numfree;
        numfree = 0;
        int carry = -1;
        while (carry >= 0 || k < prev) {
            int pend = -1;
            if (a.cache_col) {
                // ---- wave 0 alone follows the displacement chains while the row caches certify its scans: no barrier, no
                // other wave involved.  Lane = cache entry: the row's cache (requested as soon as the previous step knew the
                // next row), one round of gathers (price and owner of every cached column), the top-2 by wave shuffles,
                // the step's stores from lane 0 (agent scope: the wave's own later gathers see them, in L2).  It stops at the
                // first row whose cache does not certify its top-2 (that row is then scanned by the whole workgroup, below),
                // at the end of the sweep's list, or at the step budget. ----
                if (tid < 64) {
                    int pf_row = -1;
                    uint32_t pf_col = WC_SENT;
                    T pf_cv = 0;
                    while ((carry >= 0 || k < prev) && c_arr < arr_budget) {
                        const bool from_carry = carry >= 0;
                        const int i = from_carry ? carry : ld_i32(a.freerows + k);
                        uint32_t col; T cv;
                        if (pf_row == i) { col = pf_col; cv = pf_cv; }
                        else { col = ld_agent(a.cache_col + (int64_t)i * WC_KC + lane); cv = ld_agent(a.cache_val + (int64_t)i * WC_KC + lane); }
                        const T F = __shfl(cv, WC_KC - 1);
                        const bool valid = col != WC_SENT;
                        const T vj = V_GET(valid ? col : 0u);
                        const int32_t csj = CS_GET(valid ? col : 0u);
                        // the smallest reduced cost and its lane (cache rows are sorted by column: the lowest lane is the lowest
                        // column), then the smallest of the rest: reductions on order-preserving 64-bit keys, as in the float32 chain
                        const T h = valid ? cv - vj : INF;
                        const uint64_t kh = valid ? wide_ord<T>(h) : ~0ull;
                        const uint64_t m1 = wave_lexmin_u64(kh);
                        const int l1 = __builtin_ctzll(__ballot(kh == m1) | (1ull << 63));
                        int i0 = __shfl(csj, l1);
                        if (i0 >= 0 && m1 != ~0ull) {                         // (almost always) the next row of the chain: request its cache now
                            pf_row = i0;
                            pf_col = ld_agent(a.cache_col + (int64_t)i0 * WC_KC + lane);
                            pf_cv = ld_agent(a.cache_val + (int64_t)i0 * WC_KC + lane);
                        }
                        const uint64_t kh2 = lane == l1 ? ~0ull : kh;
                        const uint64_t m2 = wave_lexmin_u64(kh2);
                        const int l2 = __builtin_ctzll(__ballot(kh2 == m2) | (1ull << 63));
                        const T u1 = __shfl(h, l1), u2 = m2 == ~0ull ? INF : __shfl(h, l2);
                        if (from_carry) carry = -1; else k++;                 // the row is taken, certified or not
                        if (!(u2 < F)) { pend = i; break; }                   // (u2 < F: these are the row's exact lexicographic top-2)
                        c_arr++;
                        const int jfirst = (int)__shfl(col, l1);
                        int j1 = jfirst;
                        const int j2 = (int)__shfl(col, l2);
                        const T vj1 = __shfl(vj, l1);
                        const T vnew = vj1 - (u2 - u1);
                        const bool lowers = vnew < vj1;
                        if (!lowers && i0 >= 0) { j1 = j2; i0 = __shfl(csj, l2); }
                        if (lane == 0) {
                            if (lowers) { if (vl) s_vT[jfirst] = vnew; else __hip_atomic_store(v + jfirst, vnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            // (rowsol is not read during this phase and equals the inverse of colsol: rebuilt after it)
                            if (csl) s_cs16[j1] = (uint16_t)i; else st_i32(a.colsol + j1, i);
                            if (i0 >= 0 && !lowers) st_i32(a.freerows + numfree, i0);
                        }
                        if (i0 >= 0) { if (lowers) carry = i0; else numfree++; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // the stores have reached L2
                    if (lane == 0) { s_run[0] = k; s_run[1] = carry; s_run[2] = numfree; s_run[3] = pend; s_run_arr = c_arr; }
                }
                __syncthreads();
                k = s_run[0]; carry = s_run[1]; numfree = s_run[2]; pend = s_run[3]; c_arr = s_run_arr;
                __syncthreads();                                             // (s_run is rewritten by the next run)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // wave 0's prices: drop what this CU's L1 holds
                if (pend < 0 && c_arr < arr_budget) continue;                // the sweep's list is exhausted
            }
            if (pend < 0 && c_arr >= arr_budget) {
                if (carry >= 0) { if (tid == 0) st_i32(a.freerows + numfree, carry); numfree++; carry = -1; }
                while (k < prev) { const int r = ld_i32(a.freerows + k); k++; if (tid == 0) st_i32(a.freerows + numfree, r); numfree++; }
                break;
            }
            int i;
            if (pend >= 0) i = pend;
            else if (carry >= 0) { i = carry; carry = -1; }
            else { i = ld_i32(a.freerows + k); k++; }
            Top2<T> g;
            const bool certified = false;
            if (!certified) {
                const V *rp = reinterpret_cast<const V *>(cost + (int64_t)i * ld);
                Top2<T> loc; loc.u1 = INF; loc.k1 = 0xFFFFFFFFu; loc.a1 = 0; loc.u2 = INF; loc.k2 = 0xFFFFFFFFu;
                for (int q = tid; q < nq; q += BLOCK) {
                    const V x = rp[q], vv = vl ? reinterpret_cast<const V *>(s_vT)[q] : vp[q];
#pragma unroll
                    for (int e = 0; e < VW; e++) {
                        const int c = q * VW + e;
                        if (c < n) top2_push(loc, vec_get<T>(x, e) - vec_get<T>(vv, e), (uint32_t)c, vec_get<T>(vv, e));
                    }
                }
                g = wg_top2(loc, red, par);
                c_dense++;
            }
            c_arr++;
            int j1 = (int)g.k1;
            const int j2 = (int)g.k2;
            int i0 = CS_GET(j1);
            const T vj1 = g.a1;
            const T vnew = vj1 - (g.u2 - g.u1);
            const bool lowers = vnew < vj1;
            if (lowers) { if (((j1 / VW) % BLOCK) == tid) { if (vl) s_vT[j1] = vnew; else v[j1] = vnew; } }
            else if (i0 >= 0) { j1 = j2; i0 = CS_GET(j2); }
            __syncthreads();  // every wave has read colsol for this step before it changes
            if (tid == 0) { if (csl) s_cs16[j1] = (uint16_t)i; else st_i32(a.colsol + j1, i); }
            if (i0 >= 0) {
                if (lowers) carry = i0;
                else { if (tid == 0) st_i32(a.freerows + numfree, i0); numfree++; }
            }
            __syncthreads();  // the colsol store is visible (L2) before the next step reads it
        }
        __syncthreads();
        if (sweep == 0) c_free_a1 = numfree;
    }
    const long long c_free_a2 = numfree;
    // colsol back to global memory, rowsol = its inverse (ARR kept only colsol current)
    __syncthreads();
    for (int c = tid; c < n; c += BLOCK) {
        const int32_t r = CS_GET(c);
        if (csl) a.colsol[c] = r;
        if (vl) v[c] = s_vT[c];
        if (r >= 0) a.rowsol[r] = c;
    }
    __threadfence();
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#undef CS_GET
#undef V_GET

    // ---- AUGMENTATION: relaxation and the search for the next pick share one sweep ----
    int err = 0;
    for (int f = 0; f < numfree && !err; f++) {
        const int freerow = ld_i32(a.freerows + f);
        Min1<T> loc; loc.u = INF; loc.k = 0xFFFFFFFFu; loc.a = 0;
        {
            const V *rp = reinterpret_cast<const V *>(cost + (int64_t)freerow * ld);
            for (int q = tid; q < nq; q += BLOCK) {
                const V x = rp[q], vv = vp[q];
#pragma unroll
                for (int e = 0; e < VW; e++) {
                    const int c = q * VW + e;
                    if (c < n) {
                        const T dd = vec_get<T>(x, e) - vec_get<T>(vv, e);
                        d[c] = dd; pred[c] = freerow; lvl[c] = 0;
                        const uint32_t key = (uint32_t)c | (ld_i32(a.colsol + c) >= 0 ? 0x80000000u : 0u);
                        if (lex_less(dd, key, loc.u, loc.k)) { loc.u = dd; loc.k = key; loc.a = vec_get<T>(vv, e); }
                    }
                }
            }
            c_auginit++;
        }
        bool have = false;
        T curmin = 0;
        int endofpath = -1, level = 0;
        for (;;) {
            const Min1<T> g = wg_min1(loc, red, par);
            if (g.k == 0xFFFFFFFFu) { err = CYTO_ERR_INTERNAL; break; }
            const int jp = (int)(g.k & 0x7FFFFFFFu);
            if (!have || g.u != curmin) { level++; curmin = g.u; have = true; }
            if (!(g.k & 0x80000000u)) { endofpath = jp; break; }
            if (((jp / VW) % BLOCK) == tid) lvl[jp] = level;                 // scanned (its d keeps the value at the pick)
            const int i = ld_i32(a.colsol + jp);
            const T cip = cost[(int64_t)i * ld + jp];
            const T h = (cip - g.a) - curmin;
            const V *rp = reinterpret_cast<const V *>(cost + (int64_t)i * ld);
            loc.u = INF; loc.k = 0xFFFFFFFFu; loc.a = 0;
            for (int q = tid; q < nq; q += BLOCK) {
                const V x = rp[q], vv = vp[q];
#pragma unroll
                for (int e = 0; e < VW; e++) {
                    const int c = q * VW + e;
                    if (c < n && lvl[c] == 0) {
                        const T v2 = (vec_get<T>(x, e) - vec_get<T>(vv, e)) - h;
                        T dc = d[c];
                        if (v2 < dc) { dc = v2; d[c] = v2; pred[c] = i; }
                        const uint32_t key = (uint32_t)c | (ld_i32(a.colsol + c) >= 0 ? 0x80000000u : 0u);
                        if (lex_less(dc, key, loc.u, loc.k)) { loc.u = dc; loc.k = key; loc.a = vec_get<T>(vv, e); }
                    }
                }
            }
            c_augrelax++;
        }
        if (err) break;
        // price update: columns scanned at an earlier level than the final one
        for (int q = tid; q < nq; q += BLOCK) {
#pragma unroll
            for (int e = 0; e < VW; e++) {
                const int c = q * VW + e;
                if (c < n) { const int lv = lvl[c]; if (lv != 0 && lv < level) v[c] = (v[c] + d[c]) - curmin; }
            }
        }
        __threadfence_block();
        __syncthreads();  // pred / price stores of all waves are complete (and, via L2, visible)
        if (tid == 0) {
            int ep = endofpath, i;
            do {
                i = ld_i32(pred + ep);
                st_i32(a.colsol + ep, i);
                const int j1 = ep;
                ep = ld_i32(a.rowsol + i);
                st_i32(a.rowsol + i, j1);
                c_hops++;
            } while (i != freerow);
        }
        c_augs++;
        __syncthreads();
    }

    // ---- duals u and the total ----
    __threadfence_block();
    __syncthreads();
    double part = 0.0;
    if (!err) {
        for (int i = tid; i < n; i += BLOCK) {
            const int j = ld_i32(a.rowsol + i);
            const T cij = cost[(int64_t)i * ld + j];
            const T vj = __hip_atomic_load(a.v + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.u[i] = cij - vj;
            part += (double)cij;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    __shared__ double s_sum[NW];
    if ((tid & 63) == 0) s_sum[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; w++) t += s_sum[w];
        *a.total = t;
        a.counters[C_RT] = c_rt; a.counters[C_ARR] = c_arr; a.counters[C_AUG_INIT] = c_auginit;
        a.counters[C_AUG_RELAX] = c_augrelax; a.counters[C_AUGS] = c_augs; a.counters[C_HOPS] = c_hops;
        a.counters[C_FREE_CR] = c_free_cr; a.counters[C_FREE_A1] = c_free_a1; a.counters[C_FREE_A2] = c_free_a2;
        a.counters[C_ROWS_READ] = (a.cache_col ? c_dense : c_rt + c_arr) + c_auginit + c_augrelax;
        *a.status = err;
    }
}



// ==========================================================================================
// float32 fast path ("v2").  Same arithmetic, same results, far fewer HBM row reads.
//
// Observation: during REDUCTION TRANSFER and AUGMENTING ROW REDUCTION the prices v[j] only ever
// DECREASE (RT subtracts a non-negative minimum, ARR only stores vj1_new when vj1_new < v[j1]),
// so every reduced cost h(i,j) = c[i][j] - v[j] only ever INCREASES.  Hence a row cache
//     C_i = { j : h(i,j) < F_i }  (at most KCU = 63 columns, with their raw c[i][j])  and the floor F_i
// built at any earlier time stays a valid certificate: every column outside C_i still has
// h >= F_i.  A later scan of row i only needs the cached columns: if the second-smallest
// recomputed cached value is < F_i, the cached top-2 IS the exact lexicographic top-2 of the
// whole row (bit-identical to the full scan).  Otherwise the row is re-scanned from HBM by the
// whole workgroup (and its cache rebuilt).  The caches of all rows are built once, right after
// the column reduction, by a full-chip streaming kernel (build_row_caches).
//
// A cached step runs on ONE wave (lane = cache entry, lane 63 carries the floor): two coalesced
// 256-B loads, LDS gathers of v and colsol, three 32-bit DPP all-reduces on order-preserving float
// keys and a few scalar updates -- no barrier, no HBM row.
// ==========================================================================================
constexpr int BLOCK2 = 512;        // 8 waves: 256 VGPRs per lane for the column-resident state
constexpr int NW2 = BLOCK2 / 64;
enum { OP_EXIT = 0, OP_REFRESH = 1, OP_AUG = 2 };
enum { C2_DENSE_REFRESH = C_NCOUNTERS, C2_AUG_SKIPPED, C2_NCOUNTERS };

// a wave's candidate for the next pick of the dense augmentation, with everything the step needs once it wins
struct __attribute__((aligned(16))) PickRec { uint64_t key; int32_t row; float h; float vjp; int32_t g; int32_t skip; int32_t srow; };   // srow: the stored row (row map applied)
struct Scratch2 {
    uint64_t m1[2][NW2], m2[2][NW2];
    int cnt[2][NW2];
    int cmd_op, cmd_row;
    double sum[NW2];
};

__device__ __forceinline__ K2 wg_k2(K2 t, Scratch2 &s, int &par) {
    t = k2_wave_allreduce(t);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { s.m1[par][w] = t.m1; s.m2[par][w] = t.m2; }
    __syncthreads();
    K2 r; r.m1 = s.m1[par][lane & (NW2 - 1)]; r.m2 = s.m2[par][lane & (NW2 - 1)];
    k2_step<0xB1>(r); k2_step<0x4E>(r); k2_step<0x141>(r);   // 8 distinct partials per half-row: 3 steps only
    par ^= 1;
    r.m1 = readlane64(r.m1, 0); r.m2 = readlane64(r.m2, 0);
    return r;
}
__device__ __forceinline__ uint64_t wg_min64(uint64_t x, Scratch2 &s, int &par) {
    x = min64_wave_allreduce(x);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s.m1[par][w] = x;
    __syncthreads();
    uint64_t r = s.m1[par][lane & (NW2 - 1)];
    r = min64_row_allreduce(r);
    par ^= 1;
    return readlane64(r, 0);
}
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// sum over the waves of a wave-uniform int; *base = exclusive prefix for this wave
__device__ __forceinline__ int wg_sum_waves(int wave_val, Scratch2 &s, int &par, int *base) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s.cnt[par][w] = wave_val;
    __syncthreads();
    int tot = 0, b = 0;
#pragma unroll
    for (int i = 0; i < NW2; i++) { const int c = s.cnt[par][i]; if (i < w) b += c; tot += c; }
    par ^= 1;
    if (base) *base = b;
    return tot;
}
// exclusive scan of a per-thread int in thread order (set-up only; not on the hot path)
__device__ __forceinline__ int wg_exscan2(int x, Scratch2 &s, int &par, int *total) {
    const int lane = threadIdx.x & 63;
    int inc = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(inc, off); if (lane >= off) inc += y; }
    const int wtot = __shfl(inc, 63);
    int base = 0;
    *total = wg_sum_waves(wtot, s, par, &base);
    return base + inc - x;
}

struct CachePtrs {
    uint32_t *col;   // [n][KC]   slot 63: COLSENT
    float *val;      // [n][KC]   raw c[i][col]; slot 63: the floor (-inf = no usable cache)
};

#define SLOT_COL(sl) ((((sl) / 4) * BLOCK2 + tid) * 4 + ((sl) % 4))

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// One cost row through a bounds-checked buffer descriptor: lane `tid` gets 16 bytes of every
// 8 KiB chunk (a wave reads 1 KiB contiguous); bytes past the row pitch read as 0, so no branches.
template <int CH, int BS = BLOCK2>
__device__ __forceinline__ void load_row4(const float *__restrict__ cost, int64_t ld, int i, int n, int tid, float4 (&x)[CH]) {
    (void)n;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(cost + (int64_t)i * ld), 0, (int)(ld * 4), 0x00020000);
#pragma unroll
    for (int m = 0; m < CH; m++) {
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, m * BS * 16, 0);
        x[m] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
    }
}
__device__ __forceinline__ float fmin_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The <= 63 entries of a rebuilt cache row are sorted by column (wave 0, bitonic network on 64 lanes; slot 63 keeps the
// floor): lane order is then column order, so "lowest column among equal reduced costs" -- the oracle's tie-break, and
// on CytoSPACE's duplicated spot rows ties at a reduced cost of exactly 0 are the rule -- is the lowest lane of a ballot
// instead of one more wave reduction per tie.
__device__ __forceinline__ void sort_cache_row(uint32_t *__restrict__ ccol, float *__restrict__ cval, float tau) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    uint32_t key = lane < KCU ? ccol[lane] : COLSENT;
    float val = lane < KCU ? cval[lane] : 0.0f;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t pk = __shfl_xor(key, j);
            const float pv = __shfl_xor(val, j);
            const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
            const bool sw = take_min ? (pk < key) : (pk > key);
            key = sw ? pk : key; val = sw ? pv : val;
        }
    }
    // unused slots and the floor slot all carry COLSENT and end up last, in no particular order: restore their values
    if (key == COLSENT) val = 0.0f;
    if (lane == KCU) { key = COLSENT; val = tau; }
    ccol[lane] = key; cval[lane] = val;
}

// Full scan of row i by the whole workgroup: exact two smallest keys of (c[i][j]-v[j], j) over all
// columns, plus a rebuilt cache for the row.  vreg = current prices of this thread's columns.
// delta adapts the cache threshold tau = umin + delta so that KCU/2..KCU columns qualify.
template <int CH>
__device__ __forceinline__ K2 refresh_row(int i, int n, int64_t ld, const float *__restrict__ cost,
                                          const float (&vreg)[CH * 4], uint64_t validm, uint32_t *__restrict__ cache_col,
                                          float *__restrict__ cache_val, float &delta, Scratch2 &s, int &par) {
    constexpr int NC = CH * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    float4 x[CH];
    load_row4<CH>(cost, ld, i, n, tid, x);
#define HVAL(sl) (vec_get<float>(x[(sl) / 4], (sl) % 4) - vreg[sl])
    K2 loc; loc.m1 = KEYMAX; loc.m2 = KEYMAX;
#pragma unroll
    for (int sl = 0; sl < NC; sl++) {
        if ((validm >> sl) & 1) k2_push(loc, mkkey(HVAL(sl), (uint32_t)SLOT_COL(sl)));
        if ((sl & 3) == 3) SCHED_FENCE();
    }
    const K2 g = wg_k2(loc, s, par);
    const float umin = key_val(g.m1);

    // threshold search (every thread computes the same sequence)
    float lo = 0.0f, hi = INFINITY, tau = INFINITY;
    int cnt = 0;
    bool okc = false;
    if (!(delta > 0.0f) || !(delta < 1e30f)) delta = 1e-3f;
    for (int it = 0; it < 24 && !okc; it++) {
        tau = umin + delta;
        uint32_t tc = 0;
#pragma unroll
        for (int sl = 0; sl < NC; sl++) { tc += (((validm >> sl) & 1) && HVAL(sl) < tau) ? 1u : 0u; if ((sl & 3) == 3) SCHED_FENCE(); }
        cnt = wg_sum_waves((int)wave_sum_u32(tc), s, par, nullptr);
        if (cnt > KCU) {
            hi = delta;
            const float mid = (lo > 0.0f) ? 0.5f * (lo + hi) : 0.5f * delta;
            if (!(mid < hi) || !(mid > lo)) break;   // cannot separate: too many ties just above umin
            delta = mid;
        } else if (cnt < KCU / 2 && cnt < n && delta < 1e30f) {
            lo = delta;
            const float mid = (hi < INFINITY) ? 0.5f * (lo + hi) : 2.0f * delta;
            if (hi < INFINITY && (!(mid < hi) || !(mid > lo))) { okc = true; break; }  // best separable threshold
            delta = mid;
        } else {
            okc = true;
        }
    }
    if (!okc || cnt > KCU) {
        // last resort: the largest threshold known to admit <= KCU columns (possibly none)
        if (lo > 0.0f) { delta = lo; tau = umin + lo; } else { tau = -INFINITY; }
    }
    // compaction: entries ordered by (wave, lane, slot); unused entries get the sentinel
    uint32_t tc = 0;
#pragma unroll
    for (int sl = 0; sl < NC; sl++) { tc += (((validm >> sl) & 1) && HVAL(sl) < tau) ? 1u : 0u; if ((sl & 3) == 3) SCHED_FENCE(); }
    int base = 0;
    cnt = wg_sum_waves((int)wave_sum_u32(tc), s, par, &base);
    if (cnt > KCU) { tau = -INFINITY; cnt = 0; tc = 0; }   // (cannot happen; keeps the cache valid regardless)
    uint32_t *ccol = cache_col + (int64_t)i * KC;
    float *cval = cache_val + (int64_t)i * KC;
    if (__ballot(tc != 0)) {
        // exclusive prefix of the per-lane counts inside the wave
        uint32_t inc = tc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(inc, off); if (lane >= off) inc += y; }
        int pos = base + (int)(inc - tc);
        if (tc != 0) {
#pragma unroll
            for (int sl = 0; sl < NC; sl++) {
                if (((validm >> sl) & 1) && HVAL(sl) < tau) {
                    ccol[pos] = (uint32_t)SLOT_COL(sl);
                    cval[pos] = vec_get<float>(x[sl / 4], sl % 4);
                    pos++;
                }
            }
        }
    }
    if (tid >= cnt && tid < KCU) { ccol[tid] = COLSENT; cval[tid] = 0.0f; }
    if (tid == KCU) { ccol[KCU] = COLSENT; cval[KCU] = tau; }
#undef HVAL
    __syncthreads();  // the rebuilt cache is complete before anyone may read it
    sort_cache_row(ccol, cval, tau);
    return g;
}

// Caches for all rows against the post-column-reduction prices: a full-chip streaming pass.
template <int CH>
__global__ __launch_bounds__(BLOCK2) void build_row_caches(int n, int64_t ld, const float *__restrict__ cost,
                                                          const float *__restrict__ v, uint32_t *__restrict__ cache_col,
                                                          float *__restrict__ cache_val, const int32_t *__restrict__ rowmap,
                                                          const int32_t *__restrict__ same_prev) {
    constexpr int NC = CH * 4;
    __shared__ Scratch2 s;
    const int tid = threadIdx.x;
    int par = 0;
    float vreg[NC];
    uint64_t validm = 0;
#pragma unroll
    for (int sl = 0; sl < NC; sl++) {
        const int c = SLOT_COL(sl);
        vreg[sl] = 0.0f;
        if (c < n) { validm |= (1ull << sl); vreg[sl] = v[c]; }
    }
    float delta = 0.0f;
    // (a contiguous range of rows per workgroup, not a stride: with runs of identical rows -- ten per spot at c3 -- a stride that
    //  shares a factor with the run length leaves the first rows of the runs, the only ones that are built, to a few workgroups)
    const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x, i_end = min(n, ((int)blockIdx.x + 1) * per);
    for (int i = (int)blockIdx.x * per; i < i_end; i++) {
        if (same_prev && same_prev[i]) continue;                          // a copy of the previous row: replicate_group_caches
        (void)refresh_row<CH>(i, n, ld, RBASE(cost, rowmap, i, ld), vreg, validm, cache_col, cache_val, delta, s, par);
    }
}

// Runs of bitwise identical consecutive rows (same_prev[i] = row i equals row i - 1: rows_same_as_prev / rows_same_from_map --
// CytoSPACE repeats every spot row slots[s] times) have identical caches: the build kernels make the cache of the first row of
// every run, this one copies it to the others (a wave per run; c3: ten rows per spot -- a tenth of the cache-build work).
__global__ __launch_bounds__(256) void replicate_group_caches(int n, const int32_t *__restrict__ same_prev, uint32_t *__restrict__ cache_col,
                                                               float *__restrict__ cache_val) {
    // a wave per DESTINATION row (a copy of the row before it): its run's first row is found 64 rows at a time (lane l looks at row
    // i - l), so a long run -- one spot with thousands of slots, a constant matrix -- is copied by as many waves as it has rows
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int i = gw; i < n; i += nw) {
        if (!same_prev[i]) continue;                                       // the first row of its run: built, not copied
        int f = -1;
        for (int base = i; f < 0; base -= 64) {
            const int r = base - lane;
            const uint64_t starts = __ballot(r >= 0 && !same_prev[r]);    // (row 0 always starts a run)
            if (starts) f = base - (__ffsll((unsigned long long)starts) - 1);
        }
        cache_col[(int64_t)i * KC + lane] = cache_col[(int64_t)f * KC + lane];
        cache_val[(int64_t)i * KC + lane] = cache_val[(int64_t)f * KC + lane];
    }
}

// Streaming variant of refresh_row for large n: no per-lane arrays.  The row and the prices come through
// bounds-checked buffer descriptors (whole quads stay in range: ld % 4 == 0 and v is followed by u in the
// workspace; the ragged tail is handled by the per-column validity compare) and the row is swept several times -- the first sweep from HBM, the rest
// from L2: top-2 keys, threshold search (count sweeps), per-lane counts, compaction.  Same cache contract
// as refresh_row.  VAUX = cache policy of the price loads: 0x10 (sc1, agent scope) inside the chain where
// wave 0 updates prices with agent-scope stores, 0 for the read-only build pass.
#define STREAM_SWEEP(BODY)                                                                               \
    for (int q = tid; q < nquad; q += BLOCK2) {                                                          \
        const u32x4_t xr_ = __builtin_amdgcn_raw_buffer_load_b128(rrow, q * 16, 0, 0);                   \
        const u32x4_t vr_ = __builtin_amdgcn_raw_buffer_load_b128(rv, q * 16, 0, VAUX);                  \
        const uint32_t c0_ = (uint32_t)q * 4;                                                            \
        { const float raw = __uint_as_float(xr_.x); const float h = raw - __uint_as_float(vr_.x); const uint32_t c = c0_;     if (c < (uint32_t)n) { BODY } } \
        { const float raw = __uint_as_float(xr_.y); const float h = raw - __uint_as_float(vr_.y); const uint32_t c = c0_ + 1; if (c < (uint32_t)n) { BODY } } \
        { const float raw = __uint_as_float(xr_.z); const float h = raw - __uint_as_float(vr_.z); const uint32_t c = c0_ + 2; if (c < (uint32_t)n) { BODY } } \
        { const float raw = __uint_as_float(xr_.w); const float h = raw - __uint_as_float(vr_.w); const uint32_t c = c0_ + 3; if (c < (uint32_t)n) { BODY } } \
    }
template <int VAUX>
__device__ __forceinline__ K2 refresh_row_stream(int i, int n, int64_t ld, const float *__restrict__ cost, const float *gv,
                                                 uint32_t *__restrict__ cache_col, float *__restrict__ cache_val, float &delta,
                                                 float &tau_guess, Scratch2 &s, int &par) {
    // Sweeps of the row: the first from HBM (top-2 keys AND the count below the previous row's floor -- neighbouring rows need
    // similar floors, and ANY floor that admits <= 63 columns makes a valid cache), then one per step of the threshold search from
    // L2.  Every counting sweep also keeps the thread's first two columns below the threshold, so the cache is written without
    // another sweep (a thread with more than two re-sweeps its own quads).  Typically one or two sweeps per row (six before:
    // the build was bound by L2, 10 TB/s of it for 1.5 TB/s of HBM at n = 50 000).
    const int tid = threadIdx.x, lane = tid & 63;
    const int nquad = (n + 3) >> 2;
    const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(cost + (int64_t)i * ld), 0, (int)(ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gv), 0, nquad * 16, 0x00020000);
    uint32_t tc = 0, hc0 = 0, hc1 = 0;                          // this thread's columns below the threshold: count, the first two
    float hr0 = 0.0f, hr1 = 0.0f;
#define STREAM_HIT(T) if (h < (T)) { if (tc == 0) { hc0 = c; hr0 = raw; } else if (tc == 1) { hc1 = c; hr1 = raw; } tc++; }
    K2 loc; loc.m1 = KEYMAX; loc.m2 = KEYMAX;
    const float tau0 = tau_guess;
    STREAM_SWEEP(k2_push(loc, mkkey(h, c)); STREAM_HIT(tau0))
    const K2 g = wg_k2(loc, s, par);
    const float umin = key_val(g.m1);
    int base = 0;
    int cnt = wg_sum_waves((int)wave_sum_u32(tc), s, par, &base);
    float tau = tau0;
    bool have = tau0 < INFINITY && cnt <= KCU && (cnt >= KCU / 2 || cnt >= n);      // the guess fits: done after one sweep
    if (have) delta = tau - umin;
    if (!have) {
        float lo = 0.0f, hi = INFINITY;
        bool okc = false;
        if (tau0 < INFINITY && tau0 > umin && cnt > 0) {          // scale the guess by how far its count was off
            const float d0 = tau0 - umin;
            if (cnt > KCU) { hi = d0; delta = d0 * fmaxf(0.0625f, 0.75f * (float)KCU / (float)cnt); }
            else { lo = d0; delta = d0 * fminf(16.0f, 0.75f * (float)KCU / (float)cnt); }
        }
        if (!(delta > 0.0f) || !(delta < 1e30f)) delta = 1e-3f;
        for (int it = 0; it < 24 && !okc; it++) {
            tau = umin + delta;
            tc = 0;
            STREAM_SWEEP(STREAM_HIT(tau))
            cnt = wg_sum_waves((int)wave_sum_u32(tc), s, par, &base);
            if (cnt > KCU) {
                hi = delta;
                const float mid = (lo > 0.0f) ? 0.5f * (lo + hi) : 0.5f * delta;
                if (!(mid < hi) || !(mid > lo)) break;
                delta = mid;
            } else if (cnt < KCU / 2 && cnt < n && delta < 1e30f) {
                lo = delta;
                const float mid = (hi < INFINITY) ? 0.5f * (lo + hi) : 2.0f * delta;
                if (hi < INFINITY && (!(mid < hi) || !(mid > lo))) { okc = true; break; }
                delta = mid;
            } else {
                okc = true;
            }
        }
        if (!okc || cnt > KCU) {
            // the largest threshold known to admit <= KCU columns (possibly none): counted once more, with its columns
            if (lo > 0.0f) { delta = lo; tau = umin + lo; } else { tau = -INFINITY; }
            tc = 0;
            STREAM_SWEEP(STREAM_HIT(tau))
            cnt = wg_sum_waves((int)wave_sum_u32(tc), s, par, &base);
            if (cnt > KCU) { tau = -INFINITY; cnt = 0; tc = 0; }
        }
    }
    tau_guess = tau > -INFINITY ? tau : INFINITY;
#undef STREAM_HIT
    uint32_t *ccol = cache_col + (int64_t)i * KC;
    float *cval = cache_val + (int64_t)i * KC;
    uint32_t inc = tc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(inc, off); if (lane >= off) inc += y; }
    int pos = base + (int)(inc - tc);
    if (tc > 2) {
        STREAM_SWEEP(if (h < tau) { ccol[pos] = c; cval[pos] = raw; pos++; })
    } else {
        if (tc >= 1) { ccol[pos] = hc0; cval[pos] = hr0; }
        if (tc == 2) { ccol[pos + 1] = hc1; cval[pos + 1] = hr1; }
    }
    if (tid >= cnt && tid < KCU) { ccol[tid] = COLSENT; cval[tid] = 0.0f; }
    if (tid == KCU) { ccol[KCU] = COLSENT; cval[KCU] = tau; }
    __syncthreads();
    sort_cache_row(ccol, cval, tau);
    return g;
}

__global__ __launch_bounds__(BLOCK2) void build_row_caches_stream(int n, int64_t ld, const float *__restrict__ cost,
                                                                 const float *v, uint32_t *__restrict__ cache_col,
                                                                 float *__restrict__ cache_val, const int32_t *__restrict__ rowmap,
                                                                 const int32_t *__restrict__ same_prev) {
    __shared__ Scratch2 s;
    int par = 0;
    float delta = 0.0f, tau_guess = INFINITY;
    const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x, i_end = min(n, ((int)blockIdx.x + 1) * per);      // (see build_row_caches)
    for (int i = (int)blockIdx.x * per; i < i_end; i++) {
        if (same_prev && same_prev[i]) continue;                          // a copy of the previous row: replicate_group_caches
        (void)refresh_row_stream<0>(i, n, ld, RBASE(cost, rowmap, i, ld), v, cache_col, cache_val, delta, tau_guess, s, par);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The full-chip cache build as ONE WAVE PER ROW (round 4; what the build before and between the solver's phases runs).  The
// workgroup forms above hold a row in 187 VGPRs (CH = 10: one workgroup per CU, its loads never overlap its threshold search --
// 2.1 TB/s at n = 20 000) or sweep it several times between workgroup barriers (the streaming form: 3.0 TB/s at n = 50 000).
// Here a wave streams its row ONCE with U quads per lane in flight and no barrier anywhere.  Rows of >= 256 U columns: a streaming
// selection without a guess (cb_stream below: the columns under a falling threshold are compacted into a 768-byte staging line of the
// wave by ballot as they pass).  Shorter rows, and rows the selection gives up on (ties, an adversarial column order): the floor is
// guessed (the floor of the wave's previous row; ANY floor that admits <= 63 columns makes a valid cache) and the columns under it are
// staged during the sweep; a guess that does not fit is replaced by the 35th smallest of the row's 64 lane minima (49 +- 5 columns lie
// below it whatever the distribution) and the (L2 / MALL resident) row is swept again; a bisection as before when that fails too.
// Same cache contract as refresh_row.
constexpr int CBW = 4;             // waves per workgroup (they share nothing but the launch)
struct CbStage { uint32_t col[KC]; float val[KC]; float h[KC]; };

// one sweep of the row: counts the columns with h < tau and stages the first 64 of them; TRACK: also the lane minima
template <int U, bool TRACK>
__device__ __forceinline__ int cb_sweep(const __amdgpu_buffer_rsrc_t rrow, const __amdgpu_buffer_rsrc_t rv, const float *__restrict__ row,
                                        const float *__restrict__ v, int n, int lane, float tau, float &lmin, CbStage &st) {
    const int nfull = n >> 2, ntail = n & 3;
    const uint64_t lt = (1ull << lane) - 1ull;
    int cnt = 0;
    auto one = [&](uint32_t c, float raw, bool p) {                // (convergent: every lane of the wave calls it)
        const uint64_t m = __ballot(p);
        if (m) {
            const int pos = cnt + __popcll(m & lt);
            if (p && pos < KC) { st.col[pos] = c; st.val[pos] = raw; }
            cnt += __popcll(m);
        }
    };
    auto quad = [&](int q, const u32x4_t &xr, const u32x4_t &vr, bool inrange) {
        const float r0 = __uint_as_float(xr.x), r1 = __uint_as_float(xr.y), r2 = __uint_as_float(xr.z), r3 = __uint_as_float(xr.w);
        const float h0 = r0 - __uint_as_float(vr.x), h1 = r1 - __uint_as_float(vr.y), h2 = r2 - __uint_as_float(vr.z),
                    h3 = r3 - __uint_as_float(vr.w);
        float m4 = fminf(fminf(h0, h1), fminf(h2, h3));
        if (!inrange) m4 = INFINITY;
        if (TRACK) lmin = fminf(lmin, m4);
        if (__ballot(m4 < tau)) {
            const uint32_t c0 = (uint32_t)q * 4u;
            one(c0, r0, inrange && h0 < tau); one(c0 + 1, r1, inrange && h1 < tau);
            one(c0 + 2, r2, inrange && h2 < tau); one(c0 + 3, r3, inrange && h3 < tau);
        }
    };
    int base = 0;
    for (; base + 64 * U <= nfull; base += 64 * U) {                // (wave-uniform trips; all U quads of every lane in range)
        u32x4_t xr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rrow, (base + 64 * u + lane) * 16, 0, 0);
            vr[u] = __builtin_amdgcn_raw_buffer_load_b128(rv, (base + 64 * u + lane) * 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) quad(base + 64 * u + lane, xr[u], vr[u], true);
    }
    for (; base < nfull; base += 64) {                               // (reads past the descriptors' extents give zeros)
        const int q = base + lane;
        const u32x4_t xr = __builtin_amdgcn_raw_buffer_load_b128(rrow, q * 16, 0, 0);
        const u32x4_t vr = __builtin_amdgcn_raw_buffer_load_b128(rv, q * 16, 0, 0);
        quad(q, xr, vr, q < nfull);
    }
    if (ntail) {                                                     // the ragged last quad: a column per lane
        const int c = nfull * 4 + lane;
        const bool valid = lane < ntail;
        const float raw = valid ? row[c] : 0.0f;
        const float h = valid ? raw - v[c] : INFINITY;
        if (TRACK) lmin = fminf(lmin, h);
        one((uint32_t)c, raw, valid && h < tau);
    }
    return cnt;
}

// ascending bitonic sort of one float per lane
__device__ __forceinline__ float cb_sort64(float x, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float o = __shfl_xor(x, j);
            const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
            x = take_min ? fminf(x, o) : fmaxf(x, o);
        }
    }
    return x;
}
// the staged set (cnt <= 64 entries) cut down to the columns below its (K + 1)-th smallest reduced cost, which becomes the threshold
__device__ __forceinline__ void cb_compress(CbStage &st, int lane, int K, int &cnt, float &T) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t c = lane < cnt ? st.col[lane] : COLSENT;
    const float r = lane < cnt ? st.val[lane] : 0.0f;
    const float h = lane < cnt ? st.h[lane] : INFINITY;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float srt = cb_sort64(h, lane);
    const float Tn = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(srt), K));
    if (!(Tn < INFINITY)) return;                                  // no more than K entries: nothing to drop
    T = Tn;
    const bool keep = h < Tn;
    const uint64_t m = __ballot(keep);
    if (keep) { const int p = __popcll(m & ((1ull << lane) - 1ull)); st.col[p] = c; st.val[p] = r; st.h[p] = h; }
    cnt = __popcll(m);
}

// The row in ONE sweep without a guess (rows of >= 256 U columns): the first U quads of every lane (read twice) give 64 lane minima, the 17th
// smallest of them (in a long row; up to the 35th in a row of little more than 256 U columns) is the first threshold T (~19 of those
// 256 U columns lie below it, whatever the distribution); from then on every
// column below T is staged as it passes, and when the 64 staging slots run over the staged set is cut down to the columns below its
// 25th smallest reduced cost, which becomes T (a streaming selection: T only falls, so nothing that was passed over could qualify
// later; two or three cuts per row, each keeping what leaves ~54 columns at the row's end).  At the end the stage holds EVERY column
// below T, 16 ... 63 of them (~52 on average).  false: the slots
// ran over twice in one step (an adversarial order of the columns) or fewer than 16 columns are left (ties) -- the caller falls
// back on the sweeps of the search above, for which the lane minima of the whole row are returned in any case.
template <int U>
__device__ __forceinline__ bool cb_stream(const __amdgpu_buffer_rsrc_t rrow, const __amdgpu_buffer_rsrc_t rv, const float *__restrict__ row,
                                          const float *__restrict__ v, int n, int lane, float &T_out, int &cnt_out, float &lmin, CbStage &st) {
    const int nfull = n >> 2, ntail = n & 3;
    const uint64_t lt = (1ull << lane) - 1ull;
    int cnt = 0, base = 0;                                           // base: the quad the sweep has reached (wave-uniform)
    float T = INFINITY;
    bool fail = false;
    auto one = [&](uint32_t c, float raw, float h, bool valid) {     // (convergent: every lane of the wave calls it)
        bool p = valid && h < T;
        uint64_t m = __ballot(p);
        if (!m) return;
        if (cnt + __popcll(m) > KC) {
            // how many to keep: the columns still to come bring K (n - m) / m more below the new threshold when m columns have passed
            // (any order of the columns but an adversarial one), so K = 54 m / n leaves ~54 at the end of the row -- a cut early in
            // the row is a deep one, but never below 24: the threshold that leaves 8 of 7 000 columns is known to +-35 %, and rows ended
            // with anything from 25 to 63 columns (50 000^2: 356 full-row bids instead of 146).  (Cutting to 24 wherever the slots ran over left 16 ... 63, 40 on average: the build was faster
            // and the solve slower -- 50 000^2: 1 585 instead of 146 full-row bids, row reduction 20.5 -> 25 ms.)
            const int K = (int)min(48ll, max(24ll, 54ll * (4ll * base) / (long long)n));
            cb_compress(st, lane, K, cnt, T);
            p = p && h < T;
            m = __ballot(p);
            if (!m) return;
            if (cnt + __popcll(m) > KC) { fail = true; return; }
        }
        if (p) { const int pos = cnt + __popcll(m & lt); st.col[pos] = c; st.val[pos] = raw; st.h[pos] = h; }
        cnt += __popcll(m);
    };
    auto quad = [&](int q, const u32x4_t &xr, const u32x4_t &vr, bool inrange) {
        const float r0 = __uint_as_float(xr.x), r1 = __uint_as_float(xr.y), r2 = __uint_as_float(xr.z), r3 = __uint_as_float(xr.w);
        const float h0 = r0 - __uint_as_float(vr.x), h1 = r1 - __uint_as_float(vr.y), h2 = r2 - __uint_as_float(vr.z),
                    h3 = r3 - __uint_as_float(vr.w);
        float m4 = fminf(fminf(h0, h1), fminf(h2, h3));
        if (!inrange) m4 = INFINITY;
        lmin = fminf(lmin, m4);
        if (__ballot(m4 < T)) {
            const uint32_t c0 = (uint32_t)q * 4u;
            one(c0, r0, h0, inrange); one(c0 + 1, r1, h1, inrange); one(c0 + 2, r2, h2, inrange); one(c0 + 3, r3, h3, inrange);
        }
    };
    {   // the first U quads of every lane: lane minima and the first threshold; the sweep below starts over with them (8 KB that the
        // wave has just read: keeping them in registers across the sort cost a third of the occupancy)
        float m0 = INFINITY;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32x4_t xr = __builtin_amdgcn_raw_buffer_load_b128(rrow, (64 * u + lane) * 16, 0, 0);
            const u32x4_t vr = __builtin_amdgcn_raw_buffer_load_b128(rv, (64 * u + lane) * 16, 0, 0);
            m0 = fminf(m0, fminf(fminf(__uint_as_float(xr.x) - __uint_as_float(vr.x), __uint_as_float(xr.y) - __uint_as_float(vr.y)),
                                 fminf(__uint_as_float(xr.z) - __uint_as_float(vr.z), __uint_as_float(xr.w) - __uint_as_float(vr.w))));
        }
        const float srt = cb_sort64(m0, lane);
        // which lane minimum: ~c0 of the first 256 U columns are wanted below it, c0 = 54 * 256 U / n (the rest of the row brings its
        // share), 18 at least and 49 at most (a row of just 256 U columns: the 35th smallest, the two-sweep form's choice); c columns lie
        // below the (i + 1)-th smallest of 64 lane minima for i = 64 (1 - exp(-(c + 1) / 64)) (the draws it takes to hit i + 1 lanes)
        const float c0 = fminf(49.0f, fmaxf(18.0f, 54.0f * (float)(256 * U) / (float)n));
        const int i0 = __builtin_amdgcn_readfirstlane((int)(64.0f * (1.0f - __expf(-(c0 + 1.0f) * 0.015625f))));
        T = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(srt), i0));
    }
    for (; base + 64 * U <= nfull; base += 64 * U) {
        u32x4_t xr[U], vr[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rrow, (base + 64 * u + lane) * 16, 0, 0);
            vr[u] = __builtin_amdgcn_raw_buffer_load_b128(rv, (base + 64 * u + lane) * 16, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) quad(base + 64 * u + lane, xr[u], vr[u], true);
    }
    for (; base < nfull; base += 64) {
        const int q = base + lane;
        const u32x4_t xr = __builtin_amdgcn_raw_buffer_load_b128(rrow, q * 16, 0, 0);
        const u32x4_t vr = __builtin_amdgcn_raw_buffer_load_b128(rv, q * 16, 0, 0);
        quad(q, xr, vr, q < nfull);
    }
    if (ntail) {
        const int c = nfull * 4 + lane;
        const bool valid = lane < ntail;
        const float raw = valid ? row[c] : 0.0f;
        const float h = valid ? raw - v[c] : INFINITY;
        lmin = fminf(lmin, h);
        one((uint32_t)c, raw, h, valid);
    }
    if (!fail && cnt > KCU) cb_compress(st, lane, 48, cnt, T);     // (64 staged: the cache has 63 slots)
    T_out = T; cnt_out = cnt;
    return !fail && cnt <= KCU && cnt >= 16 && T < INFINITY;
}

template <int U>
__global__ __launch_bounds__(64 * CBW) void build_row_caches_wave(int n, int64_t ld, const float *__restrict__ cost,
                                                                 const float *__restrict__ v, uint32_t *__restrict__ cache_col,
                                                                 float *__restrict__ cache_val, const int32_t *__restrict__ rowmap,
                                                                 const int32_t *__restrict__ same_prev, int stream, int keep_min) {
    __shared__ CbStage stage[CBW];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (scalar: row bases and descriptors stay in SGPRs)
    CbStage &st = stage[w];
    const int gw = (int)blockIdx.x * CBW + w, nw = (int)gridDim.x * CBW;
    const int per = (n + nw - 1) / nw, i_end = min(n, (gw + 1) * per);      // (a contiguous range of rows per wave)
    const int nquad = (n + 3) >> 2;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(v), 0, nquad * 16, 0x00020000);
    float tau_guess = INFINITY, delta = 0.0f;
    for (int i = gw * per; i < i_end; i++) {
        if (same_prev && same_prev[i]) continue;                          // a copy of the previous row: replicate_group_caches
        if (keep_min > 0) {
            // A REBUILD (keep_min > 0: the caches hold an earlier build): prices only fall, so a floor stays a lower bound of every
            // uncached column for good and a cache is as good as the number of its columns that still lie below its floor -- two
            // certify a bid, one a relaxation.  A row that has keep_min of them is left alone: 512 bytes and 63 price gathers
            // instead of the row.  (A developer's knob, CYTO_CACHE_KEEP, off by default: by the time a rebuild is due hardly a row
            // has 32 such columns left -- nothing is skipped -- and leaving rows with 2 ... 16 alone brings more full-row bids than
            // it saves: few-cell-type 20 000^2 17.7 -> 29.0 / 23.8 / 19.8 / 18.9 ms; the 256-chunk batch is flat.  DESIGN "Tried".)
            const uint32_t kc0 = cache_col[(int64_t)i * KC + lane];
            const float kv0 = cache_val[(int64_t)i * KC + lane];
            const float fl0 = __shfl(kv0, KCU);
            const bool below = lane < KCU && kc0 != COLSENT && (kv0 - v[kc0]) < fl0;
            if (__popcll(__ballot(below)) >= keep_min) continue;
        }
        const float *__restrict__ row = cost + row_off(rowmap, i, ld);
        const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, (int)(ld * 4), 0x00020000);
        float tau0 = tau_guess;
        float lmin = INFINITY, dummy = 0.0f, tau = tau0;
        int cnt = 0;
        bool have = false;
        if (stream && (n >> 2) >= 64 * U) {                                // one sweep, no guess (cb_stream); else the lane minima are complete
            have = cb_stream<U>(rrow, rv, row, v, n, lane, tau, cnt, lmin, st);
            if (!have) { tau0 = INFINITY; tau = INFINITY; cnt = 0; }
        } else {
            cnt = cb_sweep<U, true>(rrow, rv, row, v, n, lane, tau0 < INFINITY ? tau0 : -INFINITY, lmin, st);
            have = tau0 < INFINITY && cnt <= KCU && (cnt >= KCU / 2 || cnt >= n);      // the guess fits: one sweep
        }
        if (!have) {
            const float umin = ord2f(wave_min_u32(f2ord(lmin)));
            float lo = 0.0f, hi = INFINITY;
            bool okc = false;
            {   // the 35th smallest of the 64 lane minima: 49 +- 5 columns lie below it whatever the distribution (distinct values;
                //  lap_wide.hip, SC_FLOOR_POS) -- the bisection that follows is for rows with ties
                float lm = lmin;
#pragma unroll
                for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        const float o = __shfl_xor(lm, j);
                        const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
                        lm = take_min ? fminf(lm, o) : fmaxf(lm, o);
                    }
                }
                const float m34 = __shfl(lm, 34);
                if (m34 < INFINITY && m34 > umin) delta = m34 - umin;
                else if (tau0 < INFINITY && tau0 > umin && cnt > 0) {     // (fewer than 35 lanes have columns) scale the guess by its count
                    const float d0 = tau0 - umin;
                    if (cnt > KCU) { hi = d0; delta = d0 * fmaxf(0.0625f, 0.75f * (float)KCU / (float)cnt); }
                    else { lo = d0; delta = d0 * fminf(16.0f, 0.75f * (float)KCU / (float)cnt); }
                }
            }
            if (!(delta > 0.0f) || !(delta < 1e30f)) delta = 1e-3f;
            for (int it = 0; it < 24 && !okc; it++) {
                tau = umin + delta;
                cnt = cb_sweep<U, false>(rrow, rv, row, v, n, lane, tau, dummy, st);
                if (cnt > KCU) {
                    hi = delta;
                    const float mid = (lo > 0.0f) ? 0.5f * (lo + hi) : 0.5f * delta;
                    if (!(mid < hi) || !(mid > lo)) break;               // cannot separate: too many ties just above umin
                    delta = mid;
                } else if (cnt < KCU / 2 && cnt < n && delta < 1e30f) {
                    lo = delta;
                    const float mid = (hi < INFINITY) ? 0.5f * (lo + hi) : 2.0f * delta;
                    if (hi < INFINITY && (!(mid < hi) || !(mid > lo))) { okc = true; break; }      // best separable threshold
                    delta = mid;
                } else {
                    okc = true;
                }
            }
            if (!okc || cnt > KCU) {
                // the largest threshold known to admit <= KCU columns (possibly none), staged once more
                if (lo > 0.0f) { delta = lo; tau = umin + lo; } else { tau = -INFINITY; }
                cnt = cb_sweep<U, false>(rrow, rv, row, v, n, lane, tau, dummy, st);
                if (cnt > KCU) { tau = -INFINITY; cnt = 0; }
            }
        }
        tau_guess = tau > -INFINITY ? tau : INFINITY;
        // the staged columns, sorted by column (unused slots last), the floor in slot 63
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t kc = lane < cnt ? st.col[lane] : COLSENT;
        float kv = lane < cnt ? st.val[lane] : 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t pk = (uint32_t)__shfl_xor((int)kc, j);
                const float pv = __shfl_xor(kv, j);
                const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
                const bool sw = take_min ? (pk < kc) : (pk > kc);
                kc = sw ? pk : kc; kv = sw ? pv : kv;
            }
        }
        if (lane == KCU) { kc = COLSENT; kv = tau; }               // (at most 63 entries: lane 63 held a sentinel)
        cache_col[(int64_t)i * KC + lane] = kc;
        cache_val[(int64_t)i * KC + lane] = kv;
    }
}

// Argument block of one problem (kernels get an array of them: one workgroup per problem).  Fields through an X-macro because
// the kernels read the block through a mirror struct whose pointers are typed as GLOBAL (LOAD_ARGS): pointers that are loaded from
// memory are generic to the compiler, and every access through them would be a FLAT instruction instead of a global one.
//   fws        float workspace: v[n] | u[n] | sumvd[n] | cassign[n] (= c[colsol[j]][j]) | 2n more
//   iws        int workspace: rowsol | colsol | matches | freerows | rtrows | pred | colgroup | ...   (n each)
//   cache_*    [n][KC] row caches;  misc: +8 double total; +16 long long counters[]; +4 int status
//   rowgid     [n] duplicate-row group of every row (consecutive identical rows share an id)
//   rowmap     [n] stored row of every LAP row, or nullptr (row i is stored row i)
//   g_hbest / g_hstamp   [ngroups] scratch for gmode 2 (stamps zeroed)
//   gmode      0: no duplicate rows; 1: per-group state in LDS; 2: in global memory
//   auxlds     augmentation: cassign (f32) + colgroup (u16) per column also in LDS
//   aug_start  jv_aug2: first free row to augment (> 0: continues after jv_aug_lazy gave up)
#define CHAIN2_FIELDS(P, S)                                                                                             \
    S(int, n) S(int64_t, ld) P(const float, cost) P(float, fws) P(int32_t, iws) P(uint32_t, cache_col) P(float, cache_val)   \
    P(char, misc) P(int32_t, rowgid) P(const int32_t, rowmap) P(float, g_hbest) P(int32_t, g_hstamp)                        \
    S(int, ngroups) S(int, gmode) S(int, auxlds) S(int, aug_start)
#define F_PTR(T, name) T *name;
#define F_GPTR(T, name) __attribute__((address_space(1))) T *name;
#define F_VAL(T, name) T name;
#define F_COPY_PTR(T, name) a.name = (T *)g.name;
#define F_COPY_VAL(T, name) a.name = g.name;
struct Chain2Args { CHAIN2_FIELDS(F_PTR, F_VAL) };
struct Chain2ArgsG { CHAIN2_FIELDS(F_GPTR, F_VAL) };
static_assert(sizeof(Chain2Args) == sizeof(Chain2ArgsG), "mirror layout");
__device__ __forceinline__ Chain2Args load_args(const Chain2Args *__restrict__ batch) {
    const Chain2ArgsG g = reinterpret_cast<const Chain2ArgsG *>(batch)[blockIdx.x];
    Chain2Args a;
    CHAIN2_FIELDS(F_COPY_PTR, F_COPY_VAL)
    return a;
}

// L2-coherent (agent-scope, relaxed) accesses to global state
__device__ __forceinline__ float ld_f32(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_f32(float *p, float x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// State accessors.  LDS_STATE: prices v (fp32) and colsol (u16, 0xFFFF = unassigned) live in LDS
// (n <= ~26k); otherwise they live in global memory and are accessed L2-coherently.
template <bool LDS_STATE> __device__ __forceinline__ float st_vget(const float *s_v, const float *gv, int j) {
    if constexpr (LDS_STATE) return s_v[j]; else return ld_f32(gv + j);
}
template <bool LDS_STATE> __device__ __forceinline__ void st_vset(float *s_v, float *gv, int j, float x) {
    if constexpr (LDS_STATE) s_v[j] = x; else st_f32(gv + j, x);
}
template <bool LDS_STATE> __device__ __forceinline__ int32_t st_csget(const uint16_t *s_cs, const int32_t *gcs, int j) {
    if constexpr (LDS_STATE) { const uint16_t c = s_cs[j]; return c == 0xFFFFu ? -1 : (int32_t)c; }
    else return ld_i32(gcs + j);
}
template <bool LDS_STATE> __device__ __forceinline__ void st_csset(uint16_t *s_cs, int32_t *gcs, int j, int32_t i) {
    if constexpr (LDS_STATE) s_cs[j] = (uint16_t)i; else st_i32(gcs + j, i);
}

template <int CH, bool LDS_STATE, int BS = BLOCK2>
__device__ __forceinline__ void load_vreg(const float *s_v, const float *gv, int n, int tid, float (&vreg)[CH * 4]) {
#pragma unroll
    for (int m = 0; m < CH; m++) {
        const int q = m * BS + tid;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q * 4 < n) {
            if constexpr (LDS_STATE) t = *reinterpret_cast<const float4 *>(s_v + q * 4);
            else {
                const int c = q * 4;
                t.x = ld_f32(gv + c);
                t.y = c + 1 < n ? ld_f32(gv + c + 1) : 0.f;
                t.z = c + 2 < n ? ld_f32(gv + c + 2) : 0.f;
                t.w = c + 3 < n ? ld_f32(gv + c + 3) : 0.f;
            }
        }
        vreg[m * 4 + 0] = t.x; vreg[m * 4 + 1] = t.y; vreg[m * 4 + 2] = t.z; vreg[m * 4 + 3] = t.w;
    }
}

template <int NWV = NW2>
__device__ __forceinline__ uint32_t wg_min_u32(uint32_t x, Scratch2 &s, int &par) {
    const uint32_t w0 = wave_min_u32(x);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t *buf = reinterpret_cast<uint32_t *>(&s.cnt[par][0]);
    if (lane == 0) buf[w] = w0;
    lds_barrier();
    uint32_t r = buf[lane & (NWV - 1)];
    r = row_min_u32(r);
    par ^= 1;
    return readlane32(r, 0);
}

#ifdef CYTO_AUG_TRACE
__device__ long long g_aug_trace[2 << 16];   // -DCYTO_AUG_TRACE (tools/trace_aug_scans.py): per search, cumulative scans and elided scans
#endif
// -DCYTO_AUG_PROF (tools/prof_aug_step.sh): s_memtime stamps of wave 0 inside a dense augmentation step
#ifdef CYTO_AUG_PROF
__device__ long long g_aug_prof[16];
#define AP_DECL long long ap_[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ap_t_ = 0;
#define AP_START ap_t_ = (long long)__builtin_amdgcn_s_memtime();
#define AP_STAMP(k) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long now_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ap_[k] += now_ - ap_t_; ap_t_ = now_; }
#define AP_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define AP_FLUSH if (threadIdx.x == 0) { for (int k_ = 0; k_ < 14; k_++) g_aug_prof[k_] += ap_[k_]; g_aug_prof[15] += 1; }
#else
#define AP_DECL
#define AP_START
#define AP_STAMP(k)
#define AP_WAITVM
#define AP_FLUSH
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fmin3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// One augmentation (all threads, uniform control): dense Dijkstra search from `freerow`, price
// update, path flip.  Same pick rule as the oracle: lexicographic minimum of (d, assigned?, column)
// over the unscanned columns.  Per column the owning lane keeps, in VGPRs, the distance d (+inf
// once the column is scanned) and the price vm (-inf once scanned: every later relaxation of that
// column is then a no-op).  The arg-min is found in two steps: the minimum value first (one 32-bit all-reduce
// on order-preserving keys), then the column among the few lanes that hold that value.
//
// A step's cost on one CU is (a) the round trip of the picked row and (b) the VALU work of relaxing n columns, so
// the relaxation is as lean as the oracle's arithmetic allows: per PAIR of columns two packed subtractions
// (v_pk_add_f32 with negated operands: fl(fl(c - v) - h), the oracle's two roundings), per column one v_min_f32,
// per four columns two v_min3_f32 for the running minimum -- 2.5 VALU operations per column.
// No predecessor is tracked while relaxing (it cost a compare and two selects per column).  The oracle's pred[j] is the
// FIRST scan, in scan order and with the free row's initial scan before all others, whose candidate equals the final
// d[j] (a later equal candidate never replaces it: updates are strict); so every scan logs (row, h), every retired
// column logs the distance it was scanned at, and the <= ~40 columns of the augmenting path find their predecessors
// afterwards by re-evaluating the logged candidates for that one column (reconstruct_pred).
// LEAN (the usual configuration: the per-column auxiliaries and the duplicate-row group state are in LDS): the pick's look-ups
// are plain LDS reads with no global-memory alternative compiled in -- the alternatives cost a branch each and a
// `s_waitcnt vmcnt(0)` at every join.
template <int CH, bool LDS_STATE, int BS, bool LEAN>
__device__ __forceinline__ int chain_augment(int n, int64_t ld, const float *__restrict__ cost, float *gv, float2 *sd,
                                             float *cassign, int32_t *rowsol, int32_t *gcolsol, int32_t *slog_row, float *slog_h,
                                             float *s_v, uint16_t *s_cs, int freerow, uint64_t validm, Scratch2 &s, int &par,
                                             long long &c_relax, long long &c_hops, long long &c_skipped, int gmode,
                                             const int32_t *rowgid, int32_t *colgroup, float *hb, int32_t *hs, int stamp, float *s_ca,
                                             uint16_t *s_cg, PickRec (*rec)[NW2], const int32_t *__restrict__ rowmap) {
    constexpr int NC = CH * 4;
    constexpr int NWV = BS / 64;
#undef SLOT_COL
#define SLOT_COL(sl) ((((sl) / 4) * BS + tid) * 4 + ((sl) % 4))
    const int tid = threadIdx.x;
    // Column state as two register VECTORS: a slot that is only known at run time, but is the same for the whole wave
    // (the retired column's, the picked column's), is then read or written with indirect register addressing
    // (s_set_gpr_idx / v_movrel: a handful of instructions) instead of a compare-and-select per slot.  The hardware indexes
    // at most 32 registers that way; beyond (NC > 32) the select chains remain.
    typedef float fvec __attribute__((ext_vector_type(NC)));
    constexpr bool VEC = NC <= 32;
    fvec vmV, dV;
    float cm[CH];
    // LDS views with their address space spelled out: through plain (generic) pointers that may also be null or point to
    // global memory the compiler emits FLAT loads, which take the long way round (measured: ~1000 cycles for the four
    // look-ups of a step)
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) int32_t lds_i32;
    lds_f32 *const l_v = (lds_f32 *)s_v;
    lds_u16 *const l_cs = (lds_u16 *)s_cs;
    lds_f32 *const l_ca = (lds_f32 *)s_ca;
    lds_u16 *const l_cg = (lds_u16 *)s_cg;
    lds_f32 *const l_hb = (lds_f32 *)hb;
    lds_i32 *const l_hs = (lds_i32 *)hs;
    typedef __attribute__((address_space(3))) PickRec lds_rec;
    lds_rec *const l_rec = (lds_rec *)&rec[0][0];          // [2][NW2]
#define VM(sl) vmV[sl]
#define DR(sl) dV[sl]
    {
        float v0[NC];
        load_vreg<CH, LDS_STATE, BS>(s_v, gv, n, tid, v0);
#pragma unroll
        for (int sl = 0; sl < NC; sl++) VM(sl) = v0[sl];
    }
    uint64_t assignedm = 0, scannedm = 0, readym = 0;
#pragma unroll
    for (int sl = 0; sl < NC; sl++)
        if (((validm >> sl) & 1) && st_csget<LDS_STATE>(s_cs, gcolsol, SLOT_COL(sl)) >= 0) assignedm |= (1ull << sl);
    {
        float4 x[CH];
        load_row4<CH, BS>(RBASE(cost, rowmap, freerow, ld), ld, freerow, n, tid, x);
#pragma unroll
        for (int sl = 0; sl < NC; sl++) {
            const bool ok = (validm >> sl) & 1;
            DR(sl) = ok ? vec_get<float>(x[sl / 4], sl % 4) - VM(sl) : INFINITY;
            VM(sl) = ok ? VM(sl) : -INFINITY;
        }
    }
#pragma unroll
    for (int m = 0; m < CH; m++) cm[m] = fmin_raw(fmin3_raw(DR(m * 4), DR(m * 4 + 1), DR(m * 4 + 2)), DR(m * 4 + 3));
    bool have = false;
    float curmin = 0.0f;
    int endofpath = -1;
    int nlog = 0;            // scans logged so far in this search (uniform)
    const int lane = tid & 63, wave = tid >> 6;
    AP_DECL
    AP_START
    for (;;) {
        // ---- pick: smallest d; among equal d an unassigned column first, then the lowest column.  ONE workgroup
        // exchange per step: every wave reduces its own columns to a candidate (value, then column among the lanes that
        // hold that value: two DPP all-reduces, no barrier), looks up what the step needs if that candidate wins (owner
        // row, its offset h, the duplicate-row skip test: independent broadcast LDS reads) and posts the record; after the
        // barrier every wave reduces the eight 64-bit keys and reads the winner's record. ----
        float lm = cm[0];
#pragma unroll
        for (int m = 1; m < CH; m++) lm = fmin_raw(lm, cm[m]);
        const uint32_t wmin = wave_min_u32(f2ord(lm));
        const float dminw = ord2f(wmin);
        AP_STAMP(8)
        uint32_t lk = 0xFFFFFFFFu;
        const uint64_t holders = __ballot(lm == dminw && dminw < INFINITY);
        bool one_group = false;
        int mg = CH - 1;
        if (VEC && __builtin_popcountll(holders) == 1) {
            // the usual case: ONE lane of the wave holds the minimum, in ONE of its groups of four.  (Two groups of the lane at the
            // same value go the general way below: an unassigned column of the later group beats an assigned one of the first.)
            int ng = cm[CH - 1] == dminw ? 1 : 0;
#pragma unroll
            for (int m = CH - 2; m >= 0; m--) { const bool eq = cm[m] == dminw; mg = eq ? m : mg; ng += eq ? 1 : 0; }   // first group that holds the value
            one_group = __builtin_amdgcn_readlane(ng, (int)__builtin_ctzll(holders)) == 1;
        }
        if (one_group) {
            // The group (found with CH compares) is broadcast, the four slots are fetched with indirect addressing, the holder picks among them.
            const int hl = (int)__builtin_ctzll(holders);
            const int mgu = __builtin_amdgcn_readlane(mg, hl);                   // wave-uniform
            if (lm == dminw) {
#pragma unroll
                for (int e = 3; e >= 0; e--) {
                    const int sl = mgu * 4 + e;
                    const float de = dV[sl];
                    const uint32_t key = (uint32_t)(((mgu * BS + tid) << 2) + e) | (uint32_t)((assignedm >> sl) & 1) << 31;
                    lk = de == dminw ? umin32(lk, key) : lk;
                }
            }
        } else if (lm == dminw && dminw < INFINITY) {
#pragma unroll
            for (int m = 0; m < CH; m++) {
                if (cm[m] == dminw) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int sl = m * 4 + e;
                        if (DR(sl) == dminw)
                            lk = umin32(lk, (uint32_t)SLOT_COL(sl) | (((assignedm >> sl) & 1) ? 0x80000000u : 0u));
                    }
                }
            }
        }
        AP_STAMP(9)
        const uint32_t wlk = wave_min_u32(lk);
        AP_STAMP(10)
        {
            int32_t iw = -1, gw = 0, skipw = 0, sroww = 0;
            float hw = 0.0f, vjpw = 0.0f;
            if (wlk != 0xFFFFFFFFu && (wlk & 0x80000000u)) {          // (wave-uniform) an assigned column: its owner row
                const int jpw = (int)(wlk & 0x7FFFFFFFu);
                float cipw;                                                     // c[i][jp]
                if constexpr (LEAN) { cipw = l_ca[jpw]; gw = (int)l_cg[jpw]; }
                else { cipw = s_ca ? l_ca[jpw] : ld_f32(cassign + jpw); gw = gmode ? (s_cg ? (int)l_cg[jpw] : ld_i32(colgroup + jpw)) : 0; }
                if constexpr (LDS_STATE) { const uint16_t c16 = l_cs[jpw]; iw = c16 == 0xFFFFu ? -1 : (int32_t)c16; vjpw = l_v[jpw]; }
                else { iw = ld_i32(gcolsol + jpw); vjpw = ld_f32(gv + jpw); }
                hw = (cipw - vjpw) - dminw;
                // the stored row of the owner (row map): looked up here, by every wave for its own candidate and next to the LDS
                // look-ups, not after the exchange -- there it was a dependent round trip to L2 in front of every row fetch
                sroww = iw;
                if (rowmap && iw >= 0) sroww = rowmap[__builtin_amdgcn_readfirstlane(iw)];
                // Exact skip for duplicated rows: if a bitwise identical row was already scanned in THIS search with an
                // offset hb >= h, every relaxation through row i is a no-op: fl(x - h) >= fl(x - hb) >= d[j] for every
                // unscanned column (prices do not change during a search and d only decreases).  The column is retired
                // exactly as usual, only the row read and the relaxation sweep are elided.
                if (gmode) {
                    float hbv; int hsv;
                    if (LEAN || gmode == 1) { hbv = l_hb[gw]; hsv = l_hs[gw]; }
                    else { hbv = ld_f32(hb + gw); hsv = ld_i32(hs + gw); }
                    skipw = ((hsv == stamp) && (hw <= hbv)) ? 1 : 0;
                }
            }
            AP_STAMP(11)
            if (lane == 0) {
                lds_rec *const dst = l_rec + par * NW2 + wave;
                dst->key = ((uint64_t)wmin << 32) | wlk; dst->row = iw; dst->h = hw; dst->vjp = vjpw; dst->g = gw; dst->skip = skipw; dst->srow = sroww;
            }
        }
        lds_barrier();
        AP_STAMP(12)
        uint64_t k8 = l_rec[par * NW2 + (lane & (NWV - 1))].key;
        uint64_t kmin = k8;
        kmin = umin64(kmin, dpp64<0xB1>(kmin)); kmin = umin64(kmin, dpp64<0x4E>(kmin));
        if constexpr (NWV > 4) kmin = umin64(kmin, dpp64<0x141>(kmin));
        kmin = readlane64(kmin, 0);
        const int wstar = (int)__builtin_ctzll(__ballot(k8 == kmin)) & (NWV - 1);     // keys of distinct waves are distinct columns
        const lds_rec *const win = l_rec + par * NW2 + wstar;
        const int32_t rw_row = win->row, rw_g = win->g, rw_skip = win->skip, rw_srow = win->srow;
        const float rw_h = win->h, rw_vjp = win->vjp;
        par ^= 1;
        AP_STAMP(0)
        const float dmin = ord2f((uint32_t)(kmin >> 32));
        const uint32_t g = (uint32_t)kmin;
        if (g == 0xFFFFFFFFu || !(dmin < INFINITY)) return CYTO_ERR_INTERNAL;
        const int jp = (int)(g & 0x7FFFFFFFu);
        if (!have || dmin != curmin) { readym |= scannedm; curmin = dmin; have = true; }
        if (!(g & 0x80000000u)) { endofpath = jp; break; }
        const int i = __builtin_amdgcn_readfirstlane(rw_row);
        const float h = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rw_h)));
        const float vjp = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rw_vjp)));
        const bool skip = __builtin_amdgcn_readfirstlane(rw_skip) != 0;
        if (gmode && !skip && lane == 0) {
            // the group's new best offset.  Every wave posts the same value: its own later look-ups (program order) see it
            // without another barrier, whichever wave is first
            const int grp = rw_g;
            if (LEAN || gmode == 1) { l_hb[grp] = h; l_hs[grp] = stamp; }
            else { st_f32(hb + grp, h); st_i32(hs + grp, stamp); }
        }
        AP_STAMP(1)
        float4 x[CH];
        if (!skip) {
            const int srow = __builtin_amdgcn_readfirstlane(rw_srow);
            load_row4<CH, BS>(cost + ((int64_t)srow - i) * ld, ld, i, n, tid, x);
            // scan log (after the loads in program order: a store does not hold them up); one store instruction of the last wave
            if (tid >= BS - 2) {
                if (tid == BS - 2) st_i32(slog_row + nlog, i);
                else st_f32(slog_h + nlog, h);
            }
            nlog++;
        }
        AP_STAMP(2)
        {   // retire column jp: remember v+d (price update) and d (predecessor search), mask the column out.
            // The slot is wave-uniform: a scalar branch per slot instead of two selects per slot.
            const int q = jp >> 2;
            const int sj = (q / BS) * 4 + (jp & 3);   // uniform
            const bool own = (q % BS) == tid;
            if (own) { scannedm |= (1ull << sj); sd[jp] = make_float2(vjp + dmin, dmin); }
            if constexpr (VEC) {   // sj is wave-uniform: indirect register addressing
                const float vo = vmV[sj], dq = dV[sj];
                vmV[sj] = own ? -INFINITY : vo;
                dV[sj] = own ? INFINITY : dq;
            } else {
#pragma unroll
                for (int sl = 0; sl < NC; sl++) {
                    const bool hit = own && (sl == sj);
                    VM(sl) = hit ? -INFINITY : VM(sl);
                    DR(sl) = hit ? INFINITY : DR(sl);
                }
            }
        }
        AP_STAMP(3)
        if (skip) {
#pragma unroll
            for (int m = 0; m < CH; m++) cm[m] = fmin_raw(fmin3_raw(DR(m * 4), DR(m * 4 + 1), DR(m * 4 + 2)), DR(m * 4 + 3));
            c_relax++;
            c_skipped++;
            AP_STAMP(6)
            continue;
        }
        AP_WAITVM
        AP_STAMP(4)
        const f32x2 hh = {h, h};
#pragma unroll
        for (int m = 0; m < CH; m++) {
            const f32x2 x01 = {x[m].x, x[m].y}, x23 = {x[m].z, x[m].w};
            const f32x2 v01 = {vmV[4 * m], vmV[4 * m + 1]}, v23 = {vmV[4 * m + 2], vmV[4 * m + 3]};
            const f32x2 a = (x01 - v01) - hh, b = (x23 - v23) - hh;
            dV[4 * m] = fmin_raw(dV[4 * m], a[0]);
            dV[4 * m + 1] = fmin_raw(dV[4 * m + 1], a[1]);
            dV[4 * m + 2] = fmin_raw(dV[4 * m + 2], b[0]);
            dV[4 * m + 3] = fmin_raw(dV[4 * m + 3], b[1]);
            cm[m] = fmin_raw(fmin3_raw(dV[4 * m], dV[4 * m + 1], dV[4 * m + 2]), dV[4 * m + 3]);
            SCHED_FENCE();
        }
        c_relax++;
        AP_STAMP(5)
    }
    // ---- the augmenting path, from the end column back to the free row (prices are still the search's prices) ----
    __syncthreads();   // the scan log and the retired columns' records are complete
    AP_START
    {
        int ep = endofpath;
        float dep = curmin;                                   // the end column's distance is the final minimum
        for (;;) {
            const float vep = st_vget<LDS_STATE>(s_v, gv, ep);
            // candidates in the oracle's order: entry 0 = the free row's initial scan (offset 0: fl(x - 0) = x), entry k + 1
            // = logged scan k.  Up to 8 gathers of c[row][ep] in flight per lane: one pass for <= 4095 scans.
            uint32_t first = 0xFFFFFFFFu;
            float cfirst = 0.0f;
            for (int e0 = 0; e0 <= nlog; e0 += 8 * BS) {
                float cv[8], hv[8];
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int e = e0 + t * BS + tid;
                    const bool in = e >= 1 && e <= nlog;
                    const int row = in ? ld_i32(slog_row + e - 1) : freerow;
                    hv[t] = in ? ld_f32(slog_h + e - 1) : 0.0f;
                    cv[t] = cost[row_off(rowmap, row, ld) + ep];
                }
#pragma unroll
                for (int t = 7; t >= 0; t--) {
                    const int e = e0 + t * BS + tid;
                    if (e <= nlog && ((cv[t] - vep) - hv[t]) == dep) { first = umin32(first, (uint32_t)e); if (first == (uint32_t)e) cfirst = cv[t]; }
                }
                if (__syncthreads_or(first != 0xFFFFFFFFu)) break;   // a match in this pass precedes every later entry
            }
            const uint32_t f = wg_min_u32<NWV>(first, s, par);
            if (f == 0xFFFFFFFFu) return CYTO_ERR_INTERNAL;
            const int i = f == 0u ? freerow : __builtin_amdgcn_readfirstlane(ld_i32(slog_row + (int)f - 1));
            const int nxt = __builtin_amdgcn_readfirstlane(ld_i32(rowsol + i));
            __builtin_amdgcn_s_barrier();                     // every wave holds the old rowsol[i] before it is overwritten
            if (first == f) {                                 // the lane that evaluated the winning entry holds c[i][ep]
                st_csset<LDS_STATE>(s_cs, gcolsol, ep, i);
                if (LEAN || s_ca) l_ca[ep] = cfirst; else st_f32(cassign + ep, cfirst);
                if (gmode) { const int gi = rowgid[i]; if (LEAN || s_cg) l_cg[ep] = (uint16_t)gi; else st_i32(colgroup + ep, gi); }
                st_i32(rowsol + i, ep);
            }
            c_hops++;
            if (i == freerow) break;
            ep = nxt;
            dep = ld_f32(&sd[ep].y);                          // the distance this (scanned) column was retired at
        }
    }
    // price update: columns scanned at an earlier level than the final one.  (LDS prices: every lane updates its own
    // columns, nobody reads another lane's before the barrier below.)
#pragma unroll
    for (int sl = 0; sl < NC; sl++)
        if ((readym >> sl) & 1) st_vset<LDS_STATE>(s_v, gv, SLOT_COL(sl), ld_f32(&sd[SLOT_COL(sl)].x) - curmin);
    __syncthreads();
    AP_STAMP(7)
    AP_FLUSH
#undef VM
#undef DR
#undef SLOT_COL
#define SLOT_COL(sl) ((((sl) / 4) * BLOCK2 + tid) * 4 + ((sl) % 4))
    return 0;
}

enum { PH_RT = 0, PH_ARR = 1, PH_AUG = 2 };

// -DCYTO_ARR_PROF (tools/prof_arr_step.py): s_memtime stamps inside a cached ARR step (the loop runs on wave 0 alone)
#ifdef CYTO_ARR_PROF
__device__ long long g_arr_prof[16];
#define RP_DECL long long rp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rp_t_ = 0, rp_n_ = 0;
#define RP_START rp_t_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define RP_STAMP(k) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); rp_[k] += now_ - rp_t_; rp_t_ = now_; }
#define RP_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define RP_WAITLDS asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define RP_COUNT rp_n_++;
#define RP_FLUSH if (threadIdx.x == 0) { for (int k_ = 0; k_ < 8; k_++) g_arr_prof[k_] = rp_[k_]; g_arr_prof[8] = rp_n_; }
#else
#define RP_DECL
#define RP_START
#define RP_STAMP(k)
#define RP_WAITVM
#define RP_WAITLDS
#define RP_COUNT
#define RP_FLUSH
#endif

// CS_LDS (only meaningful with !LDS_STATE, n <= 65535): the prices are too many for LDS but colsol (u16) still fits;
// the chain then never stores colsol to global memory (on gfx9 a load is not returned before the stores issued
// ahead of it are acknowledged, so every global store in the step delays the next step's gathers).
template <int CH, bool LDS_STATE, bool CS_LDS = false>
__global__ __launch_bounds__(BLOCK2) void jv_chain2(const Chain2Args *__restrict__ batch) {
    const Chain2Args a = load_args(batch);       // one workgroup per problem of the batch
    constexpr int NC = CH * 4;
    constexpr bool CSL = LDS_STATE || CS_LDS;     // colsol lives in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ Scratch2 s;
    __shared__ int32_t s_app[64];          // staged appends to the next sweep's free-row list (wave 0)
    const int tid = threadIdx.x, lane = tid & 63;
    // provably wave-uniform wave id: the wave-0 state machine below then lives in SGPRs with scalar branches
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = a.n;
    const int64_t ld = a.ld;
    const float *__restrict__ cost = a.cost;
    float *gv = a.fws;
    int32_t *rowsol = a.iws, *gcolsol = a.iws + n, *matches = a.iws + 2 * (int64_t)n;
    int32_t *freerows = a.iws + 3 * (int64_t)n, *rtrows = a.iws + 4 * (int64_t)n;
    const int npad = (n + 3) & ~3;
    float *s_v = reinterpret_cast<float *>(dyn_lds);
    uint16_t *s_cs = reinterpret_cast<uint16_t *>(dyn_lds + (LDS_STATE ? (size_t)npad * 4 : 0));
    int par = 0;

    uint64_t validm = 0;
#pragma unroll
    for (int sl = 0; sl < NC; sl++) if (SLOT_COL(sl) < n) validm |= (1ull << sl);

    if constexpr (CSL) {
        for (int c = tid; c < npad; c += BLOCK2) {
            if constexpr (LDS_STATE) s_v[c] = c < n ? gv[c] : 0.0f;
            const int32_t cs = c < n ? gcolsol[c] : -1;
            s_cs[c] = cs < 0 ? (uint16_t)0xFFFFu : (uint16_t)cs;
        }
    }

    // ---- free-row list (matches == 0) and reduction-transfer list (matches == 1), ascending ----
    int numfree = 0, nrt = 0;
    {
        const int R = (n + BLOCK2 - 1) / BLOCK2;
        const int r0 = min(n, tid * R), r1 = min(n, r0 + R);
        int f = 0, g = 0;
        for (int i = r0; i < r1; i++) { const int mt = matches[i]; f += (mt == 0); g += (mt == 1); }
        int of = wg_exscan2(f, s, par, &numfree);
        int og = wg_exscan2(g, s, par, &nrt);
        for (int i = r0; i < r1; i++) {
            const int mt = matches[i];
            if (mt == 0) st_i32(freerows + of++, i);
            else if (mt == 1) st_i32(rtrows + og++, i);
        }
    }
    __syncthreads();

    float delta = 0.0f, tau_guess = INFINITY;
    int c_rt = 0, c_arr = 0, c_dense = 0;
    int c_free_cr = numfree, c_free_a1 = 0, c_free_a2 = 0;

    const int arr_budget = 1000 * n + 1000000;   // == JV_ARR_BUDGET(n) of the oracle (fits: n <= FAST_NMAX)
    // chain state (meaningful in wave 0 only; uniform there)
    int phase = (n > 1) ? PH_RT : PH_ARR;
    int k = 0, sweep = 0, prev = numfree, carry = -1, cur_i = -1;
    if (phase == PH_ARR) numfree = 0;
    bool have_dense = false;
    K2 gd; gd.m1 = KEYMAX; gd.m2 = KEYMAX;
    int napp = 0;                           // entries staged in s_app since the last flush
    RP_DECL

    for (;;) {
        if (wave == 0) {
            // wave 0 runs cached steps until it needs the whole workgroup (a dense re-scan) or is done;
            // it then posts a command and falls through to the barrier below.
            for (;;) {
                // ---------- slow path: choose the next row (phase changes, new chains, budget) ----------
                if (!have_dense) {
                    if (phase == PH_RT) {
                        if (k >= nrt) { phase = PH_ARR; sweep = 0; k = 0; prev = numfree; numfree = 0; carry = -1; continue; }
                        cur_i = __builtin_amdgcn_readfirstlane(ld_i32(rtrows + k)); k++;
                    } else if (phase == PH_ARR) {
                        if (c_arr >= arr_budget && (carry >= 0 || k < prev)) {
                            // step budget exhausted (see oracle/jv_oracle_impl.h): hand the waiting rows
                            // to the augmentation phase, the displaced row first, then the list order
                            if (lane < (napp & 63)) st_i32(freerows + numfree - (napp & 63) + lane, s_app[lane]);
                            napp = 0;
                            if (carry >= 0) { if (lane == 0) st_i32(freerows + numfree, carry); numfree++; carry = -1; }
                            while (k < prev) {
                                const int r = __builtin_amdgcn_readfirstlane(ld_i32(freerows + k)); k++;
                                if (lane == 0) st_i32(freerows + numfree, r);
                                numfree++;
                            }
                            continue;
                        }
                        if (carry >= 0) { cur_i = carry; carry = -1; }
                        else if (k < prev) { cur_i = __builtin_amdgcn_readfirstlane(ld_i32(freerows + k)); k++; }
                        else if (sweep == 0) {
                            if (lane < (napp & 63)) st_i32(freerows + numfree - (napp & 63) + lane, s_app[lane]);   // flush the staged tail
                            napp = 0;
                            c_free_a1 = numfree; sweep = 1; k = 0; prev = numfree; numfree = 0; continue;
                        }
                        else {
                            if (lane < (napp & 63)) st_i32(freerows + numfree - (napp & 63) + lane, s_app[lane]);
                            napp = 0;
                            c_free_a2 = numfree; phase = PH_AUG; k = 0; continue;
                        }
                    } else {
                        // RT and ARR are done: the augmentation runs in its own kernel (jv_aug2)
                        if (lane == 0) { s.cmd_op = OP_EXIT; s.cmd_row = 0; }
                        break;
                    }
                }
                if (phase == PH_RT) {
                    // v[j1] -= min over j != j1 of (c[i][j] - v[j])
                    const int i = cur_i;
                    const int j1 = __builtin_amdgcn_readfirstlane(ld_i32(rowsol + i));
                    float mn;
                    if (have_dense) {
                        have_dense = false;
                        mn = ((int)(uint32_t)gd.m1 == j1) ? key_val(gd.m2) : key_val(gd.m1);
                    } else {
                        const uint32_t col = ld_u32(a.cache_col + (int64_t)i * KC + lane);
                        const float cv = ld_f32(a.cache_val + (int64_t)i * KC + lane);
                        const float F = __uint_as_float(readlane32(__float_as_uint(cv), KCU));
                        const bool valid = col != COLSENT && (int)col != j1;
                        const float vj = st_vget<LDS_STATE>(s_v, gv, valid ? (int)col : 0);
                        mn = ord2f(wave_min_u32(valid ? f2ord(cv - vj) : 0xFFFFFFFFu));
                        if (!(mn < F)) {
                            if (lane == 0) { s.cmd_op = OP_REFRESH; s.cmd_row = i; }
                            c_dense++;
                            break;
                        }
                    }
                    const float nv = st_vget<LDS_STATE>(s_v, gv, j1) - mn;
                    if (lane == 0) st_vset<LDS_STATE>(s_v, gv, j1, nv);
                    c_rt++;
                    continue;
                }
                // ---------- ARR: tight loop that follows one displacement chain ----------
                // Software pipelining: the row that will be displaced (the next row of the chain) is known
                // right after the first reduction, so its cache is requested then and arrives while the second
                // reduction and the bookkeeping of the current step run.
                bool need_dense = false;
                int pf_row = -1;
                uint32_t pf_col = 0;
                float pf_cv = 0.0f;
                RP_START
                for (;;) {
                    const int i = cur_i;
                    float umin, usub, vj1, cj1 = 0.0f, cj2 = 0.0f;
                    int j1, j2 = -1, i0, i02 = -1;
                    if (__builtin_expect(have_dense, 0)) {
                        have_dense = false;
                        umin = key_val(gd.m1); usub = key_val(gd.m2);
                        j1 = (int)(uint32_t)gd.m1; j2 = (int)(uint32_t)gd.m2;
                        cj1 = a.cost[row_off(a.rowmap, i, a.ld) + j1]; cj2 = a.cost[row_off(a.rowmap, i, a.ld) + j2];
                        vj1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st_vget<LDS_STATE>(s_v, gv, j1))));
                        i0 = __builtin_amdgcn_readfirstlane(st_csget<CSL>(s_cs, gcolsol, j1));
                        i02 = __builtin_amdgcn_readfirstlane(st_csget<CSL>(s_cs, gcolsol, j2));
                    } else {
                        uint32_t col;
                        float cv;
                        if (pf_row == i) { col = pf_col; cv = pf_cv; }          // requested during the previous step
                        else {
                            col = ld_u32(a.cache_col + (int64_t)i * KC + lane);
                            cv = ld_f32(a.cache_val + (int64_t)i * KC + lane);
                        }
                        RP_WAITVM
                        RP_STAMP(0)                                              // wait for the row's cache (L2)
                        const bool valid = col != COLSENT;
                        const float vj = st_vget<LDS_STATE>(s_v, gv, valid ? (int)col : 0);
                        const int32_t csj = st_csget<CSL>(s_cs, gcolsol, valid ? (int)col : 0);
                        const float F = __uint_as_float(readlane32(__float_as_uint(cv), KCU));
                        const uint32_t ord = valid ? f2ord(cv - vj) : 0xFFFFFFFFu;
                        RP_WAITLDS
                        RP_STAMP(1)                                              // LDS gathers of v and colsol
                        // minimum, its lane (ties: lowest column), then the minimum of the rest
                        const uint32_t o1 = wave_min_u32(ord);
                        const uint64_t m1 = __ballot(ord == o1);
                        const int l1 = __builtin_ctzll(m1);     // cache rows are sorted by column: lowest lane = lowest column
                        // the current owner of the best column is (almost always) the next row of the chain
                        pf_row = (int)readlane32((uint32_t)csj, l1);
                        if (pf_row >= 0) {
                            pf_col = ld_u32(a.cache_col + (int64_t)pf_row * KC + lane);
                            pf_cv = ld_f32(a.cache_val + (int64_t)pf_row * KC + lane);
                        }
                        RP_STAMP(2)                                              // first reduction, lane of the minimum, prefetch issue
                        const uint32_t o2 = wave_min_u32(lane == l1 ? 0xFFFFFFFFu : ord);
                        usub = ord2f(o2);
                        if (__builtin_expect(!(usub < F), 0)) {
                            if (lane == 0) { s.cmd_op = OP_REFRESH; s.cmd_row = i; }
                            c_dense++;
                            need_dense = true;
                            break;
                        }
                        umin = ord2f(o1);
                        j1 = (int)readlane32(col, l1);
                        vj1 = __uint_as_float(readlane32(__float_as_uint(vj), l1));
                        cj1 = __uint_as_float(readlane32(__float_as_uint(cv), l1));
                        i0 = (int)readlane32((uint32_t)csj, l1);
                        if (__builtin_expect(!((vj1 - (usub - umin)) < vj1) && i0 >= 0, 0)) {
                            const uint64_t m2 = __ballot(ord == o2 && lane != l1);
                            const int l2 = __builtin_ctzll(m2);
                            j2 = (int)readlane32(col, l2);
                            cj2 = __uint_as_float(readlane32(__float_as_uint(cv), l2));
                            i02 = (int)readlane32((uint32_t)csj, l2);
                        }
                    }
                    c_arr++;
                    RP_STAMP(3)                                                  // second reduction, read-lanes
                    RP_COUNT
                    const float vnew = vj1 - (usub - umin);
                    const bool lowers = vnew < vj1;
                    const bool swap = !lowers && i0 >= 0;
                    const int jj = swap ? j2 : j1;
                    const int i0f = swap ? i02 : i0;
                    (void)cj1; (void)cj2;
                    // rowsol is not read during ARR and equals the inverse of colsol: it is rebuilt after the chain
                    if (lane == 0) {
                        if (lowers) st_vset<LDS_STATE>(s_v, gv, j1, vnew);
                        st_csset<CSL>(s_cs, gcolsol, jj, i);
                    }
                    RP_WAITLDS
                    RP_STAMP(4)                                                  // price / colsol update
                    if (__builtin_expect(i0f >= 0 && lowers && c_arr < arr_budget, 1)) { cur_i = i0f; continue; }   // chain goes on
                    if (i0f >= 0) {
                        if (lowers) carry = i0f;       // budget reached: the slow path flushes it
                        else {
                            // the next sweep's list is staged in LDS and written 64 entries at a time: a global store in
                            // every step would hold back the next step's loads until it is acknowledged
                            if (lane == 0) s_app[napp & 63] = i0f;
                            napp++; numfree++;
                            if ((napp & 63) == 0) st_i32(freerows + numfree - 64 + lane, s_app[lane]);
                        }
                    }
                    break;
                }
                if (need_dense) break;
            }
        }
        __syncthreads();
        const int op = s.cmd_op, row = s.cmd_row;
        if (op == OP_EXIT) break;
        if constexpr (CH == 0) {
            gd = refresh_row_stream<0x10>(row, n, ld, RBASE(cost, a.rowmap, row, ld), gv, a.cache_col, a.cache_val, delta, tau_guess, s, par);
            have_dense = true;
        } else {
            float vreg[NC > 0 ? NC : 1];
            load_vreg<CH, LDS_STATE>(s_v, gv, n, tid, vreg);
            gd = refresh_row<CH>(row, n, ld, RBASE(cost, a.rowmap, row, ld), vreg, validm, a.cache_col, a.cache_val, delta, s, par);
            have_dense = true;
        }
    }

    // ---- write back prices and colsol for the augmentation kernel; rowsol = inverse of colsol ----
    float *cassign = a.fws + 3 * (int64_t)n;
    int32_t *colgroup = a.iws + 6 * (int64_t)n;
    __syncthreads();
    for (int c = tid; c < n; c += BLOCK2) {
        const int32_t r = st_csget<CSL>(s_cs, gcolsol, c);
        if constexpr (LDS_STATE) gv[c] = s_v[c];
        if constexpr (CSL) gcolsol[c] = r;
        if (r >= 0) {
            rowsol[r] = c;
            cassign[c] = cost[row_off(a.rowmap, r, ld) + c];          // c[colsol[j]][j] for the augmentation kernel
            colgroup[c] = a.gmode ? a.rowgid[r] : 0;               // duplicate-row group of the row that owns column c
        }
    }
    if (tid == 0) {
        long long *counters = reinterpret_cast<long long *>(a.misc + 16);
        counters[C_RT] = c_rt; counters[C_ARR] = c_arr;
        counters[C_FREE_CR] = c_free_cr; counters[C_FREE_A1] = c_free_a1; counters[C_FREE_A2] = c_free_a2;
        counters[C2_DENSE_REFRESH] = c_dense;
        *reinterpret_cast<int *>(a.misc + 128) = numfree;
    }
    RP_FLUSH
}

// AUGMENTATION + duals + total: one persistent workgroup, all lanes active (see chain_augment).
// (BS is a parameter because a 256-thread variant -- one wave per SIMD, twice the columns per lane -- was measured: not faster,
// a lone wave per SIMD exposes the latency of every dependent instruction.)
template <int CH, bool LDS_STATE, int BS = BLOCK2>
__global__ __launch_bounds__(BS) void jv_aug2(const Chain2Args *__restrict__ batch) {
    const Chain2Args a = load_args(batch);       // one workgroup per problem of the batch
#undef SLOT_COL
#define SLOT_COL(sl) ((((sl) / 4) * BS + tid) * 4 + ((sl) % 4))
    constexpr int NC = CH * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ Scratch2 s;
    __shared__ PickRec s_rec[2][NW2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    const int64_t ld = a.ld;
    const float *__restrict__ cost = a.cost;
    float *gv = a.fws;
    // per retired column (v + d, d) of the current search: the 2n floats behind cassign (jv_aug_lazy's distance words,
    // dead once this kernel runs); scan log: rows in the old predecessor slot, offsets in the old sumvd slot
    float2 *sd = reinterpret_cast<float2 *>(a.fws + 4 * (int64_t)n);
    float *slog_h = a.fws + 2 * (int64_t)n;
    int32_t *rowsol = a.iws, *gcolsol = a.iws + n;
    int32_t *freerows = a.iws + 3 * (int64_t)n, *slog_row = a.iws + 5 * (int64_t)n;
    const int npad = (n + 3) & ~3;
    float *s_v = reinterpret_cast<float *>(dyn_lds);
    uint16_t *s_cs = reinterpret_cast<uint16_t *>(dyn_lds + (size_t)npad * 4);
    int par = 0;
    uint64_t validm = 0;
#pragma unroll
    for (int sl = 0; sl < NC; sl++) if (SLOT_COL(sl) < n) validm |= (1ull << sl);
    if constexpr (LDS_STATE) {
        for (int c = tid; c < npad; c += BS) {
            s_v[c] = c < n ? gv[c] : 0.0f;
            const int32_t cs = c < n ? gcolsol[c] : -1;
            s_cs[c] = cs < 0 ? (uint16_t)0xFFFFu : (uint16_t)cs;
        }
    }
    // per-group best offset of the current search (duplicate-row skip): LDS if it fits (gmode 1)
    float *hb = a.g_hbest;
    int32_t *hs = a.g_hstamp;
    const int gmode = a.gmode;
    if (gmode == 1) {
        hb = reinterpret_cast<float *>(dyn_lds + (size_t)npad * 6);
        hs = reinterpret_cast<int32_t *>(dyn_lds + (size_t)npad * 6 + (size_t)a.ngroups * 4);
        for (int g = tid; g < a.ngroups; g += BS) hs[g] = 0;
    }
    float *cassign = a.fws + 3 * (int64_t)n;
    // per-column auxiliaries (assigned cost, owner's row group) in LDS when they fit (a.auxlds)
    float *s_ca = nullptr;
    uint16_t *s_cg = nullptr;
    if (a.aug_start > 0 && gmode) {
        // continuing after jv_aug_lazy, which keeps no per-column group array: rebuild it from the assignment
        int32_t *colgroup = a.iws + 6 * (int64_t)n;
        for (int c = tid; c < n; c += BS) { const int32_t r = gcolsol[c]; colgroup[c] = r >= 0 ? a.rowgid[r] : 0; }
        __syncthreads();
    }
    if (a.auxlds) {
        const size_t off = (size_t)npad * 6 + (gmode == 1 ? (size_t)a.ngroups * 8 : 0);
        s_ca = reinterpret_cast<float *>(dyn_lds + ((off + 15) & ~(size_t)15));
        s_cg = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(s_ca) + (size_t)npad * 4);
        const int32_t *colgroup = a.iws + 6 * (int64_t)n;
        for (int c = tid; c < n; c += BS) { s_ca[c] = cassign[c]; s_cg[c] = (uint16_t)(gmode ? colgroup[c] : 0); }
    }
    __syncthreads();
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    long long c_relax = 0, c_hops = 0, c_augs = 0, c_skipped = 0;
    long long rows0 = 0, base0 = 0;
    if (a.aug_start > 0) {   // carry the counters of the searches jv_aug_lazy completed
        const long long *cn = reinterpret_cast<const long long *>(a.misc + 16);
        c_relax = cn[C_AUG_RELAX]; c_hops = cn[C_HOPS]; c_augs = cn[C_AUGS]; c_skipped = cn[C2_AUG_SKIPPED];
        rows0 = cn[C_ROWS_READ]; base0 = c_augs + c_relax - c_skipped;
    }
    int err = 0;
    for (int f = a.aug_start; f < numfree && !err; f++) {
        const int freerow = __builtin_amdgcn_readfirstlane(ld_i32(freerows + f));
        if (LDS_STATE && a.auxlds && gmode != 2)
            err = chain_augment<CH, LDS_STATE, BS, true>(n, ld, cost, gv, sd, cassign, rowsol, gcolsol, slog_row, slog_h, s_v, s_cs, freerow, validm,
                                                         s, par, c_relax, c_hops, c_skipped, gmode, a.rowgid, a.iws + 6 * (int64_t)n, hb, hs, f + 1,
                                                         s_ca, s_cg, s_rec, a.rowmap);
        else
            err = chain_augment<CH, LDS_STATE, BS, false>(n, ld, cost, gv, sd, cassign, rowsol, gcolsol, slog_row, slog_h, s_v, s_cs, freerow, validm,
                                                          s, par, c_relax, c_hops, c_skipped, gmode, a.rowgid, a.iws + 6 * (int64_t)n, hb, hs, f + 1,
                                                          s_ca, s_cg, s_rec, a.rowmap);
#ifdef CYTO_AUG_TRACE
        if (threadIdx.x == 0 && f < (1 << 16)) { g_aug_trace[2 * f] = c_relax; g_aug_trace[2 * f + 1] = c_skipped; }
#endif
        c_augs++;
    }
    // ---- write back prices and colsol, then duals u and the total ----
    if constexpr (LDS_STATE) {
        for (int c = tid; c < n; c += BS) {
            gv[c] = s_v[c];
            const uint16_t cs = s_cs[c];
            gcolsol[c] = cs == 0xFFFFu ? -1 : (int32_t)cs;
        }
    }
    __syncthreads();
    float *gu = a.fws + n;
    double part = 0.0;
    for (int i = tid; i < n; i += BS) {
        const int j = ld_i32(rowsol + i);
        const float cij = cost[row_off(a.rowmap, i, ld) + j];
        const float vj = ld_f32(gv + j);
        gu[i] = cij - vj;
        part += (double)cij;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) s.sum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < BS / 64; w++) t += s.sum[w];
        *reinterpret_cast<double *>(a.misc + 8) = t;
        long long *counters = reinterpret_cast<long long *>(a.misc + 16);
        counters[C_AUG_INIT] = c_augs; counters[C_AUG_RELAX] = c_relax; counters[C_AUGS] = c_augs; counters[C_HOPS] = c_hops;
        counters[C_ROWS_READ] = a.aug_start > 0 ? rows0 + (c_augs + c_relax - c_skipped - base0)
                                                : counters[C2_DENSE_REFRESH] + c_augs + c_relax - c_skipped;
        counters[C2_AUG_SKIPPED] = c_skipped;
        *reinterpret_cast<int *>(a.misc + 4) = err;
    }
}
#undef SLOT_COL
#define SLOT_COL(sl) ((((sl) / 4) * BLOCK2 + tid) * 4 + ((sl) % 4))
#undef SLOT_COL

// ------------------------------------------------------------------------------------------
constexpr int FAST_NMAX = 1 << 18;   // float32 cached-chain path (index arithmetic is 32-bit in bytes per row)
__device__ __forceinline__ uint64_t ld_u64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u64(uint64_t *p, uint64_t x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------------------------------
// Cache-certified augmentation ("lazy" Dijkstra): the row caches also serve the augmentation.
//
// Prices never increase (RT/ARR lower them; the augmentation's update v += d - curmin has d < curmin;
// the one-ulp rounding exception is handled below), so a row's cache floor F_i stays a lower bound of the
// reduced cost of every column outside its cache.  Let T be the smallest distance of an unassigned column
// seen so far in this search: the search ends at a distance <= T, and a column whose distance is > T when
// the search ends is never scanned, so its d and pred are never observed.  A scan of row i with offset h
// changes a non-cached column j to fl(fl(c-v_j) - h) >= fl(F_i - h); if that bound is > T the scan is
// observably identical to relaxing the <= 63 cached columns only.  Strictness matters: an update equal to
// T could decide a tie between two unassigned columns.  Otherwise the whole workgroup scans the row.
//
// State: one 64-bit word per column in L2-resident global memory, (ordered d << 32) | step of the scan
// that set it (unsigned min == "strictly smaller d wins, the earlier scan on equal d", i.e. the oracle's
// `v2 < d` rule; pred is recovered as the row of that step); never written again once the column is scanned.  The pick
// structure is in LDS: for each block of 64 columns the smallest (d, assigned?, column) key.  A cached
// step runs on wave 0 alone: LDS min over the block keys, three independent loads (cache row, the picked
// column's block for its new minimum, c[i][jp]), <= 63 fire-and-forget 64-bit atomic mins + LDS mins.
// Columns whose price ROSE by rounding in a price update (possible only by an ulp when v+d crosses a
// binade) are kept in an exception list and relaxed explicitly in every cached step.
// ------------------------------------------------------------------------------------------
//   srow: [n+1]; dkey: [n] 64-bit distance words; rowgid / rowmap as in Chain2Args
//   may_bail   a dense kernel can take over: give up when the cache certificates keep failing
//   debug_exc  tests: pretend the first debug_exc columns had a price rounded upwards (exception list)
#define LAZY_FIELDS(P, S)                                                                                                \
    S(int, n) S(int64_t, ld) P(const float, cost) P(float, gv) P(float, gu) P(float, sumvd) P(float, cassign) P(uint64_t, dkey) \
    P(int32_t, rowsol) P(int32_t, colsol) P(int32_t, freerows) P(int32_t, srow) P(int32_t, slist) P(int32_t, slevel)          \
    P(const int32_t, rowgid) P(const int32_t, rowmap) P(const uint32_t, cache_col) P(const float, cache_val)                  \
    P(float, g_hbest) P(int32_t, g_hstamp) P(char, misc) S(int, ngroups) S(int, gmode) S(int, may_bail) S(int, debug_exc)
struct LazyArgs { LAZY_FIELDS(F_PTR, F_VAL) };
struct LazyArgsG { LAZY_FIELDS(F_GPTR, F_VAL) };
static_assert(sizeof(LazyArgs) == sizeof(LazyArgsG), "mirror layout");
__device__ __forceinline__ LazyArgs load_args(const LazyArgs *__restrict__ batch) {
    const LazyArgsG g = reinterpret_cast<const LazyArgsG *>(batch)[blockIdx.x];
    LazyArgs a;
    LAZY_FIELDS(F_COPY_PTR, F_COPY_VAL)
    return a;
}
struct LazyCmd { int op, row, step, stamp; float h; };
enum { LZ_DENSE = 1, LZ_EXIT = 2, LZ_ERR = 3, LZ_INIT_DENSE = 4 };
enum { C2_AUG_DENSE = C2_NCOUNTERS, C2_AUG_SPARSE_INIT, C3_NCOUNTERS };
constexpr int LZ_MAXEXC = 64;

__device__ __forceinline__ void lds_min_u64(uint64_t *p, uint64_t x) {
    (void)__hip_atomic_fetch_min(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void gl_min_u64(uint64_t *p, uint64_t x) {
    (void)__hip_atomic_fetch_min(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// -DLZ_PROF (tools/prof_lazy_step.py): s_memtime stamps of wave 0 inside a cache-certified augmentation step
#ifdef LZ_PROF
__device__ long long g_lz_prof[24];
#define LZ_STAMP(k) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)"); \
                      if ((k) > 0) prof[k] += now_ - tlast; else if (tlast) prof[0] += 0; tlast = now_; profn[k]++; }
#define LZ_WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define LZ_STAMP2(k) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)"); \
                       prof2[k] += now_ - tlast2; tlast2 = now_; }
#else
#define LZ_STAMP(k)
#define LZ_STAMP2(k)
#define LZ_WAITVM
#endif
constexpr uint64_t LZ_INFKEY = 0xFFFFFFFF00000000ull;   // "no distance yet": any real d wins the unsigned min
// Before a column's word is used in this search its block of 64 words must hold this search's values:
// blocks are reset lazily (all 64 words = "no distance") the first time a search touches them.
__device__ __forceinline__ void lz_touch_blocks(bool act, int myblk, int myepoch, uint64_t *dkey, int32_t *s_ep, int32_t *s_tl,
                                                int &ntouch, int stamp, int npad, int lane) {
    uint64_t todo = __ballot(act && myepoch != stamp);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        const int b = (int)readlane32((uint32_t)myblk, l);
        if (b * 64 + lane < npad) st_u64(dkey + b * 64 + lane, LZ_INFKEY);
        if (lane == 0) { s_ep[b] = stamp; s_tl[ntouch] = b; }
        ntouch++;
        todo &= ~__ballot(myblk == b);
    }
}
// lexicographic minimum of 64-bit keys over a wave as two 32-bit all-reduces (value, then the low word
// among the lanes holding that value): much shorter dependency chains than a 64-bit DPP butterfly
__device__ __forceinline__ uint64_t wave_lexmin_u64(uint64_t k) {
    const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
    const uint32_t m = wave_min_u32(hi);
    const uint32_t l = wave_min_u32(hi == m ? lo : 0xFFFFFFFFu);
    return ((uint64_t)m << 32) | l;
}
// LDS_STATE: prices and colsol in LDS; CS_LDS (with !LDS_STATE, n <= 65535): colsol (u16) in LDS, prices in L2
template <bool LDS_STATE, bool CS_LDS = false>
__global__ __launch_bounds__(BLOCK2) void jv_aug_lazy(const LazyArgs *__restrict__ batch) {
    const LazyArgs a = load_args(batch);         // one workgroup per problem of the batch
    constexpr bool CSL = LDS_STATE || CS_LDS;
#ifdef LZ_PROF
    long long prof[6] = {0, 0, 0, 0, 0, 0}, profn[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    long long prof2[4] = {0, 0, 0, 0}, tlast2 = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    __shared__ Scratch2 s;
    __shared__ LazyCmd cmd;
    __shared__ int s_nexc;
    __shared__ uint32_t s_T;          // ordered key of T (wave 0 lowers it with LDS atomic mins)
    // the per-scan records of the current search (column, level, v + d, row) are staged here and written to the
    // global lists 64 at a time: every global store in a step holds back the next step's loads until it is acknowledged
    __shared__ int32_t s_ljp[64], s_llv[64], s_lrw[64];
    __shared__ float s_lsv[64];
    __shared__ int s_exc[LZ_MAXEXC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = a.n;
    const int64_t ld = a.ld;
    const float *__restrict__ cost = a.cost;
    const int nquad = (n + 3) >> 2, npad = nquad * 4, nb = (n + 63) >> 6;
    float *gv = a.gv;
    size_t off = 0;
    float *s_v = nullptr;
    uint16_t *s_cs = nullptr;
    if constexpr (LDS_STATE) {
        s_v = reinterpret_cast<float *>(dyn_lds);
        s_cs = reinterpret_cast<uint16_t *>(dyn_lds + (size_t)npad * 4);
        off = ((size_t)npad * 6 + 15) & ~(size_t)15;
    } else if constexpr (CS_LDS) {
        s_cs = reinterpret_cast<uint16_t *>(dyn_lds);
        off = ((size_t)npad * 2 + 15) & ~(size_t)15;
    }
    const int nbp = (nb + 511) & ~511;               // padded with KEYMAX: the pick reads 8 keys per lane, unpredicated
    uint64_t *bmin = reinterpret_cast<uint64_t *>(dyn_lds + off); off += (size_t)nbp * 8;
    uint32_t *s_sc = reinterpret_cast<uint32_t *>(dyn_lds + off); off += (size_t)nb * 8;   // 2 words per block
    uint32_t *s_un = reinterpret_cast<uint32_t *>(dyn_lds + off); off += (size_t)nb * 8;
    int32_t *s_ep = reinterpret_cast<int32_t *>(dyn_lds + off); off += (size_t)nb * 4;   // search stamp of each block's dkey words
    int32_t *s_tl = reinterpret_cast<int32_t *>(dyn_lds + off); off += (size_t)nb * 4;   // blocks touched by the current search
    off = (off + 7) & ~(size_t)7;
    const int gmode = a.gmode;
    float *hb = a.g_hbest;
    int32_t *hs = a.g_hstamp;
    if (gmode == 1) {
        hb = reinterpret_cast<float *>(dyn_lds + off);
        hs = reinterpret_cast<int32_t *>(dyn_lds + off + (size_t)a.ngroups * 4);
        for (int g = tid; g < a.ngroups; g += BLOCK2) hs[g] = 0;
    }
    int par = 0;
    if constexpr (CSL) {
        for (int c = tid; c < npad; c += BLOCK2) {
            if constexpr (LDS_STATE) s_v[c] = c < n ? gv[c] : 0.0f;
            const int32_t cs = c < n ? a.colsol[c] : -1;
            s_cs[c] = cs < 0 ? (uint16_t)0xFFFFu : (uint16_t)cs;
        }
    }
    for (int w = tid; w < 2 * nb; w += BLOCK2) {
        uint32_t m = 0;
        for (int b = 0; b < 32; b++) { const int c = w * 32 + b; if (c < n && a.colsol[c] < 0) m |= (1u << b); }
        s_un[w] = m;
    }
    if (tid == 0) s_nexc = a.debug_exc;
    if (tid < LZ_MAXEXC) s_exc[tid] = tid < n ? tid : 0;
    for (int b = tid; b < nb; b += BLOCK2) s_ep[b] = 0;
    long long c_sparse = 0;
    // between searches: every block minimum is "empty", no column is marked scanned
    for (int b = tid; b < nbp; b += BLOCK2) bmin[b] = KEYMAX;
    for (int w = tid; w < 2 * nb; w += BLOCK2) s_sc[w] = 0;
    __syncthreads();
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    long long c_relax = 0, c_hops = 0, c_augs = 0, c_skipped = 0, c_dense = 0;
    int err = 0;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(gv, 0, nquad * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdk = __builtin_amdgcn_make_buffer_rsrc(a.dkey, 0, nquad * 32, 0x00020000);

    // Wave 0 owns the control flow: it runs whole searches (sparse init, cached steps, price update, path flip)
    // on its own and calls the other waves in -- through `cmd` and the barrier below -- only for an operation
    // that needs a full cost row (dense init of a search, dense scan of a row).
    int f = 0, freerow = -1, stamp = 0;
    bool insearch = false, have = false, dense_used = false, isparse = false;
    float curmin = 0.0f, icv = 0.0f;
    uint32_t icc = COLSENT;
    int level = 0, nscan = 0, ntouch = 0;
    int nexc = a.debug_exc > LZ_MAXEXC ? LZ_MAXEXC : a.debug_exc;   // register copy of s_nexc (it only changes at a search's end)
    bool bail = false, no_cert = a.debug_exc > LZ_MAXEXC;
    long long w_relax0 = 0, w_dense0 = 0;           // start of the current hand-over window
    // software pipeline over searches: id of the free row after next, cache row of the next one
    int id_next = -1, id1_saved = -1;
    uint32_t ncc = COLSENT;
    float ncv = 0.0f;
    if (wave == 0 && numfree > 0) {
        const int id0 = __builtin_amdgcn_readfirstlane(ld_i32(a.freerows));
        ncc = ld_u32(a.cache_col + (int64_t)id0 * KC + lane);
        ncv = ld_f32(a.cache_val + (int64_t)id0 * KC + lane);
        id_next = numfree > 1 ? ld_i32(a.freerows + 1) : -1;
        freerow = id0;
    }
    for (;;) {
        if (wave == 0) {
            bool post = false;
            while (!post) {
                if (!insearch) {
                    if (f >= numfree || err || bail) { if (lane == 0) cmd.op = err ? LZ_ERR : LZ_EXIT; break; }
                    LZ_STAMP2(0)
                    // (freerow, ncc, ncv) were requested during the previous search; request the next ones now
                    const uint32_t cc = ncc;
                    const float cv = ncv;
                    const int id1 = __builtin_amdgcn_readfirstlane(id_next);
                    id1_saved = id1;
                    stamp = f + 1;
                    have = false; curmin = 0.0f; level = 0; nscan = 0; ntouch = 0; dense_used = false; insearch = true;
                    // ---- certified sparse init: if the free row's cache floor is above the distance of an
                    // unassigned cached column, no column outside the cache can matter in this search ----
                    const bool valid = lane < KCU && cc != COLSENT;
                    const int j = valid ? (int)cc : 0;
                    const float dd = cv - st_vget<LDS_STATE>(s_v, gv, j);
                    const bool un = valid && ((s_un[j >> 5] >> (j & 31)) & 1u);
                    const uint32_t odd = valid ? f2ord(dd) : 0xFFFFFFFFu;
                    const uint32_t t0 = wave_min_u32(un ? odd : 0xFFFFFFFFu);
                    // only now the requests for the next search (loads return in issue order: issued before the price gather
                    // above, these cache-row loads from HBM would have held it back)
                    if (id1 >= 0) {
                        ncc = ld_u32(a.cache_col + (int64_t)id1 * KC + lane);
                        ncv = ld_f32(a.cache_val + (int64_t)id1 * KC + lane);
                    }
                    id_next = f + 2 < numfree ? ld_i32(a.freerows + f + 2) : -1;
                    const float floor_f = __uint_as_float(readlane32(__float_as_uint(cv), KCU));
                    const bool cert0 = !no_cert && nexc == 0 && t0 != 0xFFFFFFFFu && floor_f > ord2f(t0);
                    if (cert0 && wave_min_u32(odd) == t0) {
                        // ---- single-edge search: the smallest distance of the whole row belongs to an unassigned
                        // column (the oracle's first pick ends the search at once: no scan, no price update);
                        // nothing of the pick structure is touched ----
                        const uint64_t me = __ballot(un && odd == t0);
                        const int le = __builtin_ctzll(me);   // (cache rows are sorted by column: lowest lane = lowest column)
                        const int ep = (int)readlane32(cc, le);
                        const float cie = __uint_as_float(readlane32(__float_as_uint(cv), le));
                        if (lane == 0) {
                            st_csset<CSL>(s_cs, a.colsol, ep, freerow);
                            st_f32(a.cassign + ep, cie);
                            st_i32(a.rowsol + freerow, ep);
                            s_un[ep >> 5] &= ~(1u << (ep & 31));
                        }
                        c_hops++; c_augs++; c_sparse++;
                        f++;
                        freerow = id1_saved;
                        insearch = false;
                        LZ_STAMP2(1)
                        continue;
                    }
                    if (cert0) {
                        const bool act = valid && !(dd > ord2f(t0));
                        lz_touch_blocks(act, j >> 6, s_ep[j >> 6], a.dkey, s_ep, s_tl, ntouch, stamp, npad, lane);
                        if (act) {
                            const uint32_t od = f2ord(dd);
                            st_u64(a.dkey + j, (uint64_t)od << 32);                 // step 0 = the free row
                            lds_min_u64(bmin + (j >> 6), ((uint64_t)od << 32) | (un ? 0u : 0x80000000u) | (uint32_t)j);
                        }
                        if (lane == 0) { s_T = t0; st_i32(a.srow, freerow); }
                        c_sparse++;
                        isparse = true; icc = cc; icv = cv;
                        LZ_STAMP2(1)
                    } else {
                        if (lane == 0) { cmd.op = LZ_INIT_DENSE; cmd.row = freerow; cmd.stamp = stamp; }
                        isparse = false; dense_used = true; post = true;
                        continue;
                    }
                }
                for (;;) {
                    LZ_STAMP(0)
                    // ---- pick: smallest (d, assigned?, column) over the block minima ----
                    uint64_t k = KEYMAX;
                    for (int b0 = 0; b0 < nbp; b0 += 512) {
                        uint64_t kk[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) kk[u] = bmin[b0 + u * 64 + lane];
#pragma unroll
                        for (int u = 0; u < 8; u++) k = umin64(k, kk[u]);
                    }
                    {   // value first; the low word only needs a second reduction when several lanes tie on d
                        const uint32_t hi = (uint32_t)(k >> 32);
                        const uint32_t m = wave_min_u32(hi);
                        const uint64_t eq = __ballot(hi == m);
                        uint32_t lo;
                        if (__builtin_popcountll(eq) == 1) lo = readlane32((uint32_t)k, __builtin_ctzll(eq));
                        else lo = wave_min_u32(hi == m ? (uint32_t)k : 0xFFFFFFFFu);
                        k = ((uint64_t)m << 32) | lo;
                    }
                    const float dmin = key_val(k);
                    if (k == KEYMAX || !(dmin < INFINITY)) { err = CYTO_ERR_INTERNAL; insearch = false; break; }
                    const int jp = (int)((uint32_t)k & 0x7FFFFFFFu);
                    LZ_STAMP(1)
                    if (!have || dmin != curmin) { level++; curmin = dmin; have = true; }
                    if (!((uint32_t)k & 0x80000000u)) {
                        // ======== end of the search (wave 0 alone): price update, path flip, clean-up ========
                        LZ_STAMP2(2)
                        const int endofpath = jp;
                        {
                            const int rem = nscan & 63, base = nscan - rem;
                            if (lane < rem) {
                                st_i32(a.slist + base + lane, s_ljp[lane]); st_i32(a.slevel + base + lane, s_llv[lane]);
                                st_f32(a.sumvd + base + lane, s_lsv[lane]); st_i32(a.srow + 1 + base + lane, s_lrw[lane]);
                            }
                        }
                        for (int k2 = lane; k2 < nscan; k2 += 64) {
                            if (ld_i32(a.slevel + k2) < level) {
                                const int j = ld_i32(a.slist + k2);
                                const float vold = st_vget<LDS_STATE>(s_v, gv, j);
                                const float vnew = ld_f32(a.sumvd + k2) - curmin;
                                st_vset<LDS_STATE>(s_v, gv, j, vnew);
                                if (vnew > vold) {   // rounding pushed a price UP: column j leaves the cache certificates
                                    const int e = atomicAdd(&s_nexc, 1);
                                    if (e < LZ_MAXEXC) s_exc[e] = j;
                                }
                            }
                        }
                        nexc = s_nexc;
                        // more such columns than the list holds: the certificates are abandoned for the rest of the solve
                        // (every scan reads its full row; still exact, just slow) -- or the dense kernel takes over
                        if (nexc > LZ_MAXEXC) { nexc = LZ_MAXEXC; no_cert = true; if (a.may_bail) bail = true; }
                        // c[freerow][endofpath] is in the sparse init's cache row when the path is a single edge
                        float cie0 = 0.0f;
                        bool have_cie0 = false;
                        if (nscan == 0 && isparse) {
                            const uint64_t mm = __ballot(lane < KCU && icc == (uint32_t)endofpath);
                            if (mm) { cie0 = __uint_as_float(readlane32(__float_as_uint(icv), __builtin_ctzll(mm))); have_cie0 = true; }
                        }
                        int hops = 0;
                        if (lane == 0) {
                            int ep = endofpath;
                            const int st0 = nscan == 0 ? 0 : (int32_t)(uint32_t)ld_u64(a.dkey + ep);
                            int i = st0 == 0 ? freerow : ld_i32(a.srow + st0);
                            for (;;) {
                                st_csset<CSL>(s_cs, a.colsol, ep, i);
                                st_f32(a.cassign + ep, (have_cie0 && hops == 0) ? cie0 : cost[row_off(a.rowmap, i, ld) + ep]);
                                const int j1 = ep;
                                if (i != freerow) ep = ld_i32(a.rowsol + i);      // (the free row owns no column)
                                st_i32(a.rowsol + i, j1);
                                hops++;
                                if (i == freerow) break;
                                const int stp = (int32_t)(uint32_t)ld_u64(a.dkey + ep);
                                i = stp == 0 ? freerow : ld_i32(a.srow + stp);
                            }
                            s_un[endofpath >> 5] &= ~(1u << (endofpath & 31));
                        }
                        c_hops += __builtin_amdgcn_readfirstlane(hops);
                        // back to the between-searches state of the LDS pick structure
                        if (!dense_used) {
                            for (int k2 = lane; k2 < ntouch; k2 += 64) { const int b = s_tl[k2]; bmin[b] = KEYMAX; s_sc[2 * b] = 0; s_sc[2 * b + 1] = 0; }
                        } else {
                            for (int b = lane; b < nb; b += 64) { bmin[b] = KEYMAX; s_sc[2 * b] = 0; s_sc[2 * b + 1] = 0; }
                        }
                        LZ_STAMP2(3)
                        c_augs++;
                        f++;
                        freerow = id1_saved;
                        insearch = false;
                        // instances whose searches run deeper than the caches reach (>= 10 % full-row scans) are
                        // faster on the register-resident dense kernel: hand the remaining free rows over
                        // (judged on windows of >= 8192 scans, so that a dense-heavy start alone does not decide)
                        if (a.may_bail && c_relax - w_relax0 >= 8192) {
                            if ((c_dense - w_dense0) * 10 >= c_relax - w_relax0) bail = true;
                            w_relax0 = c_relax; w_dense0 = c_dense;
                        }
                        break;
                    }
                    const int i = __builtin_amdgcn_readfirstlane(st_csget<CSL>(s_cs, a.colsol, jp));
                    const float vjp = st_vget<LDS_STATE>(s_v, gv, jp);
                    const int step = nscan + 1;
                    const int blk = jp >> 6;
                    const int jb = blk * 64 + lane;
                    // independent loads: c[i][jp], the row's cache, the picked column's block, the row's group
                    // (the cache row last: a skipped scan continues without waiting for it)
                    const float cip_raw = ld_f32(a.cassign + jp);
                    const uint64_t dk = jb < n ? ld_u64(a.dkey + jb) : 0ull;
                    const int g_raw = gmode ? a.rowgid[i] : 0;
                    const uint32_t cc = ld_u32(a.cache_col + (int64_t)i * KC + lane);
                    const float cv = ld_f32(a.cache_val + (int64_t)i * KC + lane);
                    // retire column jp
                    if (lane == 0) {
                        s_ljp[nscan & 63] = jp; s_llv[nscan & 63] = level; s_lsv[nscan & 63] = vjp + dmin; s_lrw[nscan & 63] = i;
                        atomicOr(&s_sc[jp >> 5], 1u << (jp & 31));
                    }
                    const uint32_t scw = s_sc[blk * 2 + (lane >> 5)], unw = s_un[blk * 2 + (lane >> 5)];
                    LZ_WAITVM
                    LZ_STAMP(2)
                    const float cip = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(cip_raw)));
                    const float h = (cip - vjp) - curmin;
                    // (the word of a scanned column is never written again -- every writer tests the scanned bit -- so it
                    // keeps the step that set its distance: that is its predecessor for the path flip)
                    bool skip = false;
                    if (gmode) {
                        const int g = __builtin_amdgcn_readfirstlane(g_raw);
                        float hbv; int hsv;
                        if (gmode == 1) { hbv = hb[g]; hsv = hs[g]; } else { hbv = ld_f32(hb + g); hsv = ld_i32(hs + g); }
                        skip = (hsv == stamp) && (h <= hbv);
                        if (!skip && lane == 0) {
                            if (gmode == 1) { hb[g] = h; hs[g] = stamp; } else { st_f32(hb + g, h); st_i32(hs + g, stamp); }
                        }
                    }
                    // new minimum of the picked column's block (this step's relaxations are applied after it)
                    {
                        const bool live = jb < n && !((scw >> (lane & 31)) & 1u);
                        const bool unb = (unw >> (lane & 31)) & 1u;
                        const uint32_t hi = live ? (uint32_t)(dk >> 32) : 0xFFFFFFFFu;
                        const uint32_t m = wave_min_u32(hi);
                        const uint64_t eq = __ballot(live && hi == m), equ = __ballot(live && hi == m && unb);
                        uint64_t kb = KEYMAX;
                        if (eq) {   // lane order is column order inside a block: lowest unassigned lane, else lowest lane
                            const int wl = equ ? __builtin_ctzll(equ) : __builtin_ctzll(eq);
                            kb = ((uint64_t)m << 32) | (equ ? 0u : 0x80000000u) | (uint32_t)(blk * 64 + wl);
                        }
                        if (lane == 0) bmin[blk] = kb;
                    }
                    nscan++;
                    c_relax++;
                    if ((nscan & 63) == 0) {
                        const int base = nscan - 64;
                        st_i32(a.slist + base + lane, s_ljp[lane]); st_i32(a.slevel + base + lane, s_llv[lane]);
                        st_f32(a.sumvd + base + lane, s_lsv[lane]); st_i32(a.srow + 1 + base + lane, s_lrw[lane]);
                    }
                    LZ_STAMP(3)
                    if (skip) { c_skipped++; continue; }
                    const float floor_i = __uint_as_float(readlane32(__float_as_uint(cv), KCU));
                    const float T = ord2f(s_T);
                    if (no_cert || !((floor_i - h) > T)) {
                        if (lane == 0) { cmd.op = LZ_DENSE; cmd.row = i; cmd.step = step; cmd.h = h; cmd.stamp = stamp; }
                        dense_used = true; post = true;
                        break;
                    }
                    // ---- cached relaxation: lane = cache entry; plus the (normally empty) exception list ----
                    {
                        const bool valid = lane < KCU && cc != COLSENT;
                        const int j = valid ? (int)cc : 0;
                        // one round of LDS gathers: price, scanned / unassigned words, the block's search stamp
                        const float vj = st_vget<LDS_STATE>(s_v, gv, j);
                        const uint32_t scw = s_sc[j >> 5], unw = s_un[j >> 5];
                        const int32_t epj = s_ep[j >> 6];
                        const float v2 = (cv - vj) - h;
                        const bool scn = (scw >> (j & 31)) & 1u;
                        const bool act = valid && !scn && !(v2 > T);
                        lz_touch_blocks(act, j >> 6, epj, a.dkey, s_ep, s_tl, ntouch, stamp, npad, lane);
                        if (act) {
                            const bool un = (unw >> (j & 31)) & 1u;
                            const uint32_t o2 = f2ord(v2);
                            gl_min_u64(a.dkey + j, ((uint64_t)o2 << 32) | (uint32_t)step);
                            lds_min_u64(bmin + (j >> 6), ((uint64_t)o2 << 32) | (un ? 0u : 0x80000000u) | (uint32_t)j);
                            if (un) atomicMin(&s_T, o2);
                        }
#ifdef LZ_PROF
                        prof2[1] += __builtin_popcountll(__ballot(act));
#endif
                    }
                    LZ_STAMP(4)
                    if (nexc > 0) {
                        const bool ev = lane < nexc;
                        const int j = ev ? s_exc[lane] : 0;
                        const float vj = st_vget<LDS_STATE>(s_v, gv, j);
                        const float v2 = ev ? (cost[row_off(a.rowmap, i, ld) + j] - vj) - h : 0.0f;
                        const bool scn = (s_sc[j >> 5] >> (j & 31)) & 1u;
                        lz_touch_blocks(ev && !scn, j >> 6, s_ep[j >> 6], a.dkey, s_ep, s_tl, ntouch, stamp, npad, lane);
                        if (ev && !scn) {
                            const bool un = (s_un[j >> 5] >> (j & 31)) & 1u;
                            const uint32_t o2 = f2ord(v2);
                            gl_min_u64(a.dkey + j, ((uint64_t)o2 << 32) | (uint32_t)step);
                            lds_min_u64(bmin + (j >> 6), ((uint64_t)o2 << 32) | (un ? 0u : 0x80000000u) | (uint32_t)j);
                            if (un) atomicMin(&s_T, o2);
                        }
                    }
                    LZ_STAMP(5)
                }
            }
        }
        __syncthreads();                               // command posted; wave 0's global traffic drained
        {
            const int op = cmd.op;
            if (op == LZ_ERR) { err = CYTO_ERR_INTERNAL; break; }
            if (op == LZ_EXIT) break;
            const int stamp = cmd.stamp;               // (shadows wave 0's copy with the same value)
            if (op == LZ_INIT_DENSE) {
                const int freerow = cmd.row;
                // ================= dense init: d = c[freerow] - v for every column (whole workgroup) =================
                        float tl = INFINITY;
                {
                    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(cost + row_off(a.rowmap, freerow, ld)), 0, (int)(ld * 4), 0x00020000);
                    for (int q0 = 0; q0 < nquad; q0 += BLOCK2) {
                        const int q = q0 + tid;
                        uint64_t bk = KEYMAX;
                        if (q < nquad) {
                            const u32x4_t xr = __builtin_amdgcn_raw_buffer_load_b128(rr, q * 16, 0, 0);
                            float4 vv;
                            if constexpr (LDS_STATE) vv = *reinterpret_cast<const float4 *>(s_v + q * 4);
                            else {
                                const u32x4_t vr = __builtin_amdgcn_raw_buffer_load_b128(rv, q * 16, 0, 0x10);
                                vv = make_float4(__uint_as_float(vr.x), __uint_as_float(vr.y), __uint_as_float(vr.z), __uint_as_float(vr.w));
                            }
                            const uint32_t um = (s_un[q >> 3] >> ((q & 7) * 4)) & 0xFu;
                            const float xs[4] = {__uint_as_float(xr.x), __uint_as_float(xr.y), __uint_as_float(xr.z), __uint_as_float(xr.w)};
                            const float vs[4] = {vv.x, vv.y, vv.z, vv.w};
                            uint64_t dq[4] = {0, 0, 0, 0};
        #pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int c = q * 4 + e;
                                if (c < n) {
                                    const float dd = xs[e] - vs[e];
                                    const uint32_t od = f2ord(dd);
                                    const bool un = (um >> e) & 1u;
                                    dq[e] = (uint64_t)od << 32;                       // step 0 = the free row
                                    bk = umin64(bk, ((uint64_t)od << 32) | (un ? 0u : 0x80000000u) | (uint32_t)c);
                                    if (un) tl = fminf(tl, dd);
                                }
                            }
                            // dkey has npad entries: whole quads are stored (the pad words are never read)
                            const u32x4_t w0 = {(uint32_t)dq[0], (uint32_t)(dq[0] >> 32), (uint32_t)dq[1], (uint32_t)(dq[1] >> 32)};
                            const u32x4_t w1 = {(uint32_t)dq[2], (uint32_t)(dq[2] >> 32), (uint32_t)dq[3], (uint32_t)(dq[3] >> 32)};
                            __builtin_amdgcn_raw_buffer_store_b128(w0, rdk, q * 32, 0, 0x10);
                            __builtin_amdgcn_raw_buffer_store_b128(w1, rdk, q * 32 + 16, 0, 0x10);
                        }
                        bk = min64_row_allreduce(bk);          // 16 lanes = 16 quads = one block of 64 columns
                        if ((lane & 15) == 0 && q < nquad) { bmin[q >> 4] = bk; s_ep[q >> 4] = stamp; }
                    }
                }
                if (tid == 0) st_i32(a.srow, freerow);
                {
                    const uint32_t t0 = wg_min_u32(f2ord(tl), s, par);
                    if (tid == 0) s_T = t0;
                }
                __syncthreads();
                continue;
            }
            // ================= dense scan of row cmd.row (certificate failed): whole workgroup =================
            // Only a relaxation to a value <= T (the threshold when the scan starts; T only decreases) can be observed,
            // so the per-column words are touched for those columns only; a 64-column block none of whose columns
            // matters is skipped altogether, one that is met for the first time in this search is initialised here.
            {
                const int i = cmd.row, step = cmd.step;
                const float h = cmd.h;
                const float T0 = ord2f(s_T);
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(cost + row_off(a.rowmap, i, ld)), 0, (int)(ld * 4), 0x00020000);
                float tl2 = INFINITY;
                for (int q0 = 0; q0 < nquad; q0 += BLOCK2) {
                    const int q = q0 + tid;
                    uint32_t o2s[4] = {0, 0, 0, 0};
                    uint32_t okm = 0, um = 0;
                    if (q < nquad) {
                        const u32x4_t xr = __builtin_amdgcn_raw_buffer_load_b128(rr, q * 16, 0, 0);
                        float4 vv;
                        if constexpr (LDS_STATE) vv = *reinterpret_cast<const float4 *>(s_v + q * 4);
                        else {
                            const u32x4_t vr = __builtin_amdgcn_raw_buffer_load_b128(rv, q * 16, 0, 0x10);
                            vv = make_float4(__uint_as_float(vr.x), __uint_as_float(vr.y), __uint_as_float(vr.z), __uint_as_float(vr.w));
                        }
                        um = (s_un[q >> 3] >> ((q & 7) * 4)) & 0xFu;
                        const uint32_t sm = (s_sc[q >> 3] >> ((q & 7) * 4)) & 0xFu;
                        const float xs[4] = {__uint_as_float(xr.x), __uint_as_float(xr.y), __uint_as_float(xr.z), __uint_as_float(xr.w)};
                        const float vs[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int c = q * 4 + e;
                            const float v2 = (xs[e] - vs[e]) - h;
                            o2s[e] = f2ord(v2);
                            if (c < n && !((sm >> e) & 1u) && !(v2 > T0)) {
                                okm |= 1u << e;
                                if ((um >> e) & 1u) tl2 = fminf(tl2, v2);
                            }
                        }
                    }
                    // does any column of my block (16 lanes = 64 columns) matter?
                    const uint64_t hitm = __ballot(okm != 0);
                    if ((hitm >> (lane & 48)) & 0xFFFFull) {
                        const int b = q >> 4;
                        const bool fresh = q < nquad && s_ep[b] == stamp;
                        const bool first = q < nquad && !fresh;
                        uint64_t bk = KEYMAX;
                        if (fresh) {
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                if ((okm >> e) & 1u) {
                                    const int c = q * 4 + e;
                                    const uint64_t dkc = ld_u64(a.dkey + c);
                                    if (o2s[e] < (uint32_t)(dkc >> 32)) {
                                        st_u64(a.dkey + c, ((uint64_t)o2s[e] << 32) | (uint32_t)step);
                                        bk = umin64(bk, ((uint64_t)o2s[e] << 32) | (((um >> e) & 1u) ? 0u : 0x80000000u) | (uint32_t)c);
                                    }
                                }
                            }
                        } else if (first) {
                            uint64_t dq[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int c = q * 4 + e;
                                dq[e] = LZ_INFKEY;
                                if ((okm >> e) & 1u) {
                                    dq[e] = ((uint64_t)o2s[e] << 32) | (uint32_t)step;
                                    bk = umin64(bk, ((uint64_t)o2s[e] << 32) | (((um >> e) & 1u) ? 0u : 0x80000000u) | (uint32_t)c);
                                }
                            }
                            const u32x4_t w0 = {(uint32_t)dq[0], (uint32_t)(dq[0] >> 32), (uint32_t)dq[1], (uint32_t)(dq[1] >> 32)};
                            const u32x4_t w1 = {(uint32_t)dq[2], (uint32_t)(dq[2] >> 32), (uint32_t)dq[3], (uint32_t)(dq[3] >> 32)};
                            __builtin_amdgcn_raw_buffer_store_b128(w0, rdk, q * 32, 0, 0x10);
                            __builtin_amdgcn_raw_buffer_store_b128(w1, rdk, q * 32 + 16, 0, 0x10);
                        }
                        bk = min64_row_allreduce(bk);
                        if ((lane & 15) == 0 && q < nquad) {
                            if (bk != KEYMAX) lds_min_u64(bmin + b, bk);
                            s_ep[b] = stamp;
                        }
                    }
                }
                {
                    const uint32_t t2 = wg_min_u32(f2ord(tl2), s, par);
                    if (tid == 0) atomicMin(&s_T, t2);
                }
                c_dense++;
                __syncthreads();
            }
        }
    }
    // ---- write back prices and colsol, then duals u and the total ----
    if constexpr (CSL) {
        for (int c = tid; c < n; c += BLOCK2) {
            if constexpr (LDS_STATE) gv[c] = s_v[c];
            const uint16_t cs = s_cs[c];
            a.colsol[c] = cs == 0xFFFFu ? -1 : (int32_t)cs;
        }
    }
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < n; i += BLOCK2) {
        const int j = ld_i32(a.rowsol + i);
        if (j < 0) continue;              // still free: the kernel gave up and the dense kernel finishes (and redoes this)
        const float cij = cost[row_off(a.rowmap, i, ld) + j];
        const float vj = ld_f32(gv + j);
        a.gu[i] = cij - vj;
        part += (double)cij;
    }
#pragma unroll
    for (int off2 = 32; off2 >= 1; off2 >>= 1) part += __shfl_xor(part, off2);
    if (lane == 0) s.sum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NW2; w++) t += s.sum[w];
        *reinterpret_cast<double *>(a.misc + 8) = t;
        long long *counters = reinterpret_cast<long long *>(a.misc + 16);
        counters[C_AUG_INIT] = c_augs; counters[C_AUG_RELAX] = c_relax; counters[C_AUGS] = c_augs; counters[C_HOPS] = c_hops;
        counters[C_ROWS_READ] = counters[C2_DENSE_REFRESH] + (c_augs - c_sparse) + c_dense;
        counters[C2_AUG_SKIPPED] = c_skipped;
        counters[C2_AUG_DENSE] = c_dense;
        counters[C2_AUG_SPARSE_INIT] = c_sparse;
        *reinterpret_cast<int *>(a.misc + 136) = f;          // searches completed (== numfree unless the kernel gave up)
        *reinterpret_cast<int *>(a.misc + 4) = err;
#ifdef LZ_PROF
        for (int k = 0; k < 6; k++) { g_lz_prof[k] = prof[k]; g_lz_prof[6 + k] = profn[k]; }
        for (int k = 0; k < 4; k++) g_lz_prof[12 + k] = prof2[k];
        g_lz_prof[16] = c_augs; g_lz_prof[17] = c_relax; g_lz_prof[18] = c_dense; g_lz_prof[19] = c_sparse;
#endif
    }
}

// ------------------------------------------------------------------------------------------
// The float64 certificate of a float32 solve (cyto_lap_opts.certify).  The solvers compare fl32(c - v): an assigned row sits on a
// minimum of its reduced costs AS ROUNDED, so in exact arithmetic its column can lose to another by a fraction of an ulp -- the
// duals (u_i := c[i][rowsol_i] - v[rowsol_i], v) are feasible to ~1e-9, not exactly.  One more pass over the matrix turns that into a
// PROOF of how far from optimal the assignment can be: with every difference evaluated in float64 (exact for float32 operands
// whose exponents lie within 29 binades of each other),
//     gap = sum_i ( u_i - min_j (c[i][j] - v[j]) )  >=  total - optimum  >=  0
// (lower every u_i to its row's true minimum: the duals become feasible, their objective is total - gap).  gap == 0: the
// assignment is optimal for the float32 matrix in exact arithmetic; an instance whose optimum is unique by more than `gap` has
// exactly these indices whatever the solver's constants.  A wave per row, one sweep; the rows' terms are summed in row order
// (one workgroup, fixed tree): the same bits on every run.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dual_gap_rows(int n, int64_t ld, const float *__restrict__ cost, const int32_t *__restrict__ rowmap,
                                                     const int32_t *__restrict__ rowsol, const float *__restrict__ v,
                                                     double *__restrict__ viol) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nq = (n + 3) >> 2;
    const float4 *__restrict__ v4 = reinterpret_cast<const float4 *>(v);
    for (int i = gw; i < n; i += nw) {
        const float *__restrict__ row = cost + (int64_t)(rowmap ? rowmap[i] : i) * ld;
        const float4 *__restrict__ r4 = reinterpret_cast<const float4 *>(row);
        double m = INFINITY;
        for (int q0 = lane; q0 < nq; q0 += 64 * 8) {
            float4 x[8], p[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int q = q0 + 64 * k;
                const bool in = q < nq && q * 4 + 3 < n;                 // (whole quads; the row's last, partial quad below)
                x[k] = in ? r4[q] : make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
                p[k] = in ? v4[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                m = fmin(m, fmin(fmin((double)x[k].x - (double)p[k].x, (double)x[k].y - (double)p[k].y),
                                 fmin((double)x[k].z - (double)p[k].z, (double)x[k].w - (double)p[k].w)));
            }
        }
        if (lane < (n & 3)) { const int c = (n & ~3) + lane; m = fmin(m, (double)row[c] - (double)v[c]); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_xor(m, off));
        if (lane == 0) {
            const int j = rowsol[i];
            const double ui = (double)row[j] - (double)v[j];
            viol[i] = ui > m ? ui - m : 0.0;
        }
    }
}
// out[0] = sum (rows in ascending order within a thread, then a fixed tree), out[1] = max, out[2] = rows with a positive term
__global__ __launch_bounds__(1024) void dual_gap_finish(int n, const double *__restrict__ viol, double *__restrict__ out) {
    __shared__ double s_sum[1024], s_max[1024];
    __shared__ int s_cnt[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024, lo = t * per, hi = min(n, lo + per);
    double s = 0.0, mx = 0.0; int c = 0;
    for (int i = lo; i < hi; i++) { const double x = viol[i]; s += x; mx = fmax(mx, x); c += x > 0.0; }
    s_sum[t] = s; s_max[t] = mx; s_cnt[t] = c;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (t < w) { s_sum[t] += s_sum[t + w]; s_max[t] = fmax(s_max[t], s_max[t + w]); s_cnt[t] += s_cnt[t + w]; }
        __syncthreads();
    }
    if (t == 0) { out[0] = s_sum[0]; out[1] = s_max[0]; out[2] = (double)s_cnt[0]; }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static std::atomic<int> g_par_busy[64];     // per device: a several-searches-at-once kernel (wide_aug<..,PAR>) is in flight

static const cyto_lap_opts k_default_opts = {};

static int check_opts(const cyto_lap_opts &o) {
    if (o.chain_variant < 0 || o.chain_variant > 3 || o.augmentation < 0 || o.augmentation > 2 || o.inject_exceptions < 0 ||
        o.no_handover < 0 || o.no_handover > 1)
        return CYTO_ERR_BAD_ARG;
    if (o.group_state_global < 0 || o.group_state_global > 1 || o.aux_state_global < 0 || o.aux_state_global > 1) return CYTO_ERR_BAD_ARG;
    if (o.mode < 0 || o.mode > 2 || o.wide_rounds < -1 || o.wide_groups < -1 || o.wide_groups > 32 || o.wide_rebuild < -1 ||
        o.wide_wipe < 0 || o.wide_wipe > 2048 || o.wide_par < -1 || o.wide_par > WIDE_PAR_GMAX) return CYTO_ERR_BAD_ARG;
    if (o.cache_waves < -1 || o.cache_waves > 32 || (o.cache_unroll != 0 && o.cache_unroll != 4 && o.cache_unroll != 8) || o.cache_stream < -1 ||
        o.cache_stream > 1)
        return CYTO_ERR_BAD_ARG;
    if (o.certify < 0 || o.certify > 1 || o.polish < 0 || o.polish > 1) return CYTO_ERR_BAD_ARG;
    for (int r : o.reserved) if (r != 0) return CYTO_ERR_BAD_ARG;            // (must be zero: room for later knobs without another ABI break)
    return CYTO_OK;
}

// ---- float32: one problem of a batch (all problems of a batch have the same n) ----
struct F32Job {
    // in
    const float *cost = nullptr; int64_t ld = 0; int cost_on_device = 0;
    const int32_t *rowmap_host = nullptr; int nu = 0;   // optional row map: LAP row i reads stored row rowmap[i] (nu stored rows)
    // out (host pointers, may be null)
    int32_t *rowsol = nullptr, *colsol = nullptr; float *u = nullptr, *v = nullptr; double *total = nullptr; cyto_lap_info *info = nullptr;
    int status = CYTO_OK;
    // device state
    DevBuf slab;                // ONE block per problem for the ten work buffers every solve needs (views below): a batch of 256 problems made
                                // 3 300 hipMallocs on a process's first call -- 160 ms of a 0.67-s call (round 6, bench c4_chunks.first_call_wall_s)
    DevBuf staged, b_fws, b_iws, b_imin, b_pmin, b_parg, b_misc, b_same, b_gid, b_ccol, b_cval, b_ghb, b_ghs, b_lzhb, b_lzhs;
    DevBuf b_rowmap, b_ulist, b_ufirst, b_wide;
    int nused = 0;
    const float *dcost = nullptr; int64_t dld = 0;
    int h_nonfinite = 0, h_ngroups = 0, h_hand[3] = {0, 0, 0};
    Chain2Args c2; LazyArgs la;
    size_t shm_lazy = 0, shm_aug = 0;
    bool alive() const { return status == CYTO_OK; }
};

struct F32Plan {            // what depends on n (and the options) only: identical for every problem of the batch
    int n, colblocks, rowblocks, rows_per_block, cache_grid;
    int cache_stream = 1;           // build_row_caches_wave: the guess-free single sweep (cb_stream); 0: a neighbour's floor as the guess
    int cache_waves = 8, cache_unroll = 4, cus = 256;      // build_row_caches_wave: waves per CU, quads in flight per lane; CUs of the device (device_cus)
    int cache_keep = 0;             // a REBUILD leaves a row alone that still has this many cached columns below its floor (0: every row)
    bool force_l2, lds_variant, lazy, lz_lds_state, lz_cs_lds, no_cs_lds, wide;
    long long wide_rounds;
    int wide_groups, wide_rebuild, wide_wipe, wide_par;
    size_t shm_chain, lz_base_shm;
};

template <int CH, bool LDS_STATE>
static int launch_batch(const F32Plan &pl, std::vector<F32Job> &jobs, hipStream_t stream, hipEvent_t ev_cache_done,
                        hipEvent_t ev_arr_done, DevBuf &d_c2, DevBuf &d_la) {
    const int n = pl.n;
    std::vector<int> live;
    for (int b = 0; b < (int)jobs.size(); b++) if (jobs[b].alive()) live.push_back(b);
    const int nl = (int)live.size();
    if (!nl) return CYTO_OK;
    int rc;
    const bool cs_lds = !LDS_STATE && n <= 65535 && !pl.no_cs_lds;
    void (*kern)(const Chain2Args *) = jv_chain2<CH, LDS_STATE, false>;
    if constexpr (!LDS_STATE) { if (cs_lds) kern = jv_chain2<CH, false, true>; }
    bool caches_built = false;                                   // (the first build finds uninitialised memory: nothing to keep)
    auto build_caches = [&](const int *remaining) -> int {      // remaining: per live problem, 0 = nothing left to do for it (or null)
        const int keep = (caches_built && pl.wide) ? pl.cache_keep : 0;
        caches_built = true;
        for (int k = 0; k < nl; k++) {
            if (remaining && remaining[k] == 0) continue;
            const int b = live[k];
            const Chain2Args &a = jobs[b].c2;
            // (runs of identical rows: one cache per run, copied to the rest)
            const int32_t *same = (jobs[b].h_ngroups < n && n >= 2) ? jobs[b].b_same.as<int32_t>() : nullptr;
            if (pl.cache_waves > 0) {
                // (a wave per row; the grid: pl.cache_waves waves on every CU, four to a workgroup)
                const int g = std::max(1, std::min((n + CBW - 1) / CBW, pl.cache_waves * pl.cus / CBW));
                if (pl.cache_unroll == 8)
                    hipLaunchKernelGGL(build_row_caches_wave<8>, dim3(g), dim3(64 * CBW), 0, stream, n, a.ld, a.cost, (const float *)a.fws,
                                       a.cache_col, a.cache_val, a.rowmap, same, pl.cache_stream, keep);
                else
                    hipLaunchKernelGGL(build_row_caches_wave<4>, dim3(g), dim3(64 * CBW), 0, stream, n, a.ld, a.cost, (const float *)a.fws,
                                       a.cache_col, a.cache_val, a.rowmap, same, pl.cache_stream, keep);
            } else if constexpr (CH == 0)
                hipLaunchKernelGGL(build_row_caches_stream, dim3(pl.cache_grid), dim3(BLOCK2), 0, stream, n, a.ld, a.cost,
                                   (const float *)a.fws, a.cache_col, a.cache_val, a.rowmap, same);
            else
                hipLaunchKernelGGL((build_row_caches<CH>), dim3(pl.cache_grid), dim3(BLOCK2), 0, stream, n, a.ld, a.cost,
                                   (const float *)a.fws, a.cache_col, a.cache_val, a.rowmap, same);
            if (same)
                hipLaunchKernelGGL(replicate_group_caches, dim3(std::max(1, std::min((n + 3) / 4, 2048))), dim3(256), 0, stream, n, same,
                                   a.cache_col, a.cache_val);
        }
        CYTO_HIP(hipGetLastError());
        return CYTO_OK;
    };
    // argument blocks of the batch, in `live` order
    std::vector<Chain2Args> h_c2((size_t)nl);
    std::vector<LazyArgs> h_la((size_t)nl);
    size_t shm_aug = 0, shm_lazy = 0;
    for (int k = 0; k < nl; k++) {
        F32Job &j = jobs[live[k]];
        h_c2[k] = j.c2; h_la[k] = j.la;
        if (!LDS_STATE) h_la[k].may_bail = 0;                        // only the dense jv_aug2<CH, true> can take over
        shm_aug = std::max(shm_aug, j.shm_aug); shm_lazy = std::max(shm_lazy, j.shm_lazy);
    }
    if ((rc = d_c2.alloc(sizeof(Chain2Args) * nl, stream)) || (rc = d_la.alloc(sizeof(LazyArgs) * nl, stream))) return rc;
    CYTO_HIP(hipMemcpyAsync(d_c2.p, h_c2.data(), sizeof(Chain2Args) * nl, hipMemcpyHostToDevice, stream));
    CYTO_HIP(hipMemcpyAsync(d_la.p, h_la.data(), sizeof(LazyArgs) * nl, hipMemcpyHostToDevice, stream));

    if ((rc = build_caches(nullptr))) return rc;
    CYTO_HIP(hipEventRecord(ev_cache_done, stream));
    struct ParSlot { std::atomic<int> *p = nullptr; ~ParSlot() { if (p) p->store(0); } } par_slot;       // (released on every return path)
    bool par_slot_held = false;
    int dev_now = 0;
    (void)hipGetDevice(&dev_now);
    const int device_slot = dev_now >= 0 && dev_now < 64 ? dev_now : 0;
    if (pl.wide) {
        // the wide solver (lap_wide.hip): Jacobi reduction transfer, Jacobi rounds of row reduction, speculative shortest-path
        // augmentation -- one launch per phase for the whole batch
        const size_t nT = (size_t)n * sizeof(float);
        std::vector<WideArgs> h_wa((size_t)nl);
        for (int k = 0; k < nl; k++) {
            F32Job &j = jobs[live[k]];
            // Both several-workgroup search kernels (wide_aug_mc: one search on G workgroups; wide_aug<.., PAR>: G searches at once) meet at
            // hand-rolled grid barriers and are launched as plain kernels: their participating workgroups (blocks 0, 8, 16, ... -- one XCD)
            // must all be RESIDENT, one 1024-thread workgroup per CU.  So G is clamped to the CUs of one XCD, and ONE such kernel runs per
            // device at a time, whichever of the two it is: workgroups of two of them could each get partly scheduled and spin for the
            // rest.  A solve that finds the slot taken runs one search at a time on one workgroup (same results).
            const int xcd_cus = std::max(1, pl.cus / 8);
            int mcg = nl != 1 ? 0 : (pl.wide_groups > 0 ? pl.wide_groups : (pl.wide_groups < 0 ? 0 : wide_mc_groups(nl, n)));
            mcg = std::min(mcg, xcd_cus);
            // several searches of ONE problem at once (wide_aug<.., PAR>): by default for a single problem without runs of identical
            // rows (those finish their searches as runs of one-edge steps in the one-workgroup kernel) from 2 048 rows on
            // (runs of identical rows -- a Visium problem's slots: a tenth as many groups as rows -- finish most of their searches as
            //  runs of one-edge steps in the one-workgroup kernel; a sub-spot chunk's few doubled spots do not matter)
            const bool dup_rows = n >= 2 && (long long)j.h_ngroups * 5 < (long long)n * 4;
            int parg = (nl != 1 || mcg > 0) ? 0 : (pl.wide_par > 0 ? pl.wide_par : (pl.wide_par < 0 || dup_rows || n < 2048 ? 0 : 16));
            parg = std::min(std::min(parg, WIDE_PAR_GMAX), xcd_cus);
            if (parg == 1) parg = 0;
            if (mcg == 1) mcg = 0;
            if (parg > 1 || mcg > 1) {
                if (g_par_busy[device_slot].exchange(1) != 0) { parg = 0; mcg = 0; }
                else par_slot_held = true;
            }
            if (par_slot_held) par_slot.p = &g_par_busy[device_slot];
            const size_t mcb = mcg > 0 ? wide_mc_state_bytes(n) : 0;
            const size_t parb = parg > 0 ? wide_par_state_bytes(n, parg) : 0;
            const size_t sc_off = ((2 * nT + 255) / 256) * 256 + mcb;       // the phase machine's control block behind everything else
            const size_t par_off = sc_off + WIDE_SC_BYTES;
            const size_t scx_off = ((par_off + parb + 255) / 256) * 256;     // the machine's own arrays (lap_wide.hip: ScMem)
            if ((rc = j.b_wide.alloc(scx_off + wide_sc_ext_bytes(n), stream))) return rc;
            const Chain2Args &c = j.c2;
            WideArgs &wa = h_wa[k];
            wa.n = n; wa.ld = c.ld; wa.cost = c.cost; wa.rowmap = c.rowmap;
            wa.v = c.fws; wa.u = c.fws + n; wa.cassign = c.fws + 3 * (int64_t)n;
            wa.label = reinterpret_cast<unsigned long long *>(c.fws + 4 * (int64_t)n);
            wa.rowsol = c.iws; wa.colsol = c.iws + n; wa.matches = c.iws + 2 * (int64_t)n; wa.freerows = c.iws + 3 * (int64_t)n;
            wa.act0 = c.iws + 4 * (int64_t)n; wa.act1 = c.iws + 5 * (int64_t)n; wa.touched = c.iws + 6 * (int64_t)n;
            wa.slot_j = c.iws + 7 * (int64_t)n;
            wa.bid = reinterpret_cast<unsigned long long *>(c.iws + 8 * (int64_t)n);
            wa.slot_p = j.b_wide.as<float>(); wa.slot_c = wa.slot_p + n;
            wa.cache_col = c.cache_col; wa.cache_val = c.cache_val; wa.misc = c.misc;
            wa.max_rounds = pl.wide_rounds;
            // (the several-searches kernel never pauses -- it never looks at aug_seg --, but the row reduction's rebuild rule reads the same field)
            wa.aug_seg = mcg > 0 ? -1 : pl.wide_rebuild;
            // what a rebuild costs, in full-row relaxations of ONE workgroup (a relaxation sweeps n elements at ~5 G/s; a pass costs
            // ~0.4 ms of launches and synchronisation plus one read of every unfinished problem's matrix by the whole chip, ~100x
            // faster per element -- and the problems of a batch are rebuilt one after the other while their workgroups all wait)
            wa.aug_waste = (int)std::min<long long>(1 << 30, std::max<long long>(16, (2000000ll + (long long)nl * n * n / 100) / n));
            wa.arr_waste = std::max(8, wa.aug_waste / 3);          // (a full-row bid with its cache refresh: three sweeps)
            if (CYTO_KNOB("CYTO_ARR_WASTE").set) wa.arr_waste = std::max(1, CYTO_KNOB("CYTO_ARR_WASTE").value);     // (developer knob, read once per process: tools/batch_chunks_bench.py)
            wa.seg_quorum = nl > 1 ? std::max(1, nl / 4) : 0;
            wa.seg_sync = nullptr;
            wa.same_prev = (j.h_ngroups < n && n >= 2) ? j.b_same.as<int32_t>() : nullptr;
            wa.sc = j.b_wide.as<char>() + sc_off;
            CYTO_HIP(hipMemsetAsync(wa.sc, 0, WIDE_SC_BYTES, stream));
            wa.scx = j.b_wide.as<char>() + scx_off;
            CYTO_HIP(hipMemsetAsync(wa.scx, 0, wide_sc_ext_bytes(n), stream));
            CYTO_HIP(hipMemsetAsync(wa.scx, 0xFF, wide_sc_ones_bytes(n), stream));     // (the machine's bid words and lowest-bid arrays)
            wa.par_groups = parg; wa.par = nullptr;
            if (parg > 0) {
                const size_t np_ = ((size_t)n + 63) & ~(size_t)63;
                wa.par = j.b_wide.as<char>() + par_off;
                CYTO_HIP(hipMemsetAsync(wa.par, 0, 256, stream));                                            // the control block ...
                const int p_init[2] = {WIDE_PAR_GMAX + 1, WIDE_PAR_GMAX + 1};
                CYTO_HIP(hipMemcpyAsync(wa.par + 8, p_init, sizeof p_init, hipMemcpyHostToDevice, stream));   // ... P[0], P[1]: "no conflict"
                CYTO_HIP(hipMemsetAsync(wa.par + 256, 0xFF, (size_t)parg * np_ * 8, stream));                  // every search's labels: all-ones
                CYTO_HIP(hipMemsetAsync(wa.par + 256 + (size_t)parg * np_ * 20, 0, np_ * 4, stream));          // the claim words
                CYTO_HIP(hipStreamSynchronize(stream));                                                         // (p_init is a local)
            }
            wa.mc_groups = mcg; wa.gbmin = nullptr; wa.gdirty = nullptr; wa.gasg = nullptr; wa.gdense = nullptr; wa.ctl = nullptr;
            if (mcg > 0) {
                const size_t nblk = ((size_t)n + 63) / 64, nw32 = ((size_t)n + 31) / 32;
                char *base = j.b_wide.as<char>() + ((2 * nT + 255) / 256) * 256;
                wa.gbmin = reinterpret_cast<unsigned long long *>(base);          // nblk records of 128 bytes: [0] block minimum, [1] dirty bits
                wa.gasg = reinterpret_cast<uint32_t *>(base + nblk * 128);
                wa.gdense = wa.gasg + nw32; wa.gdirty = nullptr;
                wa.ctl = base + wide_mc_state_bytes(n) - 256;
                CYTO_HIP(hipMemsetAsync(wa.gbmin, 0, nblk * 128 + 2 * nw32 * 4, stream));
                CYTO_HIP(hipMemset2DAsync(wa.gbmin, 128, 0xFF, 8, nblk, stream));   // the block minima: all-ones
                CYTO_HIP(hipMemsetAsync(wa.ctl, 0, 256, stream));
            }
            // the post-column-reduction prices: snapshot for the reduction transfer AND the raw cost of every owner entry
            CYTO_HIP(hipMemcpyAsync(wa.cassign, wa.v, nT, hipMemcpyDeviceToDevice, stream));
            CYTO_HIP(hipMemsetAsync(wa.label, 0xFF, 2 * nT, stream));
            CYTO_HIP(hipMemsetAsync(wa.bid, 0xFF, 2 * nT, stream));
        }
        DevBuf d_wa, d_sync;
        if ((rc = d_wa.alloc(sizeof(WideArgs) * nl, stream)) || (rc = d_sync.alloc(sizeof(int32_t) * ((size_t)nl + 1), stream))) return rc;
        for (WideArgs &wa : h_wa) wa.seg_sync = d_sync.as<int32_t>();
        CYTO_HIP(hipMemcpyAsync(d_wa.p, h_wa.data(), sizeof(WideArgs) * nl, hipMemcpyHostToDevice, stream));
        if ((rc = wide_launch_rt(d_wa.as<WideArgs>(), nl, n, stream))) return rc;
        std::vector<int32_t> h_sync((size_t)nl + 1, 1);
        std::vector<char> caches_fresh((size_t)nl, 0);            // (wide_arr's word: hardly a full-row bid -- no rebuild before the searches)
        using BuildFn = decltype(build_caches);
        auto rebuild_tramp = +[](void *ctx, const int32_t *flags) -> int { return (*static_cast<BuildFn *>(ctx))(reinterpret_cast<const int *>(flags)); };
        for (int pass = 0;; pass++) {                              // the row-reduction rounds (they pause when the caches have gone stale)
            if (pass && (rc = build_caches(h_sync.data() + 1))) return rc;
            CYTO_HIP(hipMemsetAsync(d_sync.p, 0, sizeof(int32_t), stream));
            if ((rc = wide_launch_arr(d_wa.as<WideArgs>(), nl, n, stream, pl.wide_wipe, pass > 0, d_sync.as<int32_t>(), rebuild_tramp, &build_caches, nl == 1 ? &h_wa[0] : nullptr))) return rc;
            if (h_wa[0].aug_seg != 0) break;                       // (no pauses asked for: nothing to wait for)
            CYTO_HIP(hipMemcpyAsync(h_sync.data(), d_sync.p, sizeof(int32_t) * ((size_t)nl + 1), hipMemcpyDeviceToHost, stream));
            CYTO_HIP(hipStreamSynchronize(stream));
            bool done = true;
            for (int k = 0; k < nl; k++) { done = done && (h_sync[(size_t)k + 1] & 1) == 0; caches_fresh[(size_t)k] = (h_sync[(size_t)k + 1] & 2) != 0; h_sync[(size_t)k + 1] &= 1; }
            if (done) break;
        }
        // the searches, in as many launches as they ask for: wide_aug returns when its row caches have gone stale (lap_wide.hip) and
        // the whole chip rebuilds them against the prices reached -- only for the problems that still have searches to run
        for (int pass = 0;; pass++) {
            if (pass == 0) {
                std::vector<int> need((size_t)nl);
                for (int k = 0; k < nl; k++) need[(size_t)k] = caches_fresh[(size_t)k] ? 0 : 1;
                if ((rc = build_caches(need.data()))) return rc;
            } else if ((rc = build_caches(reinterpret_cast<const int *>(h_sync.data() + 1)))) return rc;
            if (pass == 0) CYTO_HIP(hipEventRecord(ev_arr_done, stream));   // (ms_aug: the search kernel -- and what later passes add)
            if (pass == 0 && h_wa[0].mc_groups == 0 && h_wa[0].par_groups == 0) {
                // repeated spot rows (CytoSPACE's slots): their one-edge searches all at once, before the search kernel
                bool dup = false;
                for (int k = 0; k < nl; k++) dup = dup || h_wa[(size_t)k].same_prev != nullptr;
                if (dup && (rc = wide_launch_claims(d_wa.as<WideArgs>(), nl, n, stream, d_sync.as<int32_t>()))) return rc;
            }
            CYTO_HIP(hipMemsetAsync(d_sync.p, 0, sizeof(int32_t), stream));
            if ((rc = wide_launch_aug(d_wa.as<WideArgs>(), nl, n, stream, h_wa[0].mc_groups, h_wa[0].par_groups))) return rc;
            if (h_wa[0].mc_groups > 0 || h_wa[0].par_groups > 0) { CYTO_HIP(hipStreamSynchronize(stream)); break; }
            CYTO_HIP(hipMemcpyAsync(h_sync.data(), d_sync.p, sizeof(int32_t) * ((size_t)nl + 1), hipMemcpyDeviceToHost, stream));
            CYTO_HIP(hipStreamSynchronize(stream));                // (d_wa is read by the kernels until here)
            bool done = true;
            for (int k = 0; k < nl; k++) done = done && h_sync[(size_t)k + 1] == 0;
            if (done) break;
        }
        return CYTO_OK;
    }
    if ((rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern)))) return rc;
    if (pl.shm_chain > (size_t)LDS_DYNAMIC_MAX || shm_aug > (size_t)LDS_DYNAMIC_MAX || shm_lazy > (size_t)LDS_DYNAMIC_MAX) return CYTO_ERR_INTERNAL;
    hipLaunchKernelGGL(kern, dim3(nl), dim3(BLOCK2), pl.shm_chain, stream, d_c2.as<Chain2Args>());
    CYTO_HIP(hipGetLastError());
    CYTO_HIP(hipEventRecord(ev_arr_done, stream));
    auto launch_dense = [&](const Chain2Args *d_args, int count) -> int {
        if constexpr (LDS_STATE) {
            void (*ka)(const Chain2Args *) = jv_aug2<CH, true, BLOCK2>;
            int r2 = set_max_dynamic_lds(reinterpret_cast<const void *>(ka));
            if (r2) return r2;
            hipLaunchKernelGGL(ka, dim3(count), dim3(BLOCK2), shm_aug, stream, d_args);
            CYTO_HIP(hipGetLastError());
            return CYTO_OK;
        } else {
            (void)d_args; (void)count;
            return CYTO_ERR_INTERNAL;     // (beyond the LDS-resident sizes the cache-certified search is always used)
        }
    };
    if (!pl.lazy) return launch_dense(d_c2.as<Chain2Args>(), nl);

    // fresh caches (floors against the prices the augmentation starts from), then the cache-certified search
    if ((rc = build_caches(nullptr))) return rc;
    void (*lk)(const LazyArgs *) = pl.lz_lds_state ? jv_aug_lazy<true> : (pl.lz_cs_lds ? jv_aug_lazy<false, true> : jv_aug_lazy<false>);
    if ((rc = set_max_dynamic_lds(reinterpret_cast<const void *>(lk)))) return rc;
    hipLaunchKernelGGL(lk, dim3(nl), dim3(BLOCK2), shm_lazy, stream, d_la.as<LazyArgs>());
    CYTO_HIP(hipGetLastError());
    if constexpr (LDS_STATE) {
        // which searches gave up?  (misc + 128: number of free rows, misc + 136: searches completed)
        bool any_bail = false;
        for (int k = 0; k < nl; k++) any_bail |= h_la[k].may_bail != 0;
        if (any_bail) {
            for (int k = 0; k < nl; k++)
                CYTO_HIP(hipMemcpyAsync(jobs[live[k]].h_hand, jobs[live[k]].c2.misc + 128, sizeof(int) * 3, hipMemcpyDeviceToHost, stream));
            CYTO_HIP(hipStreamSynchronize(stream));
            std::vector<Chain2Args> cont;
            for (int k = 0; k < nl; k++) {
                const F32Job &j = jobs[live[k]];
                if (h_la[k].may_bail && j.h_hand[2] < j.h_hand[0]) { Chain2Args a = h_c2[k]; a.aug_start = j.h_hand[2]; cont.push_back(a); }
            }
            if (!cont.empty()) {
                // (the first nl blocks of d_c2 are no longer read: jv_chain2 has finished)
                CYTO_HIP(hipMemcpyAsync(d_c2.p, cont.data(), sizeof(Chain2Args) * cont.size(), hipMemcpyHostToDevice, stream));
                if ((rc = launch_dense(d_c2.as<Chain2Args>(), (int)cont.size()))) return rc;
                CYTO_HIP(hipStreamSynchronize(stream));     // `cont` is read by the copy until here
            }
        }
    }
    return CYTO_OK;
}

// All problems have the same n.  jobs[b].status carries per-problem failures (non-finite costs, solver status); the
// return value reports failures of the batch as a whole (allocation, HIP).
static int lap_solve_f32_batch(int n, std::vector<F32Job> &jobs, int device_id, hipStream_t stream, const cyto_lap_opts &opts) {
    const int nb = (int)jobs.size();
    if (nb == 0) return CYTO_OK;
    if (n <= 0) return CYTO_ERR_BAD_ARG;
    if (n > FAST_NMAX) return CYTO_ERR_UNSUPPORTED;
    int rc = check_opts(opts);
    if (rc) return rc;
    if ((rc = select_device(device_id))) return rc;
    Events<6> ev;                                   // destroyed on every return path
    if ((rc = ev.create())) return rc;
    const hipEvent_t e0 = ev[0], e1 = ev[1], e1b = ev[2], e1c = ev[3], e1d = ev[4], e2 = ev[5];

    F32Plan pl;
    pl.n = n;
    pl.colblocks = (n + 256 * 4 - 1) / (256 * 4);
    pl.rowblocks = (2048 + pl.colblocks - 1) / pl.colblocks;
    pl.rowblocks = max(1, min(pl.rowblocks, (n + 15) / 16));
    pl.rows_per_block = (n + pl.rowblocks - 1) / pl.rowblocks;
    pl.rowblocks = (n + pl.rows_per_block - 1) / pl.rows_per_block;
    pl.cache_grid = max(1, min(n, 1024));
    // (measured, tools/cache_build_bench.py, gpurun_out/r04v: n = 20 000: 8 waves per CU x 8 quads in flight 0.404 ms = 3.96 TB/s, 50 000:
    //  2.13 ms = 4.69 TB/s -- the workgroup-per-row builders 0.750 / 3.27 ms; n = 10 000: 32 x 4 0.126 ms, 8 x 8 0.137, old 0.183:
    //  with few rows per wave the second sweep of a wave's first row, from L2, costs less than the waves it would take away)
    //  A few-cell-type matrix (tools/cache_build_bench.py 20000 --typed, gpurun_out/r04x) is another matter: its rows sit at different
    //  levels, a neighbour's floor fits 6 % of them, nearly every row is swept twice (the second time from L2 by its one wave) and what
    //  counts is how many waves there are: 8 x 8 1.36 ms, 20 x 8 1.01 ms, 32 x 4 1.02 ms, old 0.97 ms.  20 waves x 8 quads is within
    //  5 % of the best setting of every instance measured (uniform 20 000: 0.412 ms, 50 000: 2.25 ms, 10 000: 0.123 ms).
    //  n = 50 000 prefers 8 waves per CU by more than the build's own time (same box, gpurun_out/r04z: row reduction 21.5-23.0 ms after a
    //  build with 8 waves, 24.0-24.8 ms after one with 20 -- 183 instead of 146 full-row bids, each holding up a round for one 200-KB sweep).
    pl.cache_unroll = 8;
    pl.cache_waves = n > 32768 ? 8 : 20;
    if (CYTO_KNOB("CYTO_CACHE_KEEP").set) pl.cache_keep = std::max(0, std::min(63, CYTO_KNOB("CYTO_CACHE_KEEP").value));     // (developer knob)
    // (cyto_lap_opts.cache_waves / cache_unroll / cache_stream -- tools/cache_build_bench.py and the builder tests: -1 waves selects the
    //  workgroup-per-row builders)
    if (opts.cache_waves != 0) pl.cache_waves = opts.cache_waves < 0 ? 0 : opts.cache_waves;
    if (opts.cache_unroll != 0) pl.cache_unroll = opts.cache_unroll;
    if (opts.cache_stream != 0) pl.cache_stream = opts.cache_stream > 0 ? 1 : 0;
    pl.cus = device_cus(device_id);
    const int per2 = 4 * BLOCK2;
    // which chain variant: by size, or the large-n variants forced at a small n (opts.chain_variant; the test-suite
    // runs them on instances the CPU oracle solves in a second)
    // which solver: the wide one (lap_wide.hip) unless the caller selected the chain solver or one of its kernels explicitly.
    // A batch runs its problems side by side, a workgroup each, through either
    const bool chain_opts = opts.chain_variant || opts.augmentation || opts.no_handover || opts.inject_exceptions ||
                            opts.group_state_global || opts.aux_state_global;
    pl.wide = opts.mode == 2 || (opts.mode == 0 && !chain_opts);
    pl.wide_rounds = opts.wide_rounds < 0 ? 0 : (opts.wide_rounds > 0 ? opts.wide_rounds : 4096 + (long long)n / 4);
    pl.wide_groups = opts.wide_groups;
    pl.wide_rebuild = opts.wide_rebuild;
    pl.wide_wipe = opts.wide_wipe;
    pl.wide_par = opts.wide_par;
    pl.force_l2 = opts.chain_variant != 0;
    pl.no_cs_lds = opts.chain_variant == 3;            // (3: as 2, with colsol in global memory too -- what n > 65 535 uses)
    pl.lds_variant = !pl.force_l2 && n <= 13 * per2;
    // cache-certified augmentation: the default above 5120 columns (measured cross-over with the register-resident dense
    // search on uniform and duplicated-row instances), the only one beyond 26 624
    pl.lazy = opts.augmentation == 2 || !pl.lds_variant || (opts.augmentation == 0 && n > 5120);
    const int npad = (n + 3) & ~3;
    const size_t npad6 = (((size_t)npad * 6) + 15) & ~(size_t)15, npad2 = (((size_t)npad * 2) + 15) & ~(size_t)15;
    const size_t nb24 = (size_t)((((n + 63) / 64) + 511) & ~511) * 8 + (size_t)((n + 63) / 64) * 24 + 16;
    const size_t lds_budget = LDS_DYNAMIC_MAX;
    pl.lz_lds_state = n <= 65535 && !pl.force_l2 && npad6 + nb24 <= lds_budget;
    pl.lz_cs_lds = !pl.lz_lds_state && n <= 65535 && !pl.no_cs_lds && npad2 + nb24 <= lds_budget;
    pl.lz_base_shm = (pl.lz_lds_state ? npad6 : (pl.lz_cs_lds ? npad2 : 0)) + nb24;
    const bool cs_lds_chain = !pl.lds_variant && n <= 65535 && !pl.no_cs_lds;
    pl.shm_chain = pl.lds_variant ? (((size_t)npad * 6 + 15) / 16) * 16 : (cs_lds_chain ? (((size_t)npad * 2 + 15) / 16) * 16 : 16);

    const size_t nT = (size_t)n * sizeof(float), nI = (size_t)n * sizeof(int32_t);
    // ---- stage 1 (every problem, back to back on the stream: full-chip streaming kernels): column reduction, row groups ----
    CYTO_HIP(hipEventRecord(e0, stream));
    for (F32Job &j : jobs) {
        if (!j.cost || j.ld < n) { j.status = CYTO_ERR_BAD_ARG; continue; }
        // optional row map: validated on the host (non-decreasing: duplicated spot rows are contiguous, as np.repeat makes
        // them), the distinct rows in use and the first LAP row of each go to the device with it
        const int nstored = j.rowmap_host ? j.nu : n;
        std::vector<int32_t> ulist, ufirst;
        if (j.rowmap_host) {
            if (j.nu <= 0) { j.status = CYTO_ERR_BAD_ARG; continue; }
            bool ok = true;
            for (int i = 0; i < n && ok; i++) {
                const int32_t r = j.rowmap_host[i];
                ok = r >= 0 && r < j.nu && (i == 0 || r >= j.rowmap_host[i - 1]);
                if (ok && (i == 0 || r != j.rowmap_host[i - 1])) { ulist.push_back(r); ufirst.push_back(i); }
            }
            if (!ok) { j.status = CYTO_ERR_BAD_ARG; continue; }
            j.nused = (int)ulist.size();
        }
        // the kernels want 16-byte aligned rows: pitch a multiple of 4 elements
        j.dcost = j.cost; j.dld = j.ld;
        const bool aligned = j.cost_on_device && (j.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(j.cost) & 15) == 0);
        if (!aligned) {
            j.dld = ((int64_t)n + 3) / 4 * 4;
            if ((rc = j.staged.alloc((size_t)nstored * j.dld * sizeof(float), stream))) return rc;
            CYTO_HIP(hipMemcpy2DAsync(j.staged.p, j.dld * sizeof(float), j.cost, j.ld * sizeof(float), (size_t)n * sizeof(float), nstored,
                                      j.cost_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
            j.dcost = j.staged.as<float>();
        }
        if (j.rowmap_host) {
            if ((rc = j.b_rowmap.alloc(nI, stream)) || (rc = j.b_ulist.alloc((size_t)j.nused * 4, stream)) || (rc = j.b_ufirst.alloc((size_t)j.nused * 4, stream)))
                return rc;
            CYTO_HIP(hipMemcpyAsync(j.b_rowmap.p, j.rowmap_host, nI, hipMemcpyHostToDevice, stream));
            CYTO_HIP(hipMemcpyAsync(j.b_ulist.p, ulist.data(), (size_t)j.nused * 4, hipMemcpyHostToDevice, stream));
            CYTO_HIP(hipMemcpyAsync(j.b_ufirst.p, ufirst.data(), (size_t)j.nused * 4, hipMemcpyHostToDevice, stream));
            CYTO_HIP(hipStreamSynchronize(stream));       // ulist / ufirst are locals
        }
        // workspace (blocks of the device cache: no hipMalloc / hipFree once a size has been seen)
        {
            size_t off = 0;
            auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
            const size_t o_fws = take(6 * nT + 64), o_iws = take(10 * nI + 64), o_imin = take(nI), o_pmin = take((size_t)pl.rowblocks * nT),
                         o_parg = take((size_t)pl.rowblocks * nI), o_misc = take(512), o_same = take(nI), o_gid = take(nI),
                         o_ccol = take((size_t)n * KC * sizeof(uint32_t)), o_cval = take((size_t)n * KC * sizeof(float));
            if ((rc = j.slab.alloc(off, stream))) return rc;
            char *base = j.slab.as<char>();
            j.b_fws.view(base + o_fws, stream); j.b_iws.view(base + o_iws, stream); j.b_imin.view(base + o_imin, stream);
            j.b_pmin.view(base + o_pmin, stream); j.b_parg.view(base + o_parg, stream); j.b_misc.view(base + o_misc, stream);
            j.b_same.view(base + o_same, stream); j.b_gid.view(base + o_gid, stream); j.b_ccol.view(base + o_ccol, stream);
            j.b_cval.view(base + o_cval, stream);
        }
        // float workspace: v | u | ... ; int workspace: rowsol | colsol | matches | freerows | rtrows | pred | ...
        float *d_v = j.b_fws.as<float>();
        int32_t *d_rowsol = j.b_iws.as<int32_t>(), *d_colsol = d_rowsol + n, *d_matches = d_rowsol + 2 * (size_t)n;
        // misc: [0] nonfinite flag (int), [1] chain status (int), [8..16) total (double), [16..) counters, [144] ngroups
        CYTO_HIP(hipMemsetAsync(j.b_misc.p, 0, 512, stream));
        CYTO_HIP(hipMemsetAsync(d_rowsol, 0xFF, nI, stream));
        CYTO_HIP(hipMemsetAsync(d_matches, 0, nI, stream));
        {   // the column minima need every DISTINCT row once: with a row map only the stored rows in use are swept
            const int nrows = j.rowmap_host ? j.nused : n;
            const int rpb = j.rowmap_host ? (nrows + pl.rowblocks - 1) / pl.rowblocks : pl.rows_per_block;
            const int rblocks = j.rowmap_host ? (nrows + rpb - 1) / rpb : pl.rowblocks;
            hipLaunchKernelGGL(colred_partial<float>, dim3(pl.colblocks, rblocks), dim3(256), 0, stream, n, j.dld, j.dcost, rpb,
                               j.b_pmin.as<float>(), j.b_parg.as<int32_t>(), j.b_misc.as<int>(), nrows,
                               j.rowmap_host ? j.b_ulist.as<int32_t>() : (const int32_t *)nullptr,
                               j.rowmap_host ? j.b_ufirst.as<int32_t>() : (const int32_t *)nullptr);
            hipLaunchKernelGGL(colred_finish<float>, dim3((n + 255) / 256), dim3(256), 0, stream, n, rblocks, j.b_pmin.as<float>(),
                               j.b_parg.as<int32_t>(), d_v, j.b_imin.as<int32_t>(), d_rowsol, d_matches);
        }
        hipLaunchKernelGGL(colred_assign, dim3((n + 255) / 256), dim3(256), 0, stream, n, j.b_imin.as<int32_t>(), d_rowsol, d_colsol);
    }
    CYTO_HIP(hipGetLastError());
    CYTO_HIP(hipEventRecord(e1, stream));
    // duplicate-row groups: runs of bitwise identical consecutive rows
    const bool want_groups = n >= 2;
    for (F32Job &j : jobs) {
        if (!j.alive()) continue;
        j.h_ngroups = n;
        if (want_groups) {
            if (j.rowmap_host)
                hipLaunchKernelGGL(rows_same_from_map, dim3((n + 255) / 256), dim3(256), 0, stream, n, j.b_rowmap.as<int32_t>(), j.b_same.as<int32_t>());
            else
                hipLaunchKernelGGL(rows_same_as_prev<float>, dim3(min(n, 2048)), dim3(256), 0, stream, n, j.dld, j.dcost, j.b_same.as<int32_t>());
            hipLaunchKernelGGL(rows_group_ids, dim3(1), dim3(1024), 0, stream, n, j.b_same.as<int32_t>(), j.b_gid.as<int32_t>(),
                               j.b_misc.as<int>() + 36);   // misc + 144
        }
        // a non-finite cost makes every later comparison meaningless: that problem stops before the chain
        CYTO_HIP(hipMemcpyAsync(&j.h_nonfinite, j.b_misc.as<int>(), sizeof(int), hipMemcpyDeviceToHost, stream));
        if (want_groups) CYTO_HIP(hipMemcpyAsync(&j.h_ngroups, j.b_misc.as<int>() + 36, sizeof(int), hipMemcpyDeviceToHost, stream));
    }
    CYTO_HIP(hipGetLastError());
    CYTO_HIP(hipStreamSynchronize(stream));
    CYTO_HIP(hipEventRecord(e1b, stream));

    // ---- stage 2: per-problem argument blocks ----
    for (F32Job &j : jobs) {
        if (!j.alive()) continue;
        if (j.h_nonfinite) { j.status = CYTO_ERR_NONFINITE; continue; }
        float *d_v = j.b_fws.as<float>(), *d_u = d_v + n;
        int32_t *d_rowsol = j.b_iws.as<int32_t>(), *d_colsol = d_rowsol + n, *d_free = d_rowsol + 3 * (size_t)n;
        const int ng = j.h_ngroups;
        Chain2Args &c2 = j.c2;
        c2.n = n; c2.ld = j.dld; c2.cost = j.dcost; c2.fws = d_v; c2.iws = d_rowsol;
        c2.cache_col = j.b_ccol.as<uint32_t>(); c2.cache_val = j.b_cval.as<float>(); c2.misc = j.b_misc.as<char>();
        // duplicate-row skip: per-group state in LDS when it fits beside v (4 B) and colsol (2 B) per column
        c2.rowgid = j.b_gid.as<int32_t>(); c2.ngroups = ng; c2.gmode = 0; c2.g_hbest = nullptr; c2.g_hstamp = nullptr;
        c2.auxlds = 0; c2.aug_start = 0;
        c2.rowmap = j.rowmap_host ? j.b_rowmap.as<int32_t>() : nullptr;
        LazyArgs &la = j.la;
        memset(&la, 0, sizeof la);
        j.shm_lazy = pl.lz_base_shm;
        if (pl.lazy) {
            la.n = n; la.ld = j.dld; la.cost = j.dcost; la.gv = d_v; la.gu = d_u; la.sumvd = d_v + 2 * (int64_t)n;
            la.cassign = d_v + 3 * (int64_t)n; la.dkey = reinterpret_cast<uint64_t *>(d_v + 4 * (int64_t)n);
            la.rowsol = d_rowsol; la.colsol = d_colsol; la.freerows = d_free;
            la.srow = d_rowsol + 6 * (int64_t)n;      // [n+1]: runs into the next slot, which the lazy path does not use
            la.slist = d_rowsol + 8 * (int64_t)n; la.slevel = d_rowsol + 9 * (int64_t)n;
            la.rowgid = j.b_gid.as<int32_t>(); la.cache_col = j.b_ccol.as<uint32_t>(); la.cache_val = j.b_cval.as<float>();
            la.rowmap = j.rowmap_host ? j.b_rowmap.as<int32_t>() : nullptr;
            la.misc = j.b_misc.as<char>(); la.ngroups = ng; la.gmode = 0; la.g_hbest = nullptr; la.g_hstamp = nullptr;
            la.may_bail = opts.no_handover ? 0 : 1;   // (launch_batch clears it where no dense kernel can take over)
            la.debug_exc = opts.inject_exceptions;
            if (want_groups && ng < n) {
                if (!opts.group_state_global && j.shm_lazy + (size_t)ng * 8 <= lds_budget) { la.gmode = 1; j.shm_lazy += (size_t)ng * 8; }
                else {
                    la.gmode = 2;
                    if ((rc = j.b_lzhb.alloc((size_t)ng * 4, stream)) || (rc = j.b_lzhs.alloc((size_t)ng * 4, stream))) return rc;
                    CYTO_HIP(hipMemsetAsync(j.b_lzhs.p, 0, (size_t)ng * 4, stream));
                    la.g_hbest = j.b_lzhb.as<float>(); la.g_hstamp = j.b_lzhs.as<int32_t>();
                }
            }
        }
        if (want_groups && ng < n) {
            // LDS beside the group state: v (4 B) + colsol (2 B) per column on the LDS-resident path
            const size_t lds_state = pl.lds_variant ? (size_t)npad * 6 : (size_t)((n + 31) / 32) * 4;
            if (!opts.group_state_global && lds_state + (size_t)ng * 8 + 8192 <= 160 * 1024) c2.gmode = 1;
            else {
                c2.gmode = 2;
                if ((rc = j.b_ghb.alloc((size_t)ng * 4, stream)) || (rc = j.b_ghs.alloc((size_t)ng * 4, stream))) return rc;
                CYTO_HIP(hipMemsetAsync(j.b_ghs.p, 0, (size_t)ng * 4, stream));
                c2.g_hbest = j.b_ghb.as<float>(); c2.g_hstamp = j.b_ghs.as<int32_t>();
            }
        }
        // dense augmentation: its LDS (prices, colsol, per-group state, and the per-column auxiliaries when they fit)
        const size_t base_aug = (((pl.lds_variant ? (size_t)npad * 6 : 0) + (c2.gmode == 1 ? (size_t)ng * 8 : 0) + 15) / 16) * 16;
        c2.auxlds = (!opts.aux_state_global && pl.lds_variant && ng < 65536 && base_aug + (size_t)npad * 6 + 4096 <= 160 * 1024) ? 1 : 0;
        j.shm_aug = base_aug + (c2.auxlds ? (size_t)npad * 6 : 0) + 32;
    }

    // ---- stage 3: the chains, one launch per phase for the whole batch ----
    DevBuf d_c2, d_la;
    if (opts.chain_variant >= 2) rc = launch_batch<0, false>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (pl.force_l2 && n <= 5 * per2) rc = launch_batch<5, false>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (pl.force_l2 && n <= 16 * per2) rc = launch_batch<16, false>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (pl.force_l2) rc = launch_batch<0, false>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (n <= 2 * per2) rc = launch_batch<2, true>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (n <= 5 * per2) rc = launch_batch<5, true>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (n <= 10 * per2) rc = launch_batch<10, true>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (n <= 13 * per2) rc = launch_batch<13, true>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else if (n <= 16 * per2) rc = launch_batch<16, false>(pl, jobs, stream, e1c, e1d, d_c2, d_la);
    else rc = launch_batch<0, false>(pl, jobs, stream, e1c, e1d, d_c2, d_la);   // streaming dense refresh, any n
    if (rc) return rc;
    CYTO_HIP(hipEventRecord(e2, stream));
    CYTO_HIP(hipStreamSynchronize(stream));

    // ---- results ----
    float ms_colred = 0, ms_cache = 0, ms_chain = 0, ms_arr = 0, ms_aug = 0;
    bool any_alive = false;
    for (const F32Job &j : jobs) any_alive |= j.alive();
    (void)hipEventElapsedTime(&ms_colred, e0, e1);
    if (any_alive) {
        (void)hipEventElapsedTime(&ms_cache, e1b, e1c); (void)hipEventElapsedTime(&ms_chain, e1c, e2);
        (void)hipEventElapsedTime(&ms_arr, e1c, e1d); (void)hipEventElapsedTime(&ms_aug, e1d, e2);
    }
    for (F32Job &j : jobs) {
        if (!j.alive()) continue;
        int h_status = 0;
        long long h_counters[C3_NCOUNTERS] = {0};
        int32_t *d_rowsol = j.b_iws.as<int32_t>();
        float *d_v = j.b_fws.as<float>();
        CYTO_HIP(hipMemcpy(&h_status, j.b_misc.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost));
        CYTO_HIP(hipMemcpy(h_counters, j.b_misc.as<char>() + 16, sizeof(h_counters), hipMemcpyDeviceToHost));
        double h_gap[3] = {-1.0, -1.0, -1.0};
        if (opts.certify && j.info && !h_status) {
            // the float64 certificate (above): one more streaming pass, on the stream, behind the solve
            DevBuf b_viol;
            if ((rc = b_viol.alloc(((size_t)n + 4) * sizeof(double), stream))) return rc;
            const int g = std::max(1, std::min((n + 3) / 4, pl.cus * 8));
            hipLaunchKernelGGL(dual_gap_rows, dim3(g), dim3(256), 0, stream, n, j.dld, j.dcost, j.rowmap_host ? j.b_rowmap.as<int32_t>() : (const int32_t *)nullptr,
                               d_rowsol, d_v, b_viol.as<double>());
            hipLaunchKernelGGL(dual_gap_finish, dim3(1), dim3(1024), 0, stream, n, b_viol.as<double>(), b_viol.as<double>() + n);
            CYTO_HIP(hipGetLastError());
            CYTO_HIP(hipMemcpyAsync(h_gap, b_viol.as<double>() + n, sizeof h_gap, hipMemcpyDeviceToHost, stream));
            CYTO_HIP(hipStreamSynchronize(stream));
        }
        if (j.rowsol) CYTO_HIP(hipMemcpy(j.rowsol, d_rowsol, nI, hipMemcpyDeviceToHost));
        if (j.colsol) CYTO_HIP(hipMemcpy(j.colsol, d_rowsol + n, nI, hipMemcpyDeviceToHost));
        if (j.u) CYTO_HIP(hipMemcpy(j.u, d_v + n, nT, hipMemcpyDeviceToHost));
        if (j.v) CYTO_HIP(hipMemcpy(j.v, d_v, nT, hipMemcpyDeviceToHost));
        if (j.total) CYTO_HIP(hipMemcpy(j.total, j.b_misc.as<char>() + 8, sizeof(double), hipMemcpyDeviceToHost));
        if (j.info) {
            cyto_lap_info *info = j.info;
            memset(info, 0, sizeof *info);
            // (kernel times of a batch are those of the whole batch: its problems share every launch)
            info->ms_colred = ms_colred; info->ms_cache = ms_cache; info->ms_chain = ms_chain; info->ms_arr = ms_arr; info->ms_aug = ms_aug;
            info->ms_total = info->ms_colred + info->ms_cache + info->ms_chain;
            info->scans_colred = n;
            info->scans_redtransfer = h_counters[C_RT];
            info->scans_arr = h_counters[C_ARR];
            info->scans_aug_init = h_counters[C_AUG_INIT];
            info->scans_aug_relax = h_counters[C_AUG_RELAX];
            info->augmentations = h_counters[C_AUGS];
            info->path_hops = h_counters[C_HOPS];
            info->free_after_colred = h_counters[C_FREE_CR];
            info->free_after_arr1 = h_counters[C_FREE_A1];
            info->free_after_arr2 = h_counters[C_FREE_A2];
            info->hbm_row_reads = 2 * (int64_t)n + h_counters[C_ROWS_READ];
            info->dense_refreshes = h_counters[C2_DENSE_REFRESH];
            info->aug_scans_skipped = h_counters[C2_AUG_SKIPPED];
            info->aug_dense_scans = h_counters[C2_AUG_DENSE];
            info->aug_sparse_inits = h_counters[C2_AUG_SPARSE_INIT];
            info->aug_handover = -1;
            if (pl.lazy) {
                int h[3] = {0, 0, 0};
                CYTO_HIP(hipMemcpy(h, j.b_misc.as<char>() + 128, sizeof h, hipMemcpyDeviceToHost));
                if (h[2] < h[0]) info->aug_handover = h[2];
            }
            info->row_groups = j.h_ngroups;
            info->certified = h_gap[0] >= 0.0 ? 1 : 0;
            info->gap_f64 = h_gap[0] >= 0.0 ? h_gap[0] : 0.0; info->gap_max_f64 = h_gap[0] >= 0.0 ? h_gap[1] : 0.0;
            info->gap_rows = h_gap[0] >= 0.0 ? (int64_t)h_gap[2] : 0;
            if (pl.wide) {
                long long wc[WC_N] = {0};
                CYTO_HIP(hipMemcpy(wc, j.b_misc.as<char>() + 160, sizeof wc, hipMemcpyDeviceToHost));
                info->wide = 1;
                info->wide_rounds = wc[WC_ROUNDS]; info->wide_retired = wc[WC_RETIRED]; info->wide_dense_arr = wc[WC_DENSE_ARR];
                info->wide_dense_aug = wc[WC_DENSE_AUG]; info->wide_aug_rounds = wc[WC_AUG_ROUNDS]; info->wide_aug_settled = wc[WC_AUG_PROCESSED];
                info->wide_trivial = wc[WC_TRIVIAL]; info->wide_verify_passes = wc[WC_VERIFY_PASSES]; info->wide_aug_launches = wc[WC_AUG_LAUNCHES];
                info->aug_handover = -1;
                {   // phase timers the wide kernels keep (100 MHz ticks at misc + 256): diagnostics for tools/wide_large.py
                    long long dbg[16] = {0};
                    CYTO_HIP(hipMemcpy(dbg, j.b_misc.as<char>() + 256, sizeof dbg, hipMemcpyDeviceToHost));
                    info->wide_list_rounds = dbg[0]; info->wide_chain_rounds = dbg[2];
                    info->wide_ms_list = dbg[1] * 1e-5; info->wide_ms_chain = dbg[3] * 1e-5;
                    info->wide_ms_aug_rounds = dbg[8] * 1e-5; info->wide_ms_aug_verify = dbg[9] * 1e-5;
                    info->wide_ms_aug_finish = dbg[10] * 1e-5; info->wide_ms_aug_trivial = dbg[11] * 1e-5;
                    info->wide_arr_launches = dbg[12]; info->wide_scaled = dbg[13]; info->wide_phases = dbg[14];
                    info->wide_par_batches = dbg[15]; info->wide_par_discarded = dbg[7];
                }
            }
        }
        if (h_status) j.status = CYTO_ERR_INTERNAL;
    }
    return CYTO_OK;
}

// ---- float64 (the force_doubles / lapjv_compat precision): the generic chain, one problem ----
template <typename T, int CH, bool LDSCS>
static int launch_chain(const ChainArgs<T> &args, hipStream_t stream) {
    const size_t shmem = LDSCS ? (((size_t)args.n * sizeof(int32_t) + 15) / 16) * 16 : 16;
    auto kern = jv_chain<T, CH, LDSCS>;
    int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(kern));
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(1), dim3(BLOCK), shmem, stream, args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

static int lap_solve_f32(int n, const float *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                         float *u, float *v, double *total, cyto_lap_info *info, int device_id, hipStream_t stream,
                         const cyto_lap_opts &opts);
__global__ void narrow_f64_to_f32(int n, int64_t lds_, const double *__restrict__ src, int64_t ldd, float *__restrict__ dst);
__global__ __launch_bounds__(256) void widen_prices(int n, const float *__restrict__ v32, double *__restrict__ v64) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) v64[j] = (double)v32[j];
}

// float64 (the precision of `lapjv(cost, force_doubles=True)` and of `lap.lapjv`, linear_assignment_solvers.py:13-15, 36).
// By default WARM-started (oracle/jv_oracle.c: jv_oracle_warm_f64): the classic float64 solve spends nearly all its time in the
// price wars of the row reduction, and the prices those wars converge to are known to float32 resolution beforehand -- the matrix is
// narrowed on the device, the float32 wide solver runs on it, its prices (widened exactly) are the start, every row free; the
// float64 augmenting row reduction then takes ~1.2 n steps instead of ~240 n and leaves a handful of rows to the augmentation.
// cyto_lap_opts.mode = 1 (or any chain option): the cold classic chain, the parity reference of rounds 1-3.
static int lap_solve_f64(int n, const double *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                         double *u, double *v, double *total, cyto_lap_info *info, int device_id, hipStream_t stream,
                         const cyto_lap_opts &opts, const float *warm_prices_host = nullptr) {
    // warm_prices_host: the start prices are the caller's (the float32 solve of this very matrix: lap_polish_f64) -- no narrowing, no
    // float32 solve here
    typedef double T;
    constexpr int VW = VecOf<T>::W;
    if (n <= 0 || !cost || ld < n) return CYTO_ERR_BAD_ARG;
    int rc = check_opts(opts);
    if (rc) return rc;
    if (n > FAST_NMAX) return CYTO_ERR_UNSUPPORTED;
    if ((rc = select_device(device_id))) return rc;
    Events<3> ev;
    if ((rc = ev.create())) return rc;
    const hipEvent_t e0 = ev[0], e1 = ev[1], e2 = ev[2];
    DevBuf staged;
    const T *dcost = cost;
    int64_t dld = ld;
    const bool aligned = cost_on_device && (ld % VW == 0) && ((reinterpret_cast<uintptr_t>(cost) & 15) == 0);
    if (!aligned) {
        dld = ((int64_t)n + VW - 1) / VW * VW;
        if ((rc = staged.alloc((size_t)n * dld * sizeof(T), stream))) return rc;
        CYTO_HIP(hipMemcpy2DAsync(staged.p, dld * sizeof(T), cost, ld * sizeof(T), (size_t)n * sizeof(T), n,
                                  cost_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        dcost = staged.as<T>();
    }
    const int colblocks = (n + 256 * VW - 1) / (256 * VW);
    int rowblocks = (2048 + colblocks - 1) / colblocks;
    rowblocks = max(1, min(rowblocks, (n + 15) / 16));
    const int rows_per_block = (n + rowblocks - 1) / rowblocks;
    rowblocks = (n + rows_per_block - 1) / rows_per_block;
    DevBuf b_fws, b_iws, b_imin, b_pmin, b_parg, b_misc;
    const size_t nT = (size_t)n * sizeof(T), nI = (size_t)n * sizeof(int32_t);
    if ((rc = b_fws.alloc(6 * nT + 64, stream)) || (rc = b_iws.alloc(10 * nI + 64, stream)) || (rc = b_imin.alloc(nI, stream)) ||
        (rc = b_pmin.alloc((size_t)rowblocks * nT, stream)) || (rc = b_parg.alloc((size_t)rowblocks * nI, stream)) ||
        (rc = b_misc.alloc(256, stream)))
        return rc;
    T *d_v = b_fws.as<T>(), *d_u = b_fws.as<T>() + n;
    int32_t *d_rowsol = b_iws.as<int32_t>(), *d_colsol = d_rowsol + n, *d_matches = d_rowsol + 2 * (size_t)n;
    int *d_nonfinite = b_misc.as<int>();
    CYTO_HIP(hipMemsetAsync(b_misc.p, 0, 256, stream));
    CYTO_HIP(hipMemsetAsync(d_rowsol, 0xFF, nI, stream));
    CYTO_HIP(hipMemsetAsync(d_matches, 0, nI, stream));
    CYTO_HIP(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(colred_partial<T>, dim3(colblocks, rowblocks), dim3(256), 0, stream, n, dld, dcost, rows_per_block,
                       b_pmin.as<T>(), b_parg.as<int32_t>(), d_nonfinite, n, (const int32_t *)nullptr, (const int32_t *)nullptr);
    int h_nonfinite = 0;
    CYTO_HIP(hipMemcpyAsync(&h_nonfinite, d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, stream));
    CYTO_HIP(hipStreamSynchronize(stream));
    if (h_nonfinite) return CYTO_ERR_NONFINITE;
    bool warm = opts.mode != 1 && opts.chain_variant == 0 && opts.augmentation == 0 && n >= 2;
    double ms_warm = 0.0;
    if (warm) {
        // the float32 wide solve of the narrowed matrix: its prices are the start
        const int64_t ld32 = ((int64_t)n + 3) & ~(int64_t)3;
        DevBuf d32, dv32;
        std::vector<float> h_v32((size_t)n);
        cyto_lap_info li32;
        if (warm_prices_host) {
            memset(&li32, 0, sizeof li32);
            memcpy(h_v32.data(), warm_prices_host, (size_t)n * sizeof(float));
            if ((rc = dv32.alloc((size_t)n * sizeof(float), stream))) return rc;
            rc = CYTO_OK;
        } else {
        if ((rc = d32.alloc((size_t)n * ld32 * sizeof(float), stream)) || (rc = dv32.alloc((size_t)n * sizeof(float), stream))) return rc;
        hipLaunchKernelGGL(narrow_f64_to_f32, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256, n > 32768 ? 32768 : n), dim3(256), 0, stream,
                           n, dld, dcost, ld32, d32.as<float>());
        CYTO_HIP(hipGetLastError());
        rc = lap_solve_f32(n, d32.as<float>(), ld32, 1, nullptr, nullptr, nullptr, h_v32.data(), nullptr, &li32, device_id, stream, k_default_opts);
        }
        if (rc == CYTO_ERR_NONFINITE) { warm = false; rc = CYTO_OK; }       // finite float64 costs beyond float32's range: the cold start
        else if (rc) return rc;
        else {
            ms_warm = li32.ms_total;
            CYTO_HIP(hipMemcpyAsync(dv32.p, h_v32.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(widen_prices, dim3((n + 255) / 256), dim3(256), 0, stream, n, dv32.as<float>(), d_v);
            CYTO_HIP(hipMemsetAsync(d_colsol, 0xFF, nI, stream));     // nothing assigned, every row free (rowsol / matches: above)
            CYTO_HIP(hipGetLastError());
            CYTO_HIP(hipStreamSynchronize(stream));                    // (h_v32 / dv32 / d32 are locals)
        }
    }
    if (!warm) {
        hipLaunchKernelGGL(colred_finish<T>, dim3((n + 255) / 256), dim3(256), 0, stream, n, rowblocks, b_pmin.as<T>(),
                           b_parg.as<int32_t>(), d_v, b_imin.as<int32_t>(), d_rowsol, d_matches);
        hipLaunchKernelGGL(colred_assign, dim3((n + 255) / 256), dim3(256), 0, stream, n, b_imin.as<int32_t>(), d_rowsol, d_colsol);
    }
    CYTO_HIP(hipEventRecord(e1, stream));
    ChainArgs<T> ca;
    ca.n = n; ca.ld = dld; ca.cost = dcost; ca.v = d_v; ca.u = d_u;
    ca.rowsol = d_rowsol; ca.colsol = d_colsol; ca.matches = d_matches;
    ca.freerows = d_rowsol + 3 * (size_t)n; ca.rtrows = d_rowsol + 4 * (size_t)n; ca.pred = d_rowsol + 5 * (size_t)n;
    ca.total = reinterpret_cast<double *>(b_misc.as<char>() + 8);
    ca.counters = reinterpret_cast<long long *>(b_misc.as<char>() + 16);
    ca.status = b_misc.as<int>() + 1;
    ca.dwork = d_v + 4 * (size_t)n; ca.lvl = d_rowsol + 7 * (size_t)n;
    ca.cache_col = nullptr; ca.cache_val = nullptr; ca.cs_lds = 0; ca.v_lds = 0;
    // register-resident chain while it does not spill (n <= 4096), else everything streams from L2 -- with row caches
    // against the post-column-reduction prices for REDUCTION TRANSFER and AUGMENTING ROW REDUCTION
    // (opts.chain_variant == 2: without them, every scan reads its row)
    const int64_t per = (int64_t)VW * BLOCK;
    DevBuf b_ccol, b_cval;
    if (n <= 2 * per && opts.chain_variant == 0) rc = launch_chain<T, 2, true>(ca, stream);
    else {
        if (opts.chain_variant != 2) {
            if ((rc = b_ccol.alloc((size_t)n * WC_KC * sizeof(uint32_t), stream)) || (rc = b_cval.alloc((size_t)n * WC_KC * sizeof(T), stream))) return rc;
            hipLaunchKernelGGL(build_row_caches_wide<T>, dim3(n), dim3(256), 0, stream, n, dld, dcost, d_v, b_ccol.as<uint32_t>(), b_cval.as<T>());
            ca.cache_col = b_ccol.as<uint32_t>(); ca.cache_val = b_cval.as<T>();
        }
        // (chain_variant 3: the caches with colsol in global memory -- what n > 65 535 uses -- forced for the test-suite)
        ca.cs_lds = (ca.cache_col && n <= 65535 && opts.chain_variant != 3) ? 1 : 0;
        const size_t vbytes = (((size_t)n * sizeof(T)) + 15) & ~(size_t)15, csbytes = (((size_t)n * 2 + 15) / 16) * 16;
        ca.v_lds = (ca.cs_lds && vbytes + csbytes <= (size_t)LDS_DYNAMIC_MAX) ? 1 : 0;
        const size_t cs_lds = ca.cs_lds ? csbytes + (ca.v_lds ? vbytes : 0) : 16;
        if ((rc = set_max_dynamic_lds(reinterpret_cast<const void *>(jv_chain_stream<T>)))) return rc;
        hipLaunchKernelGGL(jv_chain_stream<T>, dim3(1), dim3(BLOCK), cs_lds, stream, ca);
        rc = hipGetLastError() == hipSuccess ? CYTO_OK : CYTO_ERR_HIP;
    }
    if (rc) return rc;
    CYTO_HIP(hipEventRecord(e2, stream));
    CYTO_HIP(hipStreamSynchronize(stream));
    int h_status = 0;
    long long h_counters[C_NCOUNTERS] = {0};
    CYTO_HIP(hipMemcpy(&h_status, ca.status, sizeof(int), hipMemcpyDeviceToHost));
    CYTO_HIP(hipMemcpy(h_counters, ca.counters, sizeof(h_counters), hipMemcpyDeviceToHost));
    if (rowsol) CYTO_HIP(hipMemcpy(rowsol, d_rowsol, nI, hipMemcpyDeviceToHost));
    if (colsol) CYTO_HIP(hipMemcpy(colsol, d_colsol, nI, hipMemcpyDeviceToHost));
    if (u) CYTO_HIP(hipMemcpy(u, d_u, nT, hipMemcpyDeviceToHost));
    if (v) CYTO_HIP(hipMemcpy(v, d_v, nT, hipMemcpyDeviceToHost));
    if (total) CYTO_HIP(hipMemcpy(total, ca.total, sizeof(double), hipMemcpyDeviceToHost));
    if (info) {
        memset(info, 0, sizeof *info);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1); info->ms_colred = ms;      // (warm start: the float32 solve's launches and waits included)
        (void)hipEventElapsedTime(&ms, e1, e2); info->ms_chain = ms;
        info->ms_total = info->ms_colred + info->ms_chain;
        info->f64_warm = warm ? 1 : 0; info->f64_warm_ms = ms_warm;
        info->scans_colred = n;
        info->scans_redtransfer = h_counters[C_RT];
        info->scans_arr = h_counters[C_ARR];
        info->scans_aug_init = h_counters[C_AUG_INIT];
        info->scans_aug_relax = h_counters[C_AUG_RELAX];
        info->augmentations = h_counters[C_AUGS];
        info->path_hops = h_counters[C_HOPS];
        info->free_after_colred = h_counters[C_FREE_CR];
        info->free_after_arr1 = h_counters[C_FREE_A1];
        info->free_after_arr2 = h_counters[C_FREE_A2];
        info->hbm_row_reads = n + h_counters[C_ROWS_READ];
        info->aug_handover = -1;
        info->row_groups = n;
    }
    return h_status ? CYTO_ERR_INTERNAL : CYTO_OK;
}

// The float64 POLISH of a float32 solve (cyto_lap_opts.polish): the float32 solvers compare rounded reduced costs, so on a near-tie
// (two assignments whose totals differ by less than the rounding of the float32 duals, ~1e-8) which optimal-looking assignment comes
// out depends on their constants (DESIGN "Tried", round 5: phases ending at 128 rows instead of 64 left the few-cell-type 20 000^2
// instance 2e-8 above its optimum).  When the certificate (dual_gap_rows) cannot prove the result optimal, the matrix is widened to
// float64 on the device (exactly), the float32 prices are the start with every row free, and the float64 augmenting row reduction and
// augmentation of the force_doubles path run from there (any prices with nothing assigned are a valid JV state): ~1.2 n steps.  The
// result is the optimum of the float32 matrix in float64 arithmetic -- indices that depend on the MATRIX, not on the float32 solver.
// Costs n^2 x 8 bytes of device memory and several times the solve: an option, not the default.
__global__ __launch_bounds__(256) void widen_f32_rows(int n, int64_t lds_, const float *__restrict__ src, const int32_t *__restrict__ rowmap,
                                                      int64_t ldd, double *__restrict__ dst) {
    for (int64_t row = blockIdx.y; row < n; row += gridDim.y) {
        const float *__restrict__ r = src + (int64_t)(rowmap ? rowmap[row] : row) * lds_;
        for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) dst[row * ldd + c] = (double)r[c];
    }
}
static int lap_polish_f64(int n, const float *cost, int64_t ld, int cost_on_device, const int32_t *rowmap_host, int nu, const float *v32,
                          int32_t *rowsol, int32_t *colsol, float *u, float *v, double *total, double *ms_out, int device_id,
                          hipStream_t stream) {
    int rc = select_device(device_id);
    if (rc) return rc;
    const int nstored = rowmap_host ? nu : n;
    const int64_t ldd = ((int64_t)n + 1) & ~(int64_t)1;
    DevBuf d64, staged, d_map;
    const float *dsrc = cost;
    int64_t dls = ld;
    if ((rc = d64.alloc((size_t)n * ldd * sizeof(double), stream))) return rc;
    if (!cost_on_device) {
        if ((rc = staged.alloc((size_t)nstored * n * sizeof(float), stream))) return rc;
        CYTO_HIP(hipMemcpy2DAsync(staged.p, (size_t)n * sizeof(float), cost, (size_t)ld * sizeof(float), (size_t)n * sizeof(float), nstored,
                                  hipMemcpyHostToDevice, stream));
        dsrc = staged.as<float>(); dls = n;
    }
    if (rowmap_host) {
        if ((rc = d_map.alloc((size_t)n * sizeof(int32_t), stream))) return rc;
        CYTO_HIP(hipMemcpyAsync(d_map.p, rowmap_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(widen_f32_rows, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256, n > 32768 ? 32768 : n), dim3(256), 0, stream, n, dls, dsrc,
                       rowmap_host ? d_map.as<int32_t>() : (const int32_t *)nullptr, ldd, d64.as<double>());
    CYTO_HIP(hipGetLastError());
    CYTO_HIP(hipStreamSynchronize(stream));
    staged.reset();
    std::vector<double> u64((size_t)n), v64((size_t)n);
    cyto_lap_info li;
    rc = lap_solve_f64(n, d64.as<double>(), ldd, 1, rowsol, colsol, u64.data(), v64.data(), total, &li, device_id, stream, k_default_opts, v32);
    if (rc) return rc;
    if (u) for (int i = 0; i < n; i++) u[i] = (float)u64[(size_t)i];
    if (v) for (int i = 0; i < n; i++) v[i] = (float)v64[(size_t)i];
    if (ms_out) *ms_out = li.ms_total;
    return CYTO_OK;
}

// one float32 problem = a batch of one
static int lap_solve_f32_one(int n, const float *cost, int64_t ld, int cost_on_device, const int32_t *rowmap_host, int nu, int32_t *rowsol,
                             int32_t *colsol, float *u, float *v, double *total, cyto_lap_info *info, int device_id, hipStream_t stream,
                             const cyto_lap_opts &opts) {
    if (n <= 0 || !cost || ld < n) return CYTO_ERR_BAD_ARG;
    std::vector<F32Job> jobs(1);
    F32Job &j = jobs[0];
    j.cost = cost; j.ld = ld; j.cost_on_device = cost_on_device; j.rowmap_host = rowmap_host; j.nu = nu;
    cyto_lap_info li;
    std::vector<float> v_keep;
    cyto_lap_opts o = opts;
    if (opts.polish) { o.certify = 1; if (!v) { v_keep.resize((size_t)n); v = v_keep.data(); } }
    j.rowsol = rowsol; j.colsol = colsol; j.u = u; j.v = v; j.total = total; j.info = (info || opts.polish) ? &li : nullptr;
    int rc = lap_solve_f32_batch(n, jobs, device_id, stream, o);
    rc = rc ? rc : j.status;
    if (!rc && opts.polish && li.certified && li.gap_f64 > 0.0) {
        double ms = 0.0;
        rc = lap_polish_f64(n, cost, ld, cost_on_device, rowmap_host, nu, v, rowsol, colsol, u, v_keep.empty() ? v : nullptr, total, &ms, device_id, stream);
        li.polished = 1; li.polish_ms = ms;
    }
    if (info && (info != &li)) *info = li;
    return rc;
}
static int lap_solve_f32(int n, const float *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                         float *u, float *v, double *total, cyto_lap_info *info, int device_id, hipStream_t stream,
                         const cyto_lap_opts &opts) {
    return lap_solve_f32_one(n, cost, ld, cost_on_device, nullptr, 0, rowsol, colsol, u, v, total, info, device_id, stream, opts);
}

// C ABI helper of cyto_lap_batch_f32 (batch.hip): problems of equal size go through the chains together
int lap_batch_same_n(int n, int nb, const float *const *cost, const int64_t *ld, int cost_on_device, int32_t *const *rowsol,
                     int32_t *const *colsol, float *const *u, float *const *v, double *total, cyto_lap_info *info, int *status,
                     int device_id, hipStream_t stream, const int32_t *const *rowmap, const int *nu, const cyto_lap_opts *opts) {
    std::vector<F32Job> jobs((size_t)nb);
    for (int b = 0; b < nb; b++) {
        F32Job &j = jobs[(size_t)b];
        j.cost = cost[b]; j.ld = ld[b]; j.cost_on_device = cost_on_device;
        if (rowmap && rowmap[b]) { j.rowmap_host = rowmap[b]; j.nu = nu ? nu[b] : 0; }
        j.rowsol = rowsol ? rowsol[b] : nullptr; j.colsol = colsol ? colsol[b] : nullptr;
        j.u = u ? u[b] : nullptr; j.v = v ? v[b] : nullptr;
        j.total = total ? &total[b] : nullptr; j.info = info ? &info[b] : nullptr;
    }
    const int rc = lap_solve_f32_batch(n, jobs, device_id, stream, opts ? *opts : k_default_opts);
    for (int b = 0; b < nb; b++) status[b] = rc ? rc : jobs[(size_t)b].status;
    return rc;
}

// float64 -> float32 of a row-major matrix (the cast numpy's astype(float32) performs: round to nearest even)
__global__ __launch_bounds__(256) void narrow_f64_to_f32(int n, int64_t lds_, const double *__restrict__ src, int64_t ldd,
                                                         float *__restrict__ dst) {
    for (int64_t row = blockIdx.y; row < n; row += gridDim.y)
        for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) dst[row * ldd + c] = (float)src[row * lds_ + c];
}

}  // namespace cyto

extern "C" {

// The call the reference makes, `lapjv(cost_scaled)` with a float64 numpy array solved in float32
// (linear_assignment_solvers.py:38): the matrix goes to the device as it is and is narrowed there, instead of a
// float64 -> float32 pass over 8 n^2 bytes on the host first.
int cyto_lap_f32_from_f64(int n, const double *cost_host, int64_t ld, int32_t *rowsol, int32_t *colsol, float *u, float *v,
                          double *total, cyto_lap_info *info, int device_id, void *stream_) {
    if (n <= 0 || !cost_host || ld < n) return CYTO_ERR_BAD_ARG;
    int rc = cyto::select_device(device_id);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int64_t ldd = ((int64_t)n + 3) & ~(int64_t)3;
    cyto::DevBuf d64, d32;
    if ((rc = d64.alloc((size_t)n * n * sizeof(double), stream)) || (rc = d32.alloc((size_t)n * ldd * sizeof(float), stream))) return rc;
    CYTO_HIP(hipMemcpy2DAsync(d64.p, (size_t)n * sizeof(double), cost_host, (size_t)ld * sizeof(double), (size_t)n * sizeof(double), n,
                              hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(cyto::narrow_f64_to_f32, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256, n > 32768 ? 32768 : n), dim3(256), 0,
                       stream, n, (int64_t)n, d64.as<double>(), ldd, d32.as<float>());
    CYTO_HIP(hipGetLastError());
    CYTO_HIP(hipStreamSynchronize(stream));
    d64.reset();                                           // the float64 copy is not needed during the solve
    return cyto::lap_solve_f32(n, d32.as<float>(), ldd, 1, rowsol, colsol, u, v, total, info, device_id, stream, cyto::k_default_opts);
}

int cyto_lap_f32(int n, const float *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                 float *u, float *v, double *total, cyto_lap_info *info, int device_id, void *stream) {
    return cyto::lap_solve_f32(n, cost, ld, cost_on_device, rowsol, colsol, u, v, total, info, device_id, reinterpret_cast<hipStream_t>(stream), cyto::k_default_opts);
}

int cyto_lap_f64(int n, const double *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                 double *u, double *v, double *total, cyto_lap_info *info, int device_id, void *stream) {
    return cyto::lap_solve_f64(n, cost, ld, cost_on_device, rowsol, colsol, u, v, total, info, device_id, reinterpret_cast<hipStream_t>(stream), cyto::k_default_opts);
}

#ifdef CYTO_ARR_PROF
// profiling build only (tools/prof_arr_step.py): the step-cycle accumulators of the last jv_chain2 launch
int cyto_arr_prof_read(long long *out16) {
    CYTO_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(cyto::g_arr_prof), sizeof(long long) * 16));
    return CYTO_OK;
}
#endif

#ifdef LZ_PROF
// profiling build only (tools/prof_lazy_step.py): the step-cycle accumulators of the last jv_aug_lazy launch
int cyto_lz_prof_read(long long *out24) {
    CYTO_HIP(hipMemcpyFromSymbol(out24, HIP_SYMBOL(cyto::g_lz_prof), sizeof(long long) * 24));
    return CYTO_OK;
}
#endif

#ifdef CYTO_AUG_TRACE
// debugging build only (tools/trace_aug_scans.py): cumulative scans / elided scans after every search of the last jv_aug2 launch
int cyto_aug_trace_read(long long *out, int count) {
    CYTO_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(cyto::g_aug_trace), sizeof(long long) * 2 * (size_t)count));
    return CYTO_OK;
}
#endif

#ifdef CYTO_AUG_PROF
// profiling build only (tools/prof_aug_step.py): read and clear the step-cycle accumulators of the dense augmentation
int cyto_aug_prof_read(long long *out16) {
    long long z[16] = {0};
    CYTO_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(cyto::g_aug_prof), sizeof z));
    CYTO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cyto::g_aug_prof), z, sizeof z));
    return CYTO_OK;
}
#endif

int cyto_lap_f32_rowmap(int n, const float *cost_rows, int64_t ld, int nu, int cost_on_device, const int32_t *rowmap,
                        int32_t *rowsol, int32_t *colsol, float *u, float *v, double *total, cyto_lap_info *info, int device_id,
                        void *stream, const cyto_lap_opts *opts) {
    if (n <= 0 || !cost_rows || ld < n || !rowmap || nu <= 0) return CYTO_ERR_BAD_ARG;
    return cyto::lap_solve_f32_one(n, cost_rows, ld, cost_on_device, rowmap, nu, rowsol, colsol, u, v, total, info, device_id,
                                   reinterpret_cast<hipStream_t>(stream), opts ? *opts : cyto::k_default_opts);
}

int cyto_lap_f32_opts(int n, const float *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                      float *u, float *v, double *total, cyto_lap_info *info, int device_id, void *stream, const cyto_lap_opts *opts) {
    return cyto::lap_solve_f32(n, cost, ld, cost_on_device, rowsol, colsol, u, v, total, info, device_id, reinterpret_cast<hipStream_t>(stream), opts ? *opts : cyto::k_default_opts);
}

int cyto_lap_f64_opts(int n, const double *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                      double *u, double *v, double *total, cyto_lap_info *info, int device_id, void *stream, const cyto_lap_opts *opts) {
    return cyto::lap_solve_f64(n, cost, ld, cost_on_device, rowsol, colsol, u, v, total, info, device_id, reinterpret_cast<hipStream_t>(stream), opts ? *opts : cyto::k_default_opts);
}

}  // extern "C"
