// lap_jv.hip -- Jonker-Volgenant LAP solver for gfx950 (MI355X), hand-written HIP.
//
// Replaces the reference's single third-party call
//     `_, y, _ = lapjv.lapjv(cost_scaled)`
// (/root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:34-40).
// Same four phases, same operand order and the same lowest-index tie-breaking as the CPU
// oracle (oracle/jv_oracle_impl.h), so rowsol/colsol/u/v are bit-identical to it.
//
// Kernel plan
//   colred_partial / colred_finish / colred_assign : COLUMN REDUCTION.  The only phase with
//       N^2 independent work: a full-chip streaming pass (float4 per lane, >=2k workgroups).
//   jv_chain : REDUCTION TRANSFER, AUGMENTING ROW REDUCTION and AUGMENTATION are a chain of
//       dependent row scans (each scan's arg-min chooses the next row).  One persistent
//       1024-thread workgroup runs the chain: dual prices v and Dijkstra distances d live in
//       VGPRs (each lane owns fixed columns), colsol lives in LDS, every row scan is one
//       coalesced 16 B/lane sweep of the cost row from HBM followed by a wave64 shuffle
//       reduction and one cross-wave LDS step.  No grid-wide synchronisation is needed.
//
// Compile with -ffp-contract=off (build.py does): the arithmetic is subtract/compare only,
// but nothing may be re-associated.
#include "cyto_common.h"
#include <math.h>

namespace cyto {

constexpr int BLOCK = 1024;
constexpr int NW = BLOCK / 64;

// counters written by jv_chain (index into ChainState::counters)
enum { C_RT = 0, C_ARR, C_AUG_INIT, C_AUG_RELAX, C_AUGS, C_HOPS, C_FREE_CR, C_FREE_A1, C_FREE_A2, C_ROWS_READ, C_NCOUNTERS };

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; static constexpr int W = 4; };
template <> struct VecOf<double> { using type = double2; static constexpr int W = 2; };

template <typename T> __device__ __forceinline__ T vec_get(const typename VecOf<T>::type &x, int e);
template <> __device__ __forceinline__ float vec_get<float>(const float4 &x, int e) {
    return e == 0 ? x.x : e == 1 ? x.y : e == 2 ? x.z : x.w;
}
template <> __device__ __forceinline__ double vec_get<double>(const double2 &x, int e) { return e == 0 ? x.x : x.y; }

// ------------------------------------------------------------------------------------------
// COLUMN REDUCTION
// ------------------------------------------------------------------------------------------
// Partial column minima over a block of rows.  Lane owns VW consecutive columns; each row is one
// 16-byte load per lane (a wave reads 1 KiB contiguous).  Strict '<' while rows ascend keeps the
// lowest row index on ties, exactly like the oracle's row-wise sweep.
template <typename T>
__global__ __launch_bounds__(256) void colred_partial(int n, int64_t ld, const T *__restrict__ cost,
                                                      int rows_per_block, T *__restrict__ pmin,
                                                      int32_t *__restrict__ parg, int *__restrict__ nonfinite) {
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    const int q = blockIdx.x * 256 + threadIdx.x;  // vector-column index
    const int col0 = q * VW;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
    if (col0 >= n) return;
    T mn[VW];
    int32_t arg[VW];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < VW; e++) { mn[e] = (T)INFINITY; arg[e] = r0; }
    const V *p = reinterpret_cast<const V *>(cost + (int64_t)r0 * ld) + q;
    const int64_t stride = ld / VW;
#pragma unroll 4
    for (int r = r0; r < r1; r++) {
        const V x = *p;
        p += stride;
#pragma unroll
        for (int e = 0; e < VW; e++) {
            const T xe = vec_get<T>(x, e);
            if (col0 + e < n) bad |= !__builtin_isfinite(xe);
            if (xe < mn[e]) { mn[e] = xe; arg[e] = r; }
        }
    }
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
};

// agent-scope relaxed accesses: served by L2, never by the scalar cache or a stale L1 line
__device__ __forceinline__ int32_t ld_i32(const int32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_i32(int32_t *p, int32_t x) {
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
    for (int sweep = 0; sweep < 2; sweep++) {
        int k = 0;
        const int prev = numfree;
        numfree = 0;
        int carry = -1;
        while (carry >= 0 || k < prev) {
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

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <typename T, int CH, bool LDSCS>
static int launch_chain(const ChainArgs<T> &args, hipStream_t stream) {
    const size_t shmem = LDSCS ? (((size_t)args.n * sizeof(int32_t) + 15) / 16) * 16 : 16;
    auto kern = jv_chain<T, CH, LDSCS>;
    CYTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kern, dim3(1), dim3(BLOCK), shmem, stream, args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

template <typename T>
static int lap_solve(int n, const T *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                     T *u, T *v, double *total, cyto_lap_info *info, int device_id, void *stream_) {
    constexpr int VW = VecOf<T>::W;
    if (n <= 0 || !cost || ld < n) return CYTO_ERR_BAD_ARG;
    const int64_t cap = (int64_t)16 * VW * BLOCK;
    if (n > cap) return CYTO_ERR_UNSUPPORTED;
    int rc = select_device(device_id);
    if (rc) return rc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

    // the kernels want 16-byte aligned rows: pitch a multiple of VW elements
    DevBuf staged;
    const T *dcost = cost;
    int64_t dld = ld;
    const bool aligned = cost_on_device && (ld % VW == 0) && ((reinterpret_cast<uintptr_t>(cost) & 15) == 0);
    if (!aligned) {
        dld = ((int64_t)n + VW - 1) / VW * VW;
        if ((rc = staged.alloc((size_t)n * dld * sizeof(T)))) return rc;
        CYTO_HIP(hipMemcpy2DAsync(staged.p, dld * sizeof(T), cost, ld * sizeof(T), (size_t)n * sizeof(T), n,
                                  cost_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        dcost = staged.as<T>();
    }

    // workspace
    const int colblocks = (n + 256 * VW - 1) / (256 * VW);
    int rowblocks = (2048 + colblocks - 1) / colblocks;
    rowblocks = max(1, min(rowblocks, (n + 15) / 16));
    const int rows_per_block = (n + rowblocks - 1) / rowblocks;
    rowblocks = (n + rows_per_block - 1) / rows_per_block;

    DevBuf b_v, b_u, b_rowsol, b_colsol, b_matches, b_imin, b_free, b_rt, b_pred, b_pmin, b_parg, b_misc;
    const size_t nT = (size_t)n * sizeof(T), nI = (size_t)n * sizeof(int32_t);
    if ((rc = b_v.alloc(nT)) || (rc = b_u.alloc(nT)) || (rc = b_rowsol.alloc(nI)) || (rc = b_colsol.alloc(nI)) ||
        (rc = b_matches.alloc(nI)) || (rc = b_imin.alloc(nI)) || (rc = b_free.alloc(nI)) || (rc = b_rt.alloc(nI)) ||
        (rc = b_pred.alloc(nI)) || (rc = b_pmin.alloc((size_t)rowblocks * nT)) ||
        (rc = b_parg.alloc((size_t)rowblocks * nI)) || (rc = b_misc.alloc(256)))
        return rc;
    // misc: [0] nonfinite flag (int), [1] chain status (int), [8..16) total (double), [16..) counters
    int *d_nonfinite = b_misc.as<int>();
    int *d_status = b_misc.as<int>() + 1;
    double *d_total = reinterpret_cast<double *>(b_misc.as<char>() + 8);
    long long *d_counters = reinterpret_cast<long long *>(b_misc.as<char>() + 16);
    CYTO_HIP(hipMemsetAsync(b_misc.p, 0, 256, stream));
    CYTO_HIP(hipMemsetAsync(b_rowsol.p, 0xFF, nI, stream));
    CYTO_HIP(hipMemsetAsync(b_matches.p, 0, nI, stream));

    hipEvent_t e0, e1, e2;
    CYTO_HIP(hipEventCreate(&e0));
    CYTO_HIP(hipEventCreate(&e1));
    CYTO_HIP(hipEventCreate(&e2));
    auto cleanup = [&]() { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); };

    CYTO_HIP(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(colred_partial<T>, dim3(colblocks, rowblocks), dim3(256), 0, stream, n, dld, dcost, rows_per_block,
                       b_pmin.as<T>(), b_parg.as<int32_t>(), d_nonfinite);
    hipLaunchKernelGGL(colred_finish<T>, dim3((n + 255) / 256), dim3(256), 0, stream, n, rowblocks, b_pmin.as<T>(),
                       b_parg.as<int32_t>(), b_v.as<T>(), b_imin.as<int32_t>(), b_rowsol.as<int32_t>(), b_matches.as<int32_t>());
    hipLaunchKernelGGL(colred_assign, dim3((n + 255) / 256), dim3(256), 0, stream, n, b_imin.as<int32_t>(),
                       b_rowsol.as<int32_t>(), b_colsol.as<int32_t>());
    CYTO_HIP(hipEventRecord(e1, stream));

    // a non-finite cost makes every later comparison meaningless: stop before the chain
    int h_nonfinite = 0;
    CYTO_HIP(hipMemcpyAsync(&h_nonfinite, d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, stream));
    CYTO_HIP(hipStreamSynchronize(stream));
    if (h_nonfinite) { cleanup(); return CYTO_ERR_NONFINITE; }

    ChainArgs<T> ca;
    ca.n = n; ca.ld = dld; ca.cost = dcost; ca.v = b_v.as<T>(); ca.u = b_u.as<T>();
    ca.rowsol = b_rowsol.as<int32_t>(); ca.colsol = b_colsol.as<int32_t>(); ca.matches = b_matches.as<int32_t>();
    ca.freerows = b_free.as<int32_t>(); ca.rtrows = b_rt.as<int32_t>(); ca.pred = b_pred.as<int32_t>();
    ca.total = d_total; ca.counters = d_counters; ca.status = d_status;

    hipEvent_t e1b;
    CYTO_HIP(hipEventCreate(&e1b));
    CYTO_HIP(hipEventRecord(e1b, stream));
    const int64_t per = (int64_t)VW * BLOCK;
    if (n <= 2 * per) rc = launch_chain<T, 2, true>(ca, stream);
    else if (n <= 5 * per) rc = launch_chain<T, 5, true>(ca, stream);
    else if (n <= 8 * per) rc = launch_chain<T, 8, true>(ca, stream);
    else rc = launch_chain<T, 16, false>(ca, stream);
    if (rc) { cleanup(); (void)hipEventDestroy(e1b); return rc; }
    CYTO_HIP(hipEventRecord(e2, stream));
    CYTO_HIP(hipStreamSynchronize(stream));

    int h_status = 0;
    long long h_counters[C_NCOUNTERS];
    CYTO_HIP(hipMemcpy(&h_status, d_status, sizeof(int), hipMemcpyDeviceToHost));
    CYTO_HIP(hipMemcpy(h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost));
    if (rowsol) CYTO_HIP(hipMemcpy(rowsol, b_rowsol.p, nI, hipMemcpyDeviceToHost));
    if (colsol) CYTO_HIP(hipMemcpy(colsol, b_colsol.p, nI, hipMemcpyDeviceToHost));
    if (u) CYTO_HIP(hipMemcpy(u, b_u.p, nT, hipMemcpyDeviceToHost));
    if (v) CYTO_HIP(hipMemcpy(v, b_v.p, nT, hipMemcpyDeviceToHost));
    if (total) CYTO_HIP(hipMemcpy(total, d_total, sizeof(double), hipMemcpyDeviceToHost));
    if (info) {
        memset(info, 0, sizeof *info);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1); info->ms_colred = ms;
        (void)hipEventElapsedTime(&ms, e1b, e2); info->ms_chain = ms;
        info->ms_total = info->ms_colred + info->ms_chain;
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
    }
    cleanup();
    (void)hipEventDestroy(e1b);
    return h_status ? CYTO_ERR_INTERNAL : CYTO_OK;
}

}  // namespace cyto

extern "C" {

int cyto_lap_f32(int n, const float *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                 float *u, float *v, double *total, cyto_lap_info *info, int device_id, void *stream) {
    return cyto::lap_solve<float>(n, cost, ld, cost_on_device, rowsol, colsol, u, v, total, info, device_id, stream);
}

int cyto_lap_f64(int n, const double *cost, int64_t ld, int cost_on_device, int32_t *rowsol, int32_t *colsol,
                 double *u, double *v, double *total, cyto_lap_info *info, int device_id, void *stream) {
    return cyto::lap_solve<double>(n, cost, ld, cost_on_device, rowsol, colsol, u, v, total, info, device_id, stream);
}

}  // extern "C"
