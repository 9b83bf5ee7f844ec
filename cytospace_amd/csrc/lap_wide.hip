// lap_wide.hip -- the WIDE form of the Jonker-Volgenant solve for gfx950 (MI355X), hand-written HIP.
//
// Same call it replaces as lap_jv.hip:  `_, y, _ = lapjv.lapjv(cost_scaled)`
// (/root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:34-40), same four phases.  The chain
// solver of lap_jv.hip follows the classic Gauss-Seidel order (one dependent row scan after the other: one wave of one
// CU busy); this file computes the order-free restatement of oracle/jv_oracle_impl.h ("WIDE MODE"), whose phases have
// width, bit for bit:
//
//   wide_rt    REDUCTION TRANSFER, Jacobi: every row that owns exactly one column takes its margin against the
//              post-column-reduction prices (snapshot), all margins are subtracted at once.  A wave per row, full chip.
//   wide_arr   AUGMENTING ROW REDUCTION as Jacobi rounds of an eps = 0 auction: every active free row bids for its best
//              column against the round's price snapshot; per column the lowest (price, row) wins, the displaced owner is
//              active in the next round.  One 16-wave workgroup per problem, a wave per bidding row, 16 bids in flight;
//              a round is a pure function of the state, so no visiting order exists to be reproduced.
//   wide_aug   AUGMENTATION: per free row a shortest-path search whose labels are the unique fixed point of a monotone
//              system (succ-clamped candidates), so the search is run SPECULATIVELY: every round each of the 16 waves
//              settles the best unsettled column of the column blocks it owns and relaxes that column's owner row from
//              its row cache; a label that later improves is simply settled again.  Any schedule reaches the oracle's
//              Dijkstra labels.  Row caches certify the scans (floor > final distance, checked when the search has
//              converged; rows that fail are relaxed from their full cost row and the search continues).
//
// Row caches: lap_jv.hip (build_row_caches) -- <= 63 columns per row with their raw costs, sorted by column, and a floor
// that bounds the reduced cost of every other column for as long as prices only decrease (they do: the price update of
// this mode is clamped).
#include "lap_dev.h"
#include "lap_wide.h"
#include <algorithm>

namespace cyto {

namespace {

constexpr int WT = 1024;          // threads of the persistent workgroups: 16 waves
constexpr int WNW = WT / 64;
constexpr int RTB = 256;          // threads of the reduction-transfer workgroups

// counters shared with lap_jv.hip (misc + 16, long long each)
enum { C_RT = 0, C_ARR, C_AUG_INIT, C_AUG_RELAX, C_AUGS, C_HOPS, C_FREE_CR, C_FREE_A1, C_FREE_A2, C_ROWS_READ };

template <typename T> __device__ __forceinline__ T ld_sc1(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int64_t wrow_off(const int32_t *__restrict__ rowmap, int i, int64_t ld) {
    return (int64_t)(rowmap ? rowmap[i] : i) * ld;
}
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// REDUCTION TRANSFER (Jacobi).  v0 = the post-column-reduction prices (a copy lives in a.cassign, which also is the
// raw cost of every column's owner entry at that point: the column minimum).  Row i with exactly one column j1:
//   v[j1] = v0[j1] - min_{j != j1} (c[i][j] - v0[j])
// The cache was built against v0: it holds EVERY column with c - v0 < floor, so a cached minimum <= floor is the row's.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RTB) void wide_rt(const WideArgs *__restrict__ batch) {
    const WideArgs a = batch[blockIdx.y];
    const int n = a.n;
    if (n < 2) return;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (RTB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (RTB / 64);
    const float *__restrict__ v0 = a.cassign;
    long long done = 0;
    for (int i = gw; i < n; i += nw) {
        if (a.matches[i] != 1) continue;
        const int j1 = a.rowsol[i];
        const uint32_t col = a.cache_col[(int64_t)i * KC + lane];
        const float val = a.cache_val[(int64_t)i * KC + lane];
        const float tau = __shfl(val, KCU);
        const bool valid = lane < KCU && col != COLSENT && (int)col != j1;
        uint32_t key = 0xFFFFFFFFu;
        if (valid) key = f2ord(val - v0[col]);
        uint32_t mk = wave_min_u32(key);
        if (!(mk != 0xFFFFFFFFu && ord2f(mk) <= tau)) {       // the cache cannot certify the margin: the whole row
            const float *__restrict__ row = a.cost + wrow_off(a.rowmap, i, a.ld);
            uint32_t k2 = 0xFFFFFFFFu;
            for (int c = lane; c < n; c += 64)
                if (c != j1) k2 = umin32(k2, f2ord(row[c] - v0[c]));
            mk = wave_min_u32(k2);
        }
        if (lane == 0) a.v[j1] = v0[j1] - ord2f(mk);
        done++;
    }
    if (lane == 0 && done) atomicAdd(reinterpret_cast<unsigned long long *>(a.misc + 16) + C_RT, (unsigned long long)done);
}

// ------------------------------------------------------------------------------------------------------------------
// AUGMENTING ROW REDUCTION, Jacobi rounds (oracle/jv_oracle_impl.h, wide mode).  One workgroup per problem.
// ------------------------------------------------------------------------------------------------------------------
struct ArrShared {
    int cnt[2];
    int retired, dense;
    int wcnt[WNW];
    int sm_j[64], sm_i[64];
    float sm_p[64], sm_c[64];
};

__global__ __launch_bounds__(WT) void wide_arr(const WideArgs *__restrict__ batch) {
    __shared__ ArrShared s;
    const WideArgs a = batch[blockIdx.x];
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { s.cnt[0] = 0; s.cnt[1] = 0; s.retired = 0; s.dense = 0; }
    __syncthreads();
    // the active list: every free row (in any order -- a round does not depend on it)
    for (int i0 = 0; i0 < n; i0 += WT) {
        const int i = i0 + tid;
        const bool fr = i < n && a.rowsol[i] < 0;
        const uint64_t m = __ballot(fr);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s.cnt[0], __popcll(m));
        base = __shfl(base, 0);
        if (fr) a.act0[base + __popcll(m & lanemask_lt())] = i;
    }
    __syncthreads();
    const int free_cr = s.cnt[0];
    int cur = 0;
    long long round = 0, bids = 0;
    int32_t *A = a.act0, *B = a.act1;
    for (;;) {
        const int na = s.cnt[cur];
        if (na == 0 || round >= a.max_rounds) break;
        const bool small = na <= 64;
        // ---- bids: a wave per active row ----
        for (int slot = w; slot < na; slot += WNW) {
            const int i = ld_sc1(A + slot);
            const uint32_t col = a.cache_col[(int64_t)i * KC + lane];
            const float val = a.cache_val[(int64_t)i * KC + lane];
            const float tau = __shfl(val, KCU);
            const bool valid = lane < KCU && col != COLSENT;
            const float vj = valid ? ld_sc1(a.v + col) : 0.0f;
            K2 t; t.m1 = valid ? mkkey(val - vj, col) : KEYMAX; t.m2 = KEYMAX;
            const uint64_t mykey = t.m1;
            t = k2_wave_allreduce(t);
            float u1, u2 = INFINITY, c1, c2 = 0.0f, vj1, vj2 = 0.0f;
            int j1, j2 = -1;
            if (t.m2 != KEYMAX && key_val(t.m2) < tau) {          // the cached top-2 IS the row's lexicographic top-2
                const int l1 = __ffsll((unsigned long long)__ballot(mykey == t.m1)) - 1;
                const int l2 = __ffsll((unsigned long long)__ballot(mykey == t.m2)) - 1;
                u1 = key_val(t.m1); j1 = (int)(uint32_t)t.m1; c1 = __shfl(val, l1); vj1 = __shfl(vj, l1);
                u2 = key_val(t.m2); j2 = (int)(uint32_t)t.m2; c2 = __shfl(val, l2); vj2 = __shfl(vj, l2);
            } else {                                             // the whole row
                const float *__restrict__ row = a.cost + wrow_off(a.rowmap, i, a.ld);
                K2 d; d.m1 = KEYMAX; d.m2 = KEYMAX;
                for (int c = lane; c < n; c += 64) k2_push(d, mkkey(row[c] - ld_sc1(a.v + c), (uint32_t)c));
                d = k2_wave_allreduce(d);
                u1 = key_val(d.m1); j1 = (int)(uint32_t)d.m1; c1 = row[j1]; vj1 = ld_sc1(a.v + j1);
                if (d.m2 != KEYMAX) { u2 = key_val(d.m2); j2 = (int)(uint32_t)d.m2; c2 = row[j2]; vj2 = ld_sc1(a.v + j2); }
                if (lane == 0) atomicAdd(&s.dense, 1);
            }
            const float p = vj1 - (u2 - u1);
            int jt = -1;
            float pt = 0.0f, ct = 0.0f;
            if (p < vj1) { jt = j1; pt = p; ct = c1; }
            else if (ld_sc1(a.colsol + j1) < 0) { jt = j1; pt = vj1; ct = c1; }
            else if (j2 >= 0 && u2 == u1 && ld_sc1(a.colsol + j2) < 0) { jt = j2; pt = vj2; ct = c2; }
            if (lane == 0) {
                if (jt < 0) atomicAdd(&s.retired, 1);
                if (small) { s.sm_j[slot] = jt; s.sm_i[slot] = i; s.sm_p[slot] = pt; s.sm_c[slot] = ct; }
                else {
                    if (jt >= 0) atomicMin(a.bid + jt, (unsigned long long)mkkey(pt, (uint32_t)i));
                    a.slot_j[slot] = jt; a.slot_p[slot] = pt; a.slot_c[slot] = ct;
                }
            }
        }
        bids += na;
        __syncthreads();
        // ---- per column the lowest (price, row) wins; price, owner and displaced owner change together ----
        for (int slot = tid; slot < na; slot += WT) {
            int i, jt; float pt, ct;
            bool win;
            if (small) {
                i = s.sm_i[slot]; jt = s.sm_j[slot]; pt = s.sm_p[slot]; ct = s.sm_c[slot];
                win = jt >= 0;
                for (int k = 0; k < na && win; k++)
                    if (k != slot && s.sm_j[k] == jt && (s.sm_p[k] < pt || (s.sm_p[k] == pt && s.sm_i[k] < i))) win = false;
            } else {
                i = ld_sc1(A + slot); jt = ld_sc1(a.slot_j + slot); pt = ld_sc1(a.slot_p + slot); ct = ld_sc1(a.slot_c + slot);
                win = jt >= 0 && (uint32_t)ld_sc1(a.bid + jt) == (uint32_t)i;
            }
            if (jt < 0) continue;                                // retired: stays free, bids no more
            if (win) {
                const int i0 = ld_sc1(a.colsol + jt);
                a.v[jt] = pt; a.colsol[jt] = i; a.rowsol[i] = jt; a.cassign[jt] = ct;
                if (i0 >= 0) { a.rowsol[i0] = -1; B[atomicAdd(&s.cnt[cur ^ 1], 1)] = i0; }
            } else {
                B[atomicAdd(&s.cnt[cur ^ 1], 1)] = i;
            }
        }
        __syncthreads();
        if (!small)
            for (int slot = tid; slot < na; slot += WT) {
                const int jt = ld_sc1(a.slot_j + slot);
                if (jt >= 0) a.bid[jt] = ~0ull;
            }
        if (tid == 0) s.cnt[cur] = 0;
        cur ^= 1;
        { int32_t *t_ = A; A = B; B = t_; }
        round++;
        __syncthreads();
    }
    // ---- the rows still free, in ascending order, for the augmentation ----
    int numfree = 0;
    for (int i0 = 0; i0 < n; i0 += WT) {
        const int i = i0 + tid;
        const bool fr = i < n && ld_sc1(a.rowsol + i) < 0;
        const uint64_t m = __ballot(fr);
        if (lane == 0) s.wcnt[w] = __popcll(m);
        __syncthreads();
        int pre = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < WNW; k++) { const int c = s.wcnt[k]; if (k < w) pre += c; tot += c; }
        if (fr) a.freerows[numfree + pre + __popcll(m & lanemask_lt())] = i;
        numfree += tot;
        __syncthreads();
    }
    if (tid == 0) {
        long long *ctr = reinterpret_cast<long long *>(a.misc + 16);
        long long *wc = reinterpret_cast<long long *>(a.misc + 160);
        ctr[C_ARR] = bids; ctr[C_FREE_CR] = free_cr; ctr[C_FREE_A1] = numfree; ctr[C_FREE_A2] = numfree;
        wc[WC_ROUNDS] = round; wc[WC_BIDS] = bids; wc[WC_RETIRED] = s.retired; wc[WC_ACTIVE_LEFT] = s.cnt[cur];
        wc[WC_FREE_ARR] = numfree; wc[WC_DENSE_ARR] = s.dense;
        *reinterpret_cast<int *>(a.misc + 128) = numfree;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// AUGMENTATION (oracle/jv_oracle_impl.h, wide mode): succ-clamped shortest paths, run speculatively.
//
// Per column (global, L2): label = (ordered distance << 32 | predecessor row), all-ones = unlabelled; an atomic min on it
// keeps the smallest distance and, among equal distances, the lowest row -- the oracle's pred.
// In LDS: per 64-column block the smallest (distance, column) among its DIRTY columns (labelled, assigned, not settled at
// their current label), a dirty bit and an assigned bit per column, a dense bit per row (its cache could not certify the
// search: relaxed from the full cost row), the best unassigned column so far s_T = (distance, column).
// A round: every wave takes the best dirty column of the blocks it owns (block b belongs to wave b % 16), if its distance
// is below s_T's; clears the dirty bit, reads the label, relaxes the owner row's cached columns.  Barrier.  The block
// minimum of the block a wave took from is rebuilt.  Barrier.  No wave found work: converged.
// ------------------------------------------------------------------------------------------------------------------
size_t wide_aug_lds_bytes(int n) {
    const size_t nblk = ((size_t)n + 63) / 64, nw32 = ((size_t)n + 31) / 32;
    return ((nblk * 8 + 3 * nw32 * 4 + 15) / 16) * 16;
}

struct AugShared {
    unsigned long long T;          // best unassigned column: (ordered distance << 32 | column)
    int ntouch, any, fail, anydense, rootdense, doroot, f, err;
    int scans;
};

__global__ __launch_bounds__(WT) void wide_aug(const WideArgs *__restrict__ batch) {
    extern __shared__ __align__(16) unsigned char w_smem[];
    __shared__ AugShared s;
    const WideArgs a = batch[blockIdx.x];
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nblk = (n + 63) / 64, nw32 = (n + 31) / 32;
    unsigned long long *bmin = reinterpret_cast<unsigned long long *>(w_smem);
    uint32_t *dirty = reinterpret_cast<uint32_t *>(bmin + nblk);
    uint32_t *asg = dirty + nw32;
    uint32_t *dense = asg + nw32;
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    for (int b = tid; b < nblk; b += WT) bmin[b] = ~0ull;
    for (int q = tid; q < nw32; q += WT) { dirty[q] = 0; dense[q] = 0; }
    for (int c0 = 0; c0 < nw32 * 32; c0 += WT) {                 // assigned bits, 64 columns per wave and step
        const int c = c0 + tid;
        const uint64_t m = __ballot(c < n && a.colsol[c] >= 0);
        if (lane == 0 && (c >> 5) < nw32) { asg[c >> 5] = (uint32_t)m; if ((c >> 5) + 1 < nw32) asg[(c >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    if (tid == 0) { s.T = ~0ull; s.ntouch = 0; s.any = 0; s.fail = 0; s.anydense = 0; s.rootdense = 0; s.doroot = 0; s.f = 0; s.err = 0; s.scans = 0; }
    __syncthreads();

    long long c_relax = 0, c_hops = 0, c_rounds = 0, c_proc = 0, c_trivial = 0, c_dense = 0, c_verify = 0;   // (thread 0 / wave leaders)

    // one relaxation: column `col` is offered the distance `co` (ordered) by row `row`
    auto relax_to = [&](int col, uint32_t co, int row) {
        const unsigned long long key = ((unsigned long long)co << 32) | (uint32_t)row;
        const unsigned long long old = atomicMin(a.label + col, key);
        if (key < old) {
            if (old == ~0ull) a.touched[atomicAdd(&s.ntouch, 1)] = col;
            if ((uint32_t)(old >> 32) > co) {                    // the distance itself dropped (not only the row of a tie)
                const unsigned long long ck = ((unsigned long long)co << 32) | (uint32_t)col;
                if ((asg[col >> 5] >> (col & 31)) & 1u) { atomicOr(&dirty[col >> 5], 1u << (col & 31)); atomicMin(&bmin[col >> 6], ck); }
                else atomicMin(&s.T, ck);
            }
        }
    };

    int f = 0;
    for (;;) {
        // ---- wave 0 disposes of the searches that end at once: the free row's best cached column is unassigned and its
        // cache certifies that (no column settled, no price changes: the path is one edge) ----
        if (w == 0) {
            while (f < numfree) {
                const int fr = a.freerows[f];
                const uint32_t col = a.cache_col[(int64_t)fr * KC + lane];
                const float val = a.cache_val[(int64_t)fr * KC + lane];
                const float tau = __shfl(val, KCU);
                const bool valid = lane < KCU && col != COLSENT;
                const uint32_t od = valid ? f2ord(val - ld_sc1(a.v + col)) : 0xFFFFFFFFu;
                const uint32_t omin = wave_min_u32(od);
                const bool un = valid && od == omin && !((asg[col >> 5] >> (col & 31)) & 1u);
                const uint64_t mu = __ballot(un);
                if (!(omin != 0xFFFFFFFFu && mu && tau > ord2f(omin))) break;
                const int l = __ffsll((unsigned long long)mu) - 1;     // cache rows are sorted by column: the lowest such column
                if (lane == l) {
                    a.rowsol[fr] = (int)col; a.colsol[col] = fr; a.cassign[col] = val;
                    atomicOr(&asg[col >> 5], 1u << (col & 31));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                c_trivial++; c_hops++;
                f++;
            }
            if (lane == 0) s.f = f;
        }
        __syncthreads();
        f = s.f;
        if (f >= numfree) break;
        const int fr = a.freerows[f];
        const float *__restrict__ frow = a.cost + wrow_off(a.rowmap, fr, a.ld);
        const float ftau = a.cache_val[(int64_t)fr * KC + KCU];

        // ---- root: d[j] = c[fr][j] - v[j] for the cached columns (pred = fr) ----
        if (w == 0) {
            const uint32_t col = a.cache_col[(int64_t)fr * KC + lane];
            const float val = a.cache_val[(int64_t)fr * KC + lane];
            if (lane < KCU && col != COLSENT) relax_to((int)col, f2ord(val - ld_sc1(a.v + col)), fr);
        }
        __syncthreads();

        for (;;) {
            // ================= rounds until no wave finds work =================
            for (;;) {
                const uint32_t Tord = (uint32_t)(s.T >> 32);
                unsigned long long m = ~0ull;
                for (int b = w + WNW * lane; b < nblk; b += WNW * 64) m = umin64(m, bmin[b]);
                m = min64_wave_allreduce(m);
                const bool picked = m != ~0ull && (uint32_t)(m >> 32) < Tord;
                int pj = -1;
                if (picked) {
                    pj = (int)(uint32_t)m;
                    if (lane == 0) { atomicAnd(&dirty[pj >> 5], ~(1u << (pj & 31))); s.any = 1; }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the bit is cleared before the label is read
                    const unsigned long long lab = ld_sc1(a.label + pj);
                    const uint32_t dord = (uint32_t)(lab >> 32);
                    if (dord < Tord) {
                        const int i = ld_sc1(a.colsol + pj);
                        const float d = ord2f(dord);
                        const float h = (ld_sc1(a.cassign + pj) - ld_sc1(a.v + pj)) - d;
                        const uint32_t lo = dord + 1u;
                        if (!((dense[i >> 5] >> (i & 31)) & 1u)) {
                            const uint32_t col = a.cache_col[(int64_t)i * KC + lane];
                            const float val = a.cache_val[(int64_t)i * KC + lane];
                            if (lane < KCU && col != COLSENT && (int)col != pj) {
                                uint32_t co = f2ord((val - ld_sc1(a.v + col)) - h);
                                co = co < lo ? lo : co;
                                if (co <= Tord) relax_to((int)col, co, i);
                            }
                        } else {
                            const float *__restrict__ row = a.cost + wrow_off(a.rowmap, i, a.ld);
                            for (int c = lane; c < n; c += 64) {
                                if (c == pj) continue;
                                uint32_t co = f2ord((row[c] - ld_sc1(a.v + c)) - h);
                                co = co < lo ? lo : co;
                                if (co <= (uint32_t)(s.T >> 32)) relax_to(c, co, i);
                            }
                            c_dense++;
                        }
                        c_proc++;
                    }
                }
                __syncthreads();
                if (picked) {                                   // the block the wave took from: its smallest dirty column now
                    const int b = pj >> 6, c = b * 64 + lane;
                    const bool db = c < n && ((dirty[c >> 5] >> (c & 31)) & 1u);
                    unsigned long long key = ~0ull;
                    if (__ballot(db)) {
                        if (db) key = ((ld_sc1(a.label + c) >> 32) << 32) | (uint32_t)c;
                        key = min64_wave_allreduce(key);
                    }
                    if (lane == 0) bmin[b] = key;
                }
                const int any = s.any;
                c_rounds++;
                __syncthreads();
                if (tid == 0) s.any = 0;
                if (!any) break;
                // (s.any is re-armed by the picks of the next round only after every wave has read it: the write above
                //  and the next round's s.any = 1 are separated by the program order of wave 0 and the barrier below)
                __syncthreads();
            }
            // ================= converged: do the caches certify what was skipped? =================
            __syncthreads();
            const unsigned long long Tk = s.T;
            const uint32_t Dord = (uint32_t)(Tk >> 32);
            const float D = Tk == ~0ull ? INFINITY : ord2f(Dord);
            const int nt = s.ntouch;
            for (int q = tid; q < nt; q += WT) {
                const int k = ld_sc1(a.touched + q);
                const uint32_t dord = (uint32_t)(ld_sc1(a.label + k) >> 32);
                if (dord < Dord && ((asg[k >> 5] >> (k & 31)) & 1u)) {
                    const int i = ld_sc1(a.colsol + k);
                    if (!((dense[i >> 5] >> (i & 31)) & 1u)) {
                        const float h = (ld_sc1(a.cassign + k) - ld_sc1(a.v + k)) - ord2f(dord);
                        const float bound = a.cache_val[(int64_t)i * KC + KCU] - h;
                        if (!(bound > D)) {
                            atomicOr(&dense[i >> 5], 1u << (i & 31));
                            atomicOr(&dirty[k >> 5], 1u << (k & 31));
                            atomicMin(&bmin[k >> 6], ((unsigned long long)dord << 32) | (uint32_t)k);
                            atomicAdd(&s.fail, 1);
                        }
                    }
                }
            }
            if (tid == 0 && !s.rootdense && !(ftau > D)) { s.rootdense = 1; s.doroot = 1; atomicAdd(&s.fail, 1); }
            c_verify++;
            __syncthreads();
            const int fail = s.fail;
            if (s.doroot)
                for (int c = tid; c < n; c += WT) {
                    const uint32_t co = f2ord(frow[c] - ld_sc1(a.v + c));
                    if (co <= (uint32_t)(s.T >> 32)) relax_to(c, co, fr);
                }
            __syncthreads();
            if (tid == 0) { if (fail) s.anydense = 1; s.fail = 0; s.doroot = 0; }
            __syncthreads();
            if (!fail) break;
        }

        // ---- the search has ended at s.T: price update, path flip, reset ----
        const unsigned long long Tk = s.T;
        if (Tk == ~0ull) { if (tid == 0) s.err = 1; __syncthreads(); break; }
        const uint32_t Dord = (uint32_t)(Tk >> 32);
        const float D = ord2f(Dord);
        const int sink = (int)(uint32_t)Tk;
        const int nt = s.ntouch;
        int myscans = 0;
        for (int q = tid; q < nt; q += WT) {
            const int k = ld_sc1(a.touched + q);
            const uint32_t dord = (uint32_t)(ld_sc1(a.label + k) >> 32);
            if (dord < Dord && ((asg[k >> 5] >> (k & 31)) & 1u)) {
                const float vk = ld_sc1(a.v + k);
                const float nv = (vk + ord2f(dord)) - D;
                if (nv < vk) a.v[k] = nv;
                myscans++;
            }
        }
        if (myscans) atomicAdd(&s.scans, myscans);
        if (tid == 0) {
            int j = sink;
            for (;;) {
                const int i = (int)(uint32_t)ld_sc1(a.label + j);
                const int jn = ld_sc1(a.rowsol + i);
                a.colsol[j] = i; a.rowsol[i] = j; a.cassign[j] = a.cost[wrow_off(a.rowmap, i, a.ld) + j];
                c_hops++;
                if (i == fr) break;
                j = jn;
            }
            atomicOr(&asg[sink >> 5], 1u << (sink & 31));
        }
        __syncthreads();
        for (int q = tid; q < nt; q += WT) {
            const int k = ld_sc1(a.touched + q);
            a.label[k] = ~0ull;
            atomicAnd(&dirty[k >> 5], ~(1u << (k & 31)));
            bmin[k >> 6] = ~0ull;
        }
        if (s.anydense) for (int q = tid; q < nw32; q += WT) dense[q] = 0;
        __syncthreads();
        if (tid == 0) { c_relax += s.scans; s.scans = 0; s.T = ~0ull; s.ntouch = 0; s.anydense = 0; s.rootdense = 0; s.f = f + 1; }
        f++;
        __syncthreads();
    }

    // ---- duals, total, counters ----
    __syncthreads();
    double tot = 0.0;
    for (int i = tid; i < n; i += WT) {
        const int j = ld_sc1(a.rowsol + i);
        if (j >= 0) {
            const float cij = ld_sc1(a.cassign + j);
            a.u[i] = cij - ld_sc1(a.v + j);
            tot += (double)cij;
        }
    }
    __shared__ double s_tot[WNW];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if (lane == 0) s_tot[w] = tot;
    // per-wave counters of the wave leaders
    __shared__ long long s_wc[WNW][4];
    if (lane == 0) { s_wc[w][0] = c_proc; s_wc[w][1] = c_dense; s_wc[w][2] = c_trivial; s_wc[w][3] = (w == 0) ? c_hops : 0; }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        long long proc = 0, dn = 0, triv = 0, hops0 = 0;
        for (int k = 0; k < WNW; k++) { t += s_tot[k]; proc += s_wc[k][0]; dn += s_wc[k][1]; triv += s_wc[k][2]; hops0 += s_wc[k][3]; }
        *reinterpret_cast<double *>(a.misc + 8) = t;
        long long *ctr = reinterpret_cast<long long *>(a.misc + 16);
        long long *wc = reinterpret_cast<long long *>(a.misc + 160);
        ctr[C_AUG_INIT] = numfree; ctr[C_AUG_RELAX] = c_relax; ctr[C_AUGS] = numfree; ctr[C_HOPS] = hops0;
        wc[WC_DENSE_AUG] = dn; wc[WC_AUG_ROUNDS] = c_rounds; wc[WC_AUG_PROCESSED] = proc; wc[WC_TRIVIAL] = triv; wc[WC_VERIFY_PASSES] = c_verify;
        if (s.err) *reinterpret_cast<int *>(a.misc + 4) = 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
int wide_launch_rt(const WideArgs *d_args, int nb, int n, hipStream_t stream) {
    const int blocks = std::max(1, std::min((n + RTB / 64 - 1) / (RTB / 64), 2048 / std::max(1, std::min(nb, 8))));
    hipLaunchKernelGGL(wide_rt, dim3(blocks, nb), dim3(RTB), 0, stream, d_args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

int wide_launch_arr(const WideArgs *d_args, int nb, int n, hipStream_t stream) {
    (void)n;
    hipLaunchKernelGGL(wide_arr, dim3(nb), dim3(WT), 0, stream, d_args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

int wide_launch_aug(const WideArgs *d_args, int nb, int n, hipStream_t stream) {
    const size_t shm = wide_aug_lds_bytes(n);
    if (shm > (size_t)LDS_DYNAMIC_MAX) return CYTO_ERR_UNSUPPORTED;
    int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(wide_aug));
    if (rc) return rc;
    hipLaunchKernelGGL(wide_aug, dim3(nb), dim3(WT), shm, stream, d_args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

}  // namespace cyto
