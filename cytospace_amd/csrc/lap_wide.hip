// lap_wide.hip -- the WIDE form of the Jonker-Volgenant solve for gfx950 (MI355X), hand-written HIP.
//
// Same call it replaces as lap_jv.hip:  `_, y, _ = lapjv.lapjv(cost_scaled)`
// (/root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:34-40), same four phases.  The chain
// solver of lap_jv.hip follows the classic Gauss-Seidel order (one dependent row scan after the other: one wave of one
// CU busy); this file computes the order-free restatement of oracle/jv_oracle_impl.h ("WIDE MODE"), whose phases have
// width, bit for bit:
//
//   wide_rt    REDUCTION TRANSFER, Jacobi: every row that owns exactly one column takes its margin against the
//              post-column-reduction prices (snapshot), all margins are subtracted at once; and the scale of the instance -- the
//              histogram of the rows' gaps u2 - u1 -- for the eps schedule.  A wave per row, full chip.
//   wide_sc_*  AUGMENTING ROW REDUCTION as Jacobi rounds of an auction, eps-SCALED where the instance has generic costs: eight
//              eps = 0 rounds, then up to 16 phases eps_0 / 2^k (every row unassigned at a phase's start, prices kept, the phase's
//              sequential tail cut), then a final eps = 0 phase.  The phase machine: every round ONE launch over the whole chip (a wave
//              per bid of the round before: it resolves that bid and bids at once for the row that takes its place; prices and owners
//              of the columns just bid for are read from their bid words, the arrays run one round late), the state in a control
//              block per problem, so a batch's problems run through their phases independently in the same launches.  A round is a
//              pure function of the state.
//   wide_arr   the chain rounds (<= 64 active rows) of instances that did not scale -- one 16-wave workgroup per problem, the
//              rows in registers, bids meeting in LDS -- and every problem's list of free rows for the searches.
//   wide_claim_*  the searches that are ONE EDGE (CytoSPACE's repeated spot rows: a free slot takes the next free column tied at its
//              spot's minimum) on the whole chip, before the search kernel: the serial loop's assignments by deferred acceptance.
//   wide_aug   AUGMENTATION: per free row a shortest-path search whose labels are the unique fixed point of a monotone
//              system ((distance, tight-hop count) labels), so the search is run SPECULATIVELY: every round each of the 16 waves
//              settles the best unsettled columns of the column blocks it owns and relaxes their owner rows from the
//              row caches; a label that later improves is simply settled again.  Any schedule reaches the oracle's
//              Dijkstra labels.  Row caches certify the scans (floor > final distance, checked when the search has
//              converged; rows that fail are relaxed from their full cost row and the search continues).  PAR: the searches of
//              16 consecutive free rows of ONE problem at once, a workgroup each, the conflict-free prefix committed in row order.
//
// Row caches: lap_jv.hip (build_row_caches) -- <= 63 columns per row with their raw costs, sorted by column, and a floor
// that bounds the reduced cost of every other column for as long as prices only decrease (they do: the price update of
// this mode is clamped).
#include "lap_dev.h"
#include "lap_wide.h"
#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

namespace cyto {

namespace {

constexpr int WT = 1024;          // threads of the persistent workgroups: 16 waves
constexpr int WNW = WT / 64;
constexpr int RTB = 256;          // threads of the reduction-transfer workgroups

// counters shared with lap_jv.hip (misc + 16, long long each)
enum { C_RT = 0, C_ARR, C_AUG_INIT, C_AUG_RELAX, C_AUGS, C_HOPS, C_FREE_CR, C_FREE_A1, C_FREE_A2, C_ROWS_READ };

template <typename T> __device__ __forceinline__ T ld_sc1(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_sc1(T *p, T x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int64_t wrow_off(const int32_t *__restrict__ rowmap, int i, int64_t ld) {
    return (int64_t)(rowmap ? rowmap[i] : i) * ld;
}
#define WIDE_F_GPTR(T, name) __attribute__((address_space(1))) T *name;
#define WIDE_F_COPY_PTR(T, name) a.name = (T *)g.name;
#define WIDE_F_COPY_VAL(T, name) a.name = g.name;
struct WideArgsG { WIDE_FIELDS(WIDE_F_GPTR, WIDE_F_VAL) };
static_assert(sizeof(WideArgs) == sizeof(WideArgsG), "mirror layout");
__device__ __forceinline__ WideArgs load_wide_args(const WideArgs *__restrict__ batch, int b) {
    const WideArgsG g = reinterpret_cast<const WideArgsG *>(batch)[b];
    WideArgs a;
    WIDE_FIELDS(WIDE_F_COPY_PTR, WIDE_F_COPY_VAL)
    return a;
}
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
// a value every lane holds alike, moved to an SGPR: what depends on it becomes scalar code (scalar branches instead of
// exec-mask regions, scalar address arithmetic) -- the compiler cannot see that e.g. the wave index is uniform
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ float uni(float x) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); }
__device__ __forceinline__ unsigned long long uni(unsigned long long x) {
    return ((unsigned long long)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x);
}
// One full cost row by one wave: 16 bytes per lane and step, four steps' loads in flight (rows are 16-byte aligned and their
// pitch is a multiple of four elements: lap_jv.hip stages anything else).  f(column, cost) for every column < n.
template <typename F> __device__ __forceinline__ void wave_row_sweep(const float *__restrict__ row, int n, int lane, F &&f) {
    constexpr int U = 8;                                   // 8 KB of the row in flight per wave (12 measured slower: registers): a lone load per step is latency-bound
    const int nq = (n + 3) >> 2;
    const float4 *__restrict__ r4 = reinterpret_cast<const float4 *>(row);
    for (int q0 = lane; q0 < nq; q0 += 64 * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int q = q0 + 64 * u; x[u] = q < nq ? r4[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + 64 * u, c = q * 4;
            if (q >= nq) continue;
            f(c, x[u].x);
            if (c + 1 < n) f(c + 1, x[u].y);
            if (c + 2 < n) f(c + 2, x[u].z);
            if (c + 3 < n) f(c + 3, x[u].w);
        }
    }
}
// the quads [q_lo, q_hi) of a row by one wave (a slice: the whole workgroup shares a row)
template <typename F> __device__ __forceinline__ void wave_row_sweep_q(const float *__restrict__ row, int q_lo, int q_hi, int n, int lane, F &&f) {
    constexpr int U = 4;
    const float4 *__restrict__ r4 = reinterpret_cast<const float4 *>(row);
    for (int q0 = q_lo + lane; q0 < q_hi; q0 += 64 * U) {
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int q = q0 + 64 * u; x[u] = q < q_hi ? r4[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + 64 * u, c = q * 4;
            if (q >= q_hi) continue;
            f(c, x[u].x);
            if (c + 1 < n) f(c + 1, x[u].y);
            if (c + 2 < n) f(c + 2, x[u].z);
            if (c + 3 < n) f(c + 3, x[u].w);
        }
    }
}
// Search labels (oracle/jv_oracle_impl.h, WIDE MODE): a 44-bit label value LV = ordered distance (32) | tight-hop count k (12),
// then 20 bits of identity (the predecessor row in `label`, the column itself in the block minima and in the best unassigned
// column): one unsigned 64-bit compare orders (distance, k, identity) lexicographically; key >> 32 is the ordered distance.
constexpr uint32_t LKMAX = 4095u;
// Which of the 64 sorted lane minima of a row is tried first as the floor of its fresh cache.  The number of columns below the
// (k + 1)-th smallest lane minimum is the number of draws it takes to hit k + 1 of 64 lanes, less one (distribution-free when the
// values are distinct): k = 34 -> 49 +- 5 columns, more than 63 in 0.4 % of the rows; k = 47 (rounds 3-4) -> 86 +- 10: the first
// collecting sweep failed in 99.8 % of the rows and the second candidate (k = 23) left caches of 29 columns.
constexpr int SC_FLOOR_POS = 34;
__device__ __forceinline__ unsigned long long lkey(unsigned long long lv, uint32_t id) { return (lv << 20) | id; }
__device__ __forceinline__ unsigned long long lv_of(unsigned long long key) { return key >> 20; }
__device__ __forceinline__ uint32_t lid_of(unsigned long long key) { return (uint32_t)key & 0xFFFFFu; }
// the label value an edge gives: raw candidate (ordered) c_raw out of a column labelled (dord, k).  A candidate that does not
// exceed the label it comes from (a tight edge; rounding can even put it below) keeps the distance and counts one more tight hop
// -- nothing is added to the distance (LKMAX hops in a row: the next representable distance).
__device__ __forceinline__ unsigned long long edge_lv(uint32_t c_raw, uint32_t dord, uint32_t k) {
    return c_raw > dord ? ((unsigned long long)c_raw << 12)
                        : (k < LKMAX ? (((unsigned long long)dord << 12) | (k + 1u)) : ((unsigned long long)(dord + 1u) << 12));
}
// value of lane l (wave-uniform l) without the LDS round trip of __shfl
__device__ __forceinline__ uint32_t rdlane(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
__device__ __forceinline__ float rdlane(float x, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(x), l)); }

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// The control block of the row-reduction phase (WideArgs.sc, 2 KB, zeroed by the driver): the scale of the instance that
// wide_rt measures, and the state of the phase machine below (wide_sc_*).
// ------------------------------------------------------------------------------------------------------------------
enum { SC_LEGACY = 0, SC_EPS = 1, SC_FINAL = 2, SC_HANDOVER = 3, SC_DONE = 4 };   // modes (>= SC_HANDOVER: the machine is through)
enum { SC_ACT_NONE = 0, SC_ACT_ROUND = 1, SC_ACT_RESET = 2 };                      // what a launch does
// `fresh`: the list launch L reads was made by wide_sc_init (rows that have not bid yet); else its rows bid in launch L - 1
struct ScSlot { int mode, k, rip, fresh, act; float eps; long long total, bids; int heavy, pad_; };   // heavy: the instance bids from full rows a lot (as of the launch before: the same for every workgroup)
struct ScRound { int cnt, retired; };   // per launch (three cells in rotation: L % 3): rows that bid in it, how many of them retired
struct ScCtl {
    int hist[256];                 // rows per binary exponent (the exponent FIELD) of their gap u2 - u1 at the post-column-reduction prices
    uint32_t vmaxbits;             // bits of max_j |v0[j]|
    int e0;                        // exponent field of eps_0 (0: the instance never scales)
    float epsmin;                  // phases end below this eps (the resolution of the prices)
    int stop;                      // WIDE_STOP(n)
    ScRound rnd[3];
    int retired, dense, dense_mark, phases, free_cr;
    int fin_buf, fin_cnt;          // where the machine stopped: the record buffer and the length of the list it leaves
    int want_mark, pad2_;          // the launch at which fresh row caches were last asked for
    unsigned long long wbase[2];   // per bid-word buffer: the launch at which it was last wiped (the words' 12-bit tag is relative to it)
    ScSlot slot[2];                // launch L reads slot[L & 1] and leaves slot[(L + 1) & 1]
};
static_assert(sizeof(ScCtl) <= 2048, "control block of the row-reduction phase");
// the constants of the restatement (oracle/jv_oracle.h: JV_WIDE_*)
constexpr int SC_K0 = 8, SC_NPH = 16, SC_PHCAP = 1024, SC_EMULT = 3, SC_ESTEP = 1;
constexpr int SC_STOP_FINAL = 16;       // the final eps = 0 phase ends at min(wide_stop(n), 16) active rows (JV_WIDE_STOP_FINAL)
constexpr int SC_UNROLL = 4;           // quads of a full-row sweep in flight per lane (2: 67 registers, 7 waves per SIMD; 3: 79, 6; 4: 107, 4 -- and 4 is the fastest: gpurun_out/r05h)
#ifndef HEAVY_SPREAD
#define HEAVY_SPREAD 1
#endif
constexpr int SC_SMALL = 2048;         // launches with at most so many bids to resolve give every bid a wave of its own (a matter of speed only)
constexpr int SC_COARSE = 4;           // phases whose full-row bids leave the row caches alone (a matter of speed only)
#ifndef WIDE_STOP_CAP
#define WIDE_STOP_CAP 64
#endif
__host__ __device__ inline int wide_stop(int n) { return n / 128 < 8 ? 8 : (n / 128 > WIDE_STOP_CAP ? WIDE_STOP_CAP : n / 128); }
// the next representable value below x (+0 and -0 are one value): oracle pred_
__device__ __forceinline__ float pred_f32(float x) {
    const uint32_t o = f2ord(x);
    float r = ord2f(o - 1u);
    if (!(r < x)) r = ord2f(o - 2u);
    return r;
}

// ------------------------------------------------------------------------------------------------------------------
// REDUCTION TRANSFER (Jacobi) -- and the scale of the instance.  v0 = the post-column-reduction prices (a copy lives in
// a.cassign, which also is the raw cost of every column's owner entry at that point: the column minimum).
// Every row takes the lexicographic top-2 (u1, j1), (u2, j2) of c[i][.] - v0[.]: its gap u2 - u1 goes into the histogram of
// binary exponents (integer counts: no order); a row that owns exactly one column jo subtracts its margin
//   v[jo] = v0[jo] - min_{j != jo} (c[i][j] - v0[j])        (= u2 if j1 == jo, else u1)
// The cache was built against v0: it holds EVERY column with c - v0 < floor, so a cached second minimum < floor is the row's.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RTB) void wide_rt(const WideArgs *__restrict__ batch) {
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    const int n = a.n;
    if (n < 2) return;
    __shared__ int s_hist[256];
    __shared__ uint32_t s_vmax;
    for (int e = threadIdx.x; e < 256; e += RTB) s_hist[e] = 0;
    if (threadIdx.x == 0) s_vmax = 0;
    __syncthreads();
    ScCtl *sc = reinterpret_cast<ScCtl *>(a.sc);
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (RTB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (RTB / 64);
    const float *__restrict__ v0 = a.cassign;
    {
        uint32_t m = 0;
        for (int j = blockIdx.x * RTB + threadIdx.x; j < n; j += gridDim.x * RTB) { const uint32_t x = __float_as_uint(fabsf(v0[j])); m = m > x ? m : x; }
        if (m) atomicMax(&s_vmax, m);
    }
    long long done = 0;
    for (int i = gw; i < n; i += nw) {
        const uint32_t col = a.cache_col[(int64_t)i * KC + lane];
        const float val = a.cache_val[(int64_t)i * KC + lane];
        const float tau = __shfl(val, KCU);
        const bool valid = lane < KCU && col != COLSENT;
        const uint32_t key = valid ? f2ord(val - v0[col]) : 0xFFFFFFFFu;
        const uint32_t k1 = wave_min_u32(key);
        const int l1 = __ffsll((unsigned long long)__ballot(key == k1)) - 1;
        const uint32_t k2 = wave_min_u32(lane == l1 ? 0xFFFFFFFFu : key);
        float u1, u2; int j1;
        if (k2 != 0xFFFFFFFFu && ord2f(k2) < tau) { u1 = ord2f(k1); u2 = ord2f(k2); j1 = (int)rdlane(col, l1); }
        else {                                                  // the cache cannot certify the top-2: the whole row
            const float *__restrict__ row = a.cost + wrow_off(a.rowmap, i, a.ld);
            K2 d; d.m1 = KEYMAX; d.m2 = KEYMAX;
            wave_row_sweep(row, n, lane, [&](int c, float x) { k2_push(d, mkkey(x - v0[c], (uint32_t)c)); });
            d = k2_wave_allreduce(d);
            u1 = key_val(d.m1); u2 = key_val(d.m2); j1 = (int)(uint32_t)d.m1;
        }
        if (lane == 0) atomicAdd(&s_hist[(__float_as_uint(u2 - u1) >> 23) & 0xFFu], 1);
        if (a.matches[i] == 1) {
            const int jo = a.rowsol[i];
            if (lane == 0) a.v[jo] = v0[jo] - (j1 == jo ? u2 : u1);
            done++;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 256; e += RTB) if (s_hist[e]) atomicAdd(&sc->hist[e], s_hist[e]);
    if (threadIdx.x == 0 && s_vmax) atomicMax(&sc->vmaxbits, s_vmax);
    if (lane == 0 && done) atomicAdd(reinterpret_cast<unsigned long long *>(a.misc + 16) + C_RT, (unsigned long long)done);
}

// ------------------------------------------------------------------------------------------------------------------
// AUGMENTING ROW REDUCTION, Jacobi rounds (oracle/jv_oracle_impl.h, wide mode).  One workgroup per problem.
//
// VLDS: the prices (f32) and the column owners (u16, 0xFFFF = unassigned) live in LDS beside their global copies (every
// update goes to both), so a bid is: the row's cache (two coalesced 256-B loads), 63 LDS gathers, two DPP reductions, two
// LDS look-ups.  Otherwise (n > ~26 000) both stay in L2 and are read with agent-scope loads.
//
// Two regimes.  LIST rounds (more than 64 active rows: the first few hundred rounds): the active rows are a list in
// global memory, a wave takes four of them at a time (row ids, then the four caches, requested together); the bids of a
// round meet in the per-column 64-bit atomic-min words `bid`.  CHAIN rounds (<= 64 active rows -- the long tail of the
// price wars, thousands of rounds): every wave holds up to four rows IN REGISTERS; the row a slot works on in the next round
// is the owner it displaces, known BEFORE the round's barrier, so that row's cache is requested at once and has arrived when
// the next round starts; bids meet in LDS (a ballot finds a better bid on the same column), two barriers per round and no
// global round trip on the path.  When half of the rows have dropped out the rest is dealt out again, one round-robin over
// the waves.  Both regimes realise the same round (a pure function of the state).
// ------------------------------------------------------------------------------------------------------------------
constexpr int ACS = 4;                 // rows per wave in a chain round / requested together in a list round
constexpr int ASL = WNW * ACS;         // slots of a chain round

struct ArrShared {
    int cnt[2];
    int retired, dense, ndeal;
    int pause;                          // the list rounds return to the driver for fresh row caches
    int wcnt[WNW];
    int sm_j[ASL], sm_i[ASL];
    float sm_p[ASL];
    int deal[ASL];
    int rf_cnt[WNW];                    // cache refresh (a wave rebuilds the cache of the row it just read in full): staging
    uint32_t rf_col[WNW][KC];
    float rf_val[WNW][KC];
};

constexpr size_t ARR_SHARED_BYTES = (sizeof(ArrShared) + 15) / 16 * 16;     // (in the dynamic region: a kernel's static LDS is limited to 2 KB here)
size_t wide_arr_lds_bytes(int n, bool vlds, bool clds) {
    const size_t npad = ((size_t)n + 3) & ~(size_t)3;
    return ARR_SHARED_BYTES + ((npad * ((vlds ? 4 : 0) + (clds ? 2 : 0)) + 15) / 16) * 16;
}
bool wide_arr_vlds(int n) { return n <= 65534 && wide_arr_lds_bytes(n, true, true) + 1024 <= (size_t)LDS_DYNAMIC_MAX; }
bool wide_arr_clds(int n) { return n <= 65534 && wide_arr_lds_bytes(n, false, true) + 1024 <= (size_t)LDS_DYNAMIC_MAX; }

template <bool VLDS, bool CLDS> struct ArrCtx {
    WideArgs a;
    float *s_v; uint16_t *s_cs; ArrShared *s;
    int lane;
    __device__ __forceinline__ float getv(int j) const { return VLDS ? s_v[j] : ld_sc1(a.v + j); }
    __device__ __forceinline__ int getcs(int j) const {
        if (CLDS) { const uint16_t x = s_cs[j]; return x == 0xFFFFu ? -1 : (int)x; }
        return ld_sc1(a.colsol + j);
    }
    // a won bid: price, owner, displaced owner change together (LDS copies and global)
    __device__ __forceinline__ void apply(int i, int jt, float pt, float ct, int i0) const {
        if (VLDS) s_v[jt] = pt;
        if (CLDS) s_cs[jt] = (uint16_t)i;
        a.v[jt] = pt; a.colsol[jt] = i; a.rowsol[i] = jt; a.cassign[jt] = ct;
        if (i0 >= 0) a.rowsol[i0] = -1;
    }
    // exact lexicographic top-2 of the whole row (the cache could not certify) -- and a fresh cache for the row against the
    // current prices, so that its next bids are certified again (in a price war the same few rows bid thousands of times).
    // The new floor is one of the 64 per-lane minima of the sweep (sorted; the 35th, else the 17th, 8th ... smallest: SC_FLOOR_POS): every
    // column below it is collected in a second sweep of the now L2-resident row; more than 63 of them -> the next candidate.
    // (Inlined on purpose: as an out-of-line call its results travel through scratch memory, and every bid -- also the
    // certified ones -- then stores and reloads them through the vector memory path.)
    __device__ __forceinline__ void top2_full(int i, int w, uint32_t &col, float &val, float &u1, int &j1, float &c1, float &vj1,
                                              float &u2, int &j2, float &c2, float &vj2) const {
        const int n = a.n;
        const float *__restrict__ row = a.cost + wrow_off(a.rowmap, i, a.ld);
        K2 d; d.m1 = KEYMAX; d.m2 = KEYMAX;
        wave_row_sweep(row, n, lane, [&](int c, float x) { k2_push(d, mkkey(x - getv(c), (uint32_t)c)); });
        uint32_t lm = (uint32_t)(d.m1 >> 32);                     // this lane's smallest reduced cost (ordered)
        d = k2_wave_allreduce(d);
        u1 = key_val(d.m1); j1 = (int)(uint32_t)d.m1; c1 = uni(row[j1]); vj1 = uni(getv(j1));
        u2 = INFINITY; j2 = -1; c2 = 0.0f; vj2 = 0.0f;
        if (d.m2 != KEYMAX) { u2 = key_val(d.m2); j2 = (int)(uint32_t)d.m2; c2 = uni(row[j2]); vj2 = uni(getv(j2)); }
        if (lane == 0) atomicAdd(&s->dense, 1);
        // ---- the row's new cache ----
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {                       // bitonic sort of the 64 lane minima, ascending over the lanes
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)lm, j);
                const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
                lm = take_min ? umin32(lm, o) : (lm > o ? lm : o);
            }
        }
        uint32_t tk = 0;                                           // ordered floor; 0 = none found
        int cnt = 0;
        for (int pos = SC_FLOOR_POS; pos >= 2 && !tk; pos = (pos + 1) / 2 - 1) {
            const uint32_t cand = rdlane(lm, pos);
            if (cand == 0xFFFFFFFFu) continue;
            if (lane == 0) s->rf_cnt[w] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wave_row_sweep(row, n, lane, [&](int c, float x) {
                if (f2ord(x - getv(c)) < cand) {
                    const int p = atomicAdd(&s->rf_cnt[w], 1);
                    if (p < KCU) { s->rf_col[w][p] = (uint32_t)c; s->rf_val[w][p] = x; }
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cnt = uni(s->rf_cnt[w]);
            if (cnt <= KCU) tk = cand;
        }
        const float tau = tk ? ord2f(tk) : -INFINITY;
        uint32_t kc = (tk && lane < cnt) ? s->rf_col[w][lane] : COLSENT;
        float kv = (tk && lane < cnt) ? s->rf_val[w][lane] : 0.0f;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {                       // cache rows are kept sorted by column (unused slots last)
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
        a.cache_col[(int64_t)i * KC + lane] = kc;
        a.cache_val[(int64_t)i * KC + lane] = kv;
        col = kc; val = kv;
    }
    // the bid of row i (its cache row in col / val, lane = entry): target column jt (-1: the row retires), price, raw cost of
    // the entry, and the owner it would displace
    // eps > 0 (a scaled phase): every bid lowers its column's price by the gap + eps, by one ulp at least -- no claims, nobody retires
    __device__ __forceinline__ void bid_of(int i, int w, uint32_t &col, float &val, int &jt, float &pt, float &ct, int &i0, float eps = 0.0f) const {
        // cache rows are sorted by column: among equal reduced costs the lowest lane is the lowest column, so the
        // lexicographic (value, column) top-2 is two 32-bit min reductions and two ballots
        const float tau = rdlane(val, KCU);
        const bool valid = lane < KCU && col != COLSENT;
        const float vj = valid ? getv((int)col) : 0.0f;
        const uint32_t key = valid ? f2ord(val - vj) : 0xFFFFFFFFu;
        const uint32_t k1 = wave_min_u32(key);
        const int l1 = __ffsll((unsigned long long)__ballot(key == k1)) - 1;
        const uint32_t key2 = lane == l1 ? 0xFFFFFFFFu : key;
        const uint32_t k2 = wave_min_u32(key2);
        float u1, u2, c1, c2, vj1, vj2;
        int j1, j2;
        if (k2 != 0xFFFFFFFFu && ord2f(k2) < tau) {           // the cached top-2 IS the row's lexicographic top-2
            const int l2 = __ffsll((unsigned long long)__ballot(key2 == k2)) - 1;
            u1 = ord2f(k1); j1 = (int)rdlane(col, l1); c1 = rdlane(val, l1); vj1 = rdlane(vj, l1);
            u2 = ord2f(k2); j2 = (int)rdlane(col, l2); c2 = rdlane(val, l2); vj2 = rdlane(vj, l2);
        } else {
            top2_full(i, w, col, val, u1, j1, c1, vj1, u2, j2, c2, vj2);
        }
        jt = -1; pt = 0.0f; ct = 0.0f; i0 = -1;
        const int o1 = uni(getcs(j1));
        if (eps > 0.0f) {
            float p = vj1 - ((u2 - u1) + eps);
            if (!(p < vj1)) p = pred_f32(vj1);
            jt = j1; pt = p; ct = c1; i0 = o1;
            jt = uni(jt); pt = uni(pt); ct = uni(ct); i0 = uni(i0);
            return;
        }
        const float p = vj1 - (u2 - u1);
        if (p < vj1) { jt = j1; pt = p; ct = c1; i0 = o1; }
        else if (o1 < 0) { jt = j1; pt = vj1; ct = c1; }
        else if (j2 >= 0 && u2 == u1 && uni(getcs(j2)) < 0) { jt = j2; pt = vj2; ct = c2; }
        jt = uni(jt); pt = uni(pt); ct = uni(ct); i0 = uni(i0);
    }
};

// The rounds on the whole chip: the phase machine of the row reduction (oracle/jv_oracle_impl.h, WIDE MODE).
//   LEGACY   eps = 0 rounds from the column reduction's state (claims, retirements).  After SC_K0 of them an instance whose active
//            list is still longer than wide_stop(n) -- generic costs: no ties to retire on, price wars without end -- SCALES; any
//            other one goes on until its list is short (<= 64 rows: the chain rounds of wide_arr take over) or empty.
//   EPS      phase k: every row unassigned, prices kept; rounds with eps_k = eps_0 / 2^k (every bid lowers its column's price by the
//            gap + eps_k) until the list is down to wide_stop(n) rows -- the sequential tail of a phase is cut, the next phase
//            takes every row up again.  eps_0 = 2^SC_EMULT x the median binade of the rows' gaps at the post-column-reduction prices.
//   FINAL    the same with eps = 0 and the claim / retire rules: what it leaves free goes to the searches.
// A round is ONE launch over all CUs (wide_sc_round).  Launch L does, a wave per row that bid in launch L - 1: the RESOLUTION of that
// row's bid -- did it win its column (the column's bid word)?  then price, owner, displaced owner are written to the arrays -- and at
// once the BID of the row that takes its place in the next round (the row itself if it lost, the owner it displaced if it won): no
// list is built in between, the wave that knows the outcome makes the next bid.  So the arrays (v, colsol) run ONE ROUND LATE while a
// launch is under way -- a column that received a bid in launch L - 1 has its new price and owner in its bid word (the winner's), and
// that is where the bids of launch L read them (word of launch L - 1 -> fresh; else the arrays, complete through launch L - 2);
// bid words are double-buffered by the launch's parity (launch L reads buffer (L - 1) & 1 and merges into buffer L & 1).  Full-row
// bids read the arrays and, through a bitmap of the columns bid for in launch L - 1 (three bitmaps in rotation), the fresh words.
// The bids of launch L are SPECULATIVE: whether the round they belong to takes place at all (or the phase ended with the list
// launch L - 1 left: its length is known only when that launch is over) is what launch L + 1 decides, from the same pure function of
// (state, list length) as the oracle -- if not, the bids are dropped: a bid changes nothing but bid words, records and counters kept
// per launch.  A phase boundary is a launch too: everything unassigned, and every row bids (nothing to resolve, no owners).
// The driver enqueues launches in groups and learns after each group who is through, one group late (the next group is queued
// before it asks: the chip never waits for the host; launches of a machine that is through return at once).
// The bid word of a column: | 12 bits ~(launch - base) | 32 bits ordered price | 20 bits row |, merged with an atomic min: within a
// launch the lowest (price, row) wins, and ANY bid of a later launch beats what earlier ones left behind -- the words are never reset
// between rounds; wide_sc_wipe (every 2048 launches per buffer) resets them and moves the buffer's `base`.
constexpr int HEADB = 256;             // threads of the machine's workgroups
// ---- the machine's bids.  A row whose cache certifies its top-2 is one wave's work (the chain of a bid is L2 round trips: what
// counts is how many bids are in flight, so a wave per bid while the chip has room).  A row whose cache cannot certify is read in
// full -- by the WHOLE workgroup, after the wave's certified bids of the iteration: one wave sweeping an 80 KB row three times
// (wide_arr's top2_full) is 170 us, and the round waits for its slowest bid.  Prices and owners do not change while the bids of
// a round are made (the resolution is the next launch): plain loads, 16 bytes of the row and of the prices per lane and step.
struct ScShared {
    int nq, cnt, fill_, obase;
    uint32_t cand;
    int lrow[HEADB];                   // rows that bid out of the tile being resolved
    int qrow[HEADB / 64], qslot[HEADB / 64];
    unsigned long long km1[HEADB / 64], km2[HEADB / 64];
    uint32_t lmin[HEADB];
    uint32_t ccol[KC];
    float cval[KC];
};
constexpr size_t SC_SHARED_BYTES = (sizeof(ScShared) + 15) / 16 * 16;

// Where a launch's bids read prices and owners: the bid words of the launch before (fresh: that column's winner) or the arrays.
struct ScView {
    const unsigned long long *wsrc;   // bid words of launch L - 1 (null: none -- the first launch, a phase boundary)
    const uint32_t *pr;               // the prices as of the end of launch L - 1 (ordered; the machine's own price arrays); null: a.v holds them
    uint32_t tg;                      // tag of launch L - 1 in those words
    bool own_none;                    // a phase boundary: every column is unassigned (the arrays are being cleared by this very launch)
};
__device__ __forceinline__ bool sc_word_fresh(const ScView &vw, unsigned long long w) {
    return (uint32_t)(w >> 52) == vw.tg && ((uint32_t)w & 0xFFFFFu) != 0xFFFFFu;
}
__device__ __forceinline__ float sc_word_price(unsigned long long w) { return ord2f((uint32_t)(w >> 20)); }
// price and owner of column c as of the end of launch L - 1 (a dependent word load only where there is a word to ask)
__device__ __forceinline__ void sc_price_owner(const WideArgs &a, const ScView &vw, int c, float &p, int &o) {
    p = vw.pr ? ord2f(vw.pr[c]) : a.v[c]; o = vw.own_none ? -1 : a.colsol[c];
    if (vw.wsrc) { const unsigned long long w = vw.wsrc[c]; if (sc_word_fresh(vw, w)) o = (int)((uint32_t)w & 0xFFFFFu); }
}

// the decision of a bid from its row's lexicographic top-2 (oracle: JV_WIDE_ROUND): target column (-1: the row retires), price, raw cost
// of the entry, the owner it would displace (o1 / o2: the owners of the two columns).  eps > 0 (a scaled phase): every bid lowers its
// column's price by the gap + eps, by one ulp at least -- no claims, nobody retires
struct Top2 { float u1, c1, vj1, u2, c2, vj2; int j1, j2, o1, o2; };
__device__ __forceinline__ void sc_decide(const Top2 &t, float eps, int &jt, float &pt, float &ct, int &i0) {
    jt = -1; pt = 0.0f; ct = 0.0f; i0 = -1;
    const int o1 = t.o1;
    if (eps > 0.0f) {
        float p = t.vj1 - ((t.u2 - t.u1) + eps);
        if (!(p < t.vj1)) p = pred_f32(t.vj1);
        jt = t.j1; pt = p; ct = t.c1; i0 = o1;
    } else {
        const float p = t.vj1 - (t.u2 - t.u1);
        if (p < t.vj1) { jt = t.j1; pt = p; ct = t.c1; i0 = o1; }
        else if (o1 < 0) { jt = t.j1; pt = t.vj1; ct = t.c1; }
        else if (t.j2 >= 0 && t.u2 == t.u1 && t.o2 < 0) { jt = t.j2; pt = t.vj2; ct = t.c2; }
    }
    jt = uni(jt); pt = uni(pt); ct = uni(ct); i0 = uni(i0);
}
// one wave: the top-2 from the row's cache (lane = entry); false: the cache cannot certify it.  The prices of the 63 cached columns
// are gathered from the machine's price array (complete through the launch before); the owners of the two columns that matter are
// read last -- a column's bid word of the launch before if it is fresh (its winner), else the owner array.
// GOWN: words and owners of all cached columns are gathered with the prices (a launch with few bids is a chain of round trips: one less;
// for a launch with many bids that tripled the lines a bid pulls in: 8 000 bids took 60 us)
template <bool GOWN = false>
__device__ __forceinline__ bool sc_top2_cached(const WideArgs &a, const ScView &vw, int lane, uint32_t col, float val, Top2 &t) {
    const float tau = rdlane(val, KCU);
    const bool valid = lane < KCU && col != COLSENT;
    const float vj = valid ? (vw.pr ? ord2f(vw.pr[col]) : a.v[col]) : 0.0f;
    int ow = -1;
    if (GOWN && !vw.own_none) {
        ow = valid ? a.colsol[col] : -1;
        if (vw.wsrc) { const unsigned long long w = valid ? vw.wsrc[col] : ~0ull; if (sc_word_fresh(vw, w)) ow = (int)((uint32_t)w & 0xFFFFFu); }
    }
    const uint32_t key = valid ? f2ord(val - vj) : 0xFFFFFFFFu;
    const uint32_t k1 = wave_min_u32(key);
    const int l1 = __ffsll((unsigned long long)__ballot(key == k1)) - 1;
    const uint32_t key2 = lane == l1 ? 0xFFFFFFFFu : key;
    const uint32_t k2 = wave_min_u32(key2);
    if (!(k2 != 0xFFFFFFFFu && ord2f(k2) < tau)) return false;
    const int l2 = __ffsll((unsigned long long)__ballot(key2 == k2)) - 1;
    t.u1 = ord2f(k1); t.j1 = (int)rdlane(col, l1); t.c1 = rdlane(val, l1); t.vj1 = rdlane(vj, l1); t.o1 = (int)rdlane((uint32_t)ow, l1);
    t.u2 = ord2f(k2); t.j2 = (int)rdlane(col, l2); t.c2 = rdlane(val, l2); t.vj2 = rdlane(vj, l2); t.o2 = (int)rdlane((uint32_t)ow, l2);
    if (vw.own_none) { t.o1 = -1; t.o2 = -1; }
    else if (!GOWN) {
        const int c1 = a.colsol[t.j1], c2 = a.colsol[t.j2];        // (all four requested together)
        unsigned long long w1 = ~0ull, w2 = ~0ull;
        if (vw.wsrc) { w1 = vw.wsrc[t.j1]; w2 = vw.wsrc[t.j2]; }
        t.o1 = sc_word_fresh(vw, uni(w1)) ? (int)((uint32_t)uni(w1) & 0xFFFFFu) : uni(c1);
        t.o2 = sc_word_fresh(vw, uni(w2)) ? (int)((uint32_t)uni(w2) & 0xFFFFFu) : uni(c2);
    }
    return true;
}
// the whole workgroup (HEADB threads): f(column, cost, price) for every column of the row, 16 bytes of row and prices per lane and step
// (the prices: the machine's ordered price array, or a.v at a phase boundary)
template <int U, typename F> __device__ __forceinline__ void block_row_sweep(const float *__restrict__ row, const float *__restrict__ v, const ScView &vw, int n, F &&f) {
    const int nq = (n + 3) >> 2;
    const float4 *__restrict__ r4 = reinterpret_cast<const float4 *>(row);
    const float4 *__restrict__ v4 = reinterpret_cast<const float4 *>(vw.pr ? reinterpret_cast<const float *>(vw.pr) : v);   // (16-byte aligned, whole quads stay in range)
    const bool ordered = vw.pr != nullptr;
    for (int q0 = threadIdx.x; q0 < nq; q0 += HEADB * U) {
        float4 x[U], p[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + HEADB * u;
            x[u] = q < nq ? r4[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            p[u] = q < nq ? v4[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + HEADB * u, c = q * 4;
            if (q >= nq) continue;
            if (ordered) { p[u].x = ord2f(__float_as_uint(p[u].x)); p[u].y = ord2f(__float_as_uint(p[u].y)); p[u].z = ord2f(__float_as_uint(p[u].z)); p[u].w = ord2f(__float_as_uint(p[u].w)); }
            f(c, x[u].x, p[u].x);
            if (c + 1 < n) f(c + 1, x[u].y, p[u].y);
            if (c + 2 < n) f(c + 2, x[u].z, p[u].z);
            if (c + 3 < n) f(c + 3, x[u].w, p[u].w);
        }
    }
}
// the whole workgroup: exact lexicographic top-2 of row i -- and a fresh cache for it against the current prices (floor = one of the 64
// minima of the columns c with (c / 4) % 64 == l, sorted: the 35th, else the 17th, 8th ... smallest; the columns below it are collected in
// a second sweep of the now L2-resident row; more than 63 of them -> the next candidate).  Every thread returns the same Top2.
template <int U>
__device__ __forceinline__ Top2 sc_top2_block(const WideArgs &a, const ScView &vw, ScShared &ss, int i, bool rebuild) {
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *__restrict__ row = a.cost + wrow_off(a.rowmap, i, a.ld);
    K2 d; d.m1 = KEYMAX; d.m2 = KEYMAX;
    block_row_sweep<U>(row, a.v, vw, n, [&](int c, float x, float vc) { k2_push(d, mkkey(x - vc, (uint32_t)c)); });
    const uint32_t my_min = (uint32_t)(d.m1 >> 32);              // the smallest reduced cost among THIS thread's columns (ordered)
    ss.lmin[tid] = my_min;
    d = k2_wave_allreduce(d);
    if (lane == 0) { ss.km1[w] = d.m1; ss.km2[w] = d.m2; }
    __syncthreads();
    K2 g; g.m1 = ss.km1[0]; g.m2 = ss.km2[0];
#pragma unroll
    for (int k = 1; k < HEADB / 64; k++) { K2 o; o.m1 = ss.km1[k]; o.m2 = ss.km2[k]; k2_merge(g, o); }
    Top2 t;
    t.u1 = key_val(g.m1); t.j1 = (int)(uint32_t)g.m1; t.c1 = row[t.j1]; sc_price_owner(a, vw, t.j1, t.vj1, t.o1);
    t.u2 = INFINITY; t.j2 = -1; t.c2 = 0.0f; t.vj2 = 0.0f; t.o2 = -1;
    if (g.m2 != KEYMAX) { t.u2 = key_val(g.m2); t.j2 = (int)(uint32_t)g.m2; t.c2 = row[t.j2]; sc_price_owner(a, vw, t.j2, t.vj2, t.o2); }
    if (!rebuild) { __syncthreads(); return t; }                   // (the staging words are free again for the next row)
    // ---- the row's new cache ----
    uint32_t lm = 0xFFFFFFFFu;
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < HEADB / 64; k++) lm = umin32(lm, ss.lmin[k * 64 + lane]);
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {                       // bitonic sort of the 64 minima, ascending over the lanes
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)lm, j);
                const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
                lm = take_min ? umin32(lm, o) : (lm > o ? lm : o);
            }
        }
    }
    uint32_t tk = 0;                                               // ordered floor; 0 = none found
    int cnt = 0;
    for (int pos = SC_FLOOR_POS; pos >= 2 && !tk; pos = (pos + 1) / 2 - 1) {
        if (w == 0 && lane == 0) { ss.cand = rdlane(lm, pos); ss.cnt = 0; }
        __syncthreads();
        const uint32_t cand = ss.cand;
        if (cand != 0xFFFFFFFFu && my_min < cand)                    // (a thread none of whose columns lies below the candidate reads nothing: ~4 of 5)
            block_row_sweep<U>(row, a.v, vw, n, [&](int c, float x, float vc) {
                if (f2ord(x - vc) < cand) {
                    const int p = atomicAdd(&ss.cnt, 1);
                    if (p < KCU) { ss.ccol[p] = (uint32_t)c; ss.cval[p] = x; }
                }
            });
        __syncthreads();
        cnt = ss.cnt;
        if (cand != 0xFFFFFFFFu && cnt <= KCU) tk = cand;
        __syncthreads();                                           // (everybody has read the count before the next candidate resets it)
    }
    if (w == 0) {
        const float tau = tk ? ord2f(tk) : -INFINITY;
        uint32_t kc = (tk && lane < cnt) ? ss.ccol[lane] : COLSENT;
        float kv = (tk && lane < cnt) ? ss.cval[lane] : 0.0f;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {                       // cache rows are kept sorted by column (unused slots last)
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
        a.cache_col[(int64_t)i * KC + lane] = kc;
        a.cache_val[(int64_t)i * KC + lane] = kv;
    }
    return t;
}

__device__ __forceinline__ unsigned long long bidkey(long long round, float price, int row) {
    return ((unsigned long long)(~(uint32_t)round & 0xFFFu) << 52) | ((unsigned long long)f2ord(price) << 20) | (uint32_t)row;
}
__device__ __forceinline__ bool bid_won(unsigned long long word, int row) { return (int)((uint32_t)word & 0xFFFFFu) == row; }

struct ArrHead { int cnt[2]; int started, free_cr; long long round, bids; int retired, dense; int done, launches; long long list_rounds; int no_more, pad_; };

// ---- the machine's memory beyond the driver's arrays (WideArgs.scx, wide_sc_ext_bytes(n) bytes, 256-byte aligned): the two buffers of
// bid words (all-ones at the start: the first wide_sc_ones_bytes(n) bytes), the machine's two price arrays (ordered floats; wide_sc_init
// copies the prices in), the bid records of a launch (two buffers by the launch's parity: 16 bytes { row, column or -1, price,
// owner it would displace } and the raw cost of the entry).
struct ScRec { int i, jt; float pt; int i0; };
// (accessors instead of a table of pointers: a table indexed by the launch's parity would live in scratch memory)
__host__ __device__ inline size_t sc_np(int n) { return ((size_t)n + 63) & ~(size_t)63; }
__device__ __forceinline__ unsigned long long *sc_words(char *scx, int n, int b) { return reinterpret_cast<unsigned long long *>(scx + sc_np(n) * 8 * (size_t)b); }
__device__ __forceinline__ uint32_t *sc_price(char *scx, int n, int k) { return reinterpret_cast<uint32_t *>(scx + sc_np(n) * (16 + 4 * (size_t)k)); }
__device__ __forceinline__ ScRec *sc_recs(char *scx, int n, int b) { return reinterpret_cast<ScRec *>(scx + sc_np(n) * (24 + 16 * (size_t)b)); }
__device__ __forceinline__ float *sc_rcts(char *scx, int n, int b) { return reinterpret_cast<float *>(scx + sc_np(n) * (56 + 4 * (size_t)b)); }

__device__ __forceinline__ float sc_eps_of(const ScCtl *sc, int k) {       // eps of phase k, 0 = there is no such phase
    if (k >= SC_NPH) return 0.0f;
    const int ek = sc->e0 - SC_ESTEP * k;
    if (ek < 1) return 0.0f;
    const float eps = __uint_as_float((uint32_t)ek << 23);
    return eps < sc->epsmin ? 0.0f : eps;
}
// the step the machine takes from state S with `na` active rows (oracle: the head of the phase machine's loop; every thread of a
// launch computes the same)
__device__ __forceinline__ ScSlot sc_step(const ScCtl *sc, const ScSlot &S, int na, int n, long long max_rounds) {
    ScSlot N = S;
    N.act = SC_ACT_NONE; N.fresh = 0;
    if (S.mode >= SC_HANDOVER) return N;
    // (the budget of rounds ends the LEGACY rounds and the scaled phases, never the FINAL phase: the assignments a scaled phase leaves
    //  satisfy eps-complementary slackness only, the searches need the eps = 0 phase's)
    const bool over = S.total >= max_rounds;
    bool next_phase = false, last = false;
    if (S.mode == SC_LEGACY) {
        if (over || na == 0) { N.mode = SC_DONE; return N; }
        if (S.rip == SC_K0 && na > sc->stop && sc->e0 > 0) { next_phase = true; N.k = -1; }
        else if (S.rip >= SC_K0 && na <= ASL) { N.mode = SC_HANDOVER; return N; }
    } else if (S.mode == SC_EPS) {
        if (over) { next_phase = true; last = true; }
        else if (S.rip >= 1 && (na <= sc->stop || S.rip >= SC_PHCAP)) next_phase = true;
    } else if (S.rip >= 1 && (na <= (sc->stop < SC_STOP_FINAL ? sc->stop : SC_STOP_FINAL) || S.rip >= SC_PHCAP)) { N.mode = SC_DONE; return N; }
    if (next_phase) {
        N.k = N.k + 1;
        const float eps = last ? 0.0f : sc_eps_of(sc, N.k);
        N.mode = eps > 0.0f ? SC_EPS : SC_FINAL; N.eps = eps; N.rip = 0; N.act = SC_ACT_RESET;
        return N;
    }
    N.act = SC_ACT_ROUND; N.rip = S.rip + 1; N.total = S.total + 1; N.bids = S.bids + na;
    return N;
}

__global__ __launch_bounds__(HEADB) void wide_sc_init(const WideArgs *__restrict__ batch) {
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    ScCtl *sc = reinterpret_cast<ScCtl *>(a.sc);
    const int n = a.n, lane = threadIdx.x & 63;
    for (int i0 = blockIdx.x * HEADB; i0 < n; i0 += gridDim.x * HEADB) {
        const int i = i0 + threadIdx.x;
        const bool fr = i < n && a.rowsol[i] < 0;
        const uint64_t m = __ballot(fr);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&sc->rnd[2].cnt, __popcll(m));        // (the list launch 0 reads: cell (0 - 1) mod 3)
        base = __shfl(base, 0);
        if (fr) a.act0[base + __popcll(m & lanemask_lt())] = i;
        if (i < n) { const uint32_t o = f2ord(a.v[i]); sc_price((char *)a.scx, n, 0)[i] = o; sc_price((char *)a.scx, n, 1)[i] = o; }     // the machine's price arrays
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // the median binade of the gaps (wide_rt's histogram), eps_0 = 2^SC_EMULT times it; no scaling when half the gaps are zero
        long long cum = 0; int me = 0;
        for (int e = 0; e < 256; e++) { cum += sc->hist[e]; if (cum * 2 >= n) { me = e; break; } }
        sc->e0 = me > 0 ? (me + SC_EMULT > 254 ? 254 : me + SC_EMULT) : 0;
        sc->epsmin = __uint_as_float(sc->vmaxbits) * 1.1920928955078125e-07f;
        sc->stop = wide_stop(n);
        ScSlot S; S.mode = SC_LEGACY; S.k = 0; S.rip = 0; S.fresh = 1; S.act = SC_ACT_NONE; S.eps = 0.0f; S.total = 0; S.bids = 0; S.heavy = 0; S.pad_ = 0;
        sc->slot[0] = S;
    }
}

// Launch L of the machine (see the header above): the step the state and the length of the list that bid in launch L - 1 ask for --
//   ROUND  those bids stand: a wave per bid resolves it and bids at once for the row that takes its place;
//   RESET  a phase begins (those bids are dropped): everything unassigned, every row bids;
//   none   the machine is through (those bids are dropped; their rows are the list it leaves) --
// or, in launch 0, the first bids of the rows the column reduction left free.
template <int U, bool DIRECT>
__global__ __launch_bounds__(HEADB) void wide_sc_round(const WideArgs *__restrict__ batch, int L, int n_arg, char *sc_direct, char *scx_direct, int *hs, int small_max) {
    extern __shared__ __align__(16) unsigned char w_smem[];
    // DIRECT (one problem): the control block, the machine's arrays and n are kernel arguments, so the state and -- unconditionally, a
    // wave per slot -- the record a launch with few bids will resolve are requested before the argument block has arrived
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    const int n = DIRECT ? n_arg : a.n;
    ScCtl *sc = reinterpret_cast<ScCtl *>(DIRECT ? sc_direct : (char *)a.sc);
    char *scx = DIRECT ? scx_direct : (char *)a.scx;
    const int lane = threadIdx.x & 63, w = uni((int)(threadIdx.x >> 6));
    const int pb = (L + 1) & 1, cb = L & 1;                      // record / word buffers: the launch before, this launch
    const int rp = (L + 2) % 3, rc = L % 3, rn = (L + 1) % 3;    // per-launch cells and bitmaps: the launch before, this one, the next
    const ScRec *rsrc = sc_recs(scx, n, pb);
    const float *csrc = sc_rcts(scx, n, pb);
    const int slot_s = (int)blockIdx.x * (HEADB / 64) + w;
    int4 rr_s = make_int4(-1, -1, 0, -1);
    float rct_s = 0.0f;
    if (slot_s < n) { rr_s = *reinterpret_cast<const int4 *>(rsrc + slot_s); rct_s = csrc[slot_s]; }
    const ScSlot S = sc->slot[L & 1];
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    // (what the driver watches, pinned host memory: the launch under way -- reported by the batch's first problem whatever its state)
    if (lead && hs && blockIdx.y == 0) __hip_atomic_store(hs, L + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (S.mode >= SC_HANDOVER) {                                 // through: the state stays (both slots)
        if (lead && sc->slot[(L + 1) & 1].mode != S.mode) { ScSlot N = S; N.act = SC_ACT_NONE; sc->slot[(L + 1) & 1] = N; }
        return;
    }
    const int np = sc->rnd[rp].cnt;                              // rows that bid in launch L - 1 (launch 0: rows on wide_sc_init's list)
    const bool first = S.fresh != 0;
    ScSlot N;
    if (first) { N = S; N.fresh = 0; N.act = SC_ACT_NONE; } else N = sc_step(sc, S, np, n, a.max_rounds);
    const int act = N.act;
    const bool through = !first && act == SC_ACT_NONE;
    if (lead) {
        // what the driver watches besides the launch under way (pinned host memory, no synchronisation): the machines that are through, and
        // per problem "rebuild my row caches" -- the full-row bids since the last rebuild have reached a.arr_waste
        // (measured on the few-cell-type 20 000^2 instance: a rebuild whenever a.arr_waste full-row bids have been made -- ~14
        //  rebuilds of 1 ms -- gives a 25 ms row reduction; rebuilding only from n / 2 full-row bids per group on and never in the
        //  coarse phases 34 ms: a round waits for the workgroup with the most full-row bids, so few of them already cost every round)
        if (hs) {
            if (through) __hip_atomic_fetch_add(hs + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else if (a.aug_seg == 0 && L - sc->want_mark >= 64 && sc->dense - sc->dense_mark >= a.arr_waste) {      // (once per 64 launches at most)
                sc->dense_mark = sc->dense; sc->want_mark = L;
                __hip_atomic_store(hs + 4 + blockIdx.y, L + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        N.heavy = (long long)sc->dense * 32 > N.bids ? 1 : 0;
        sc->slot[(L + 1) & 1] = N;
        sc->rnd[rn].cnt = 0; sc->rnd[rn].retired = 0;
        if (first) { sc->free_cr = np; sc->rnd[rc].cnt = np; }
        if (act == SC_ACT_ROUND) sc->retired += sc->rnd[rp].retired;   // the round took place: its retirements count
        if (act == SC_ACT_RESET) { sc->phases += 1; sc->rnd[rc].cnt = n; }
        if (through) { sc->fin_buf = pb; sc->fin_cnt = np; }
    }
    if (through) return;
    // The machine's two price arrays (ordered floats, merged with atomic mins: prices only fall): launch L reads P[(L - 1) & 1] -- complete
    // through round L - 1 -- and merges into P[L & 1] (complete through round L - 2 when the launch starts) the bids it resolves (round L - 1)
    // and the bids it makes: complete through round L when it ends.  Dropped bids (a phase ended) have spoilt the array they were merged
    // into, P[(L - 1) & 1] at the phase boundary L: that launch reads the price array proper (a.v: complete, the resolutions wrote it)
    // and copies it over the spoilt one.
    uint32_t *pw = sc_price(scx, n, cb);
    if (act == SC_ACT_RESET) {
        uint32_t *pfix = sc_price(scx, n, pb);
        for (int i = blockIdx.x * HEADB + threadIdx.x; i < n; i += gridDim.x * HEADB) { a.rowsol[i] = -1; a.colsol[i] = -1; pfix[i] = f2ord(a.v[i]); }
    }
    // few bids to resolve: a wave per bid (below); else tiles
    // (an instance that bids from full rows a lot -- S.heavy -- takes the tiles even for few bids when it has the chip to itself (a batch's
    //  problems have 64-256 workgroups each: measured slower there): ONE bid per workgroup while there are workgroups enough: a full-row bid is the whole workgroup's work, and the launch waits for the workgroup with the most of them)
    const bool small = act == SC_ACT_ROUND && !(S.heavy && HEAVY_SPREAD && gridDim.x >= 512) && np <= small_max && np <= (int)gridDim.x * (HEADB / 64);
    if (small && (int)blockIdx.x * (HEADB / 64) >= np) return;
    ScShared &ss = *reinterpret_cast<ScShared *>(w_smem);
    if (threadIdx.x == 0) { ss.nq = 0; ss.fill_ = 0; ss.cnt = 0; }
    __syncthreads();
    ScView vw;
    vw.wsrc = act == SC_ACT_ROUND ? sc_words(scx, n, pb) : nullptr;
    vw.pr = act == SC_ACT_RESET ? nullptr : sc_price(scx, n, pb);
    vw.tg = ~(uint32_t)((long long)(L - 1) - (long long)sc->wbase[pb]) & 0xFFFu;
    vw.own_none = act == SC_ACT_RESET;
    unsigned long long *wdst = sc_words(scx, n, cb);
    ScRec *rdst = sc_recs(scx, n, cb);
    float *cdst = sc_rcts(scx, n, cb);
    const long long tag = (long long)L - (long long)sc->wbase[cb];
    const float eps = N.eps;
    // a coarse phase (eps_k many times the span of a row's 63 cached columns) moves the prices past every cache within a bid or two:
    // there a full-row bid does not rebuild its row's cache (half its cost) -- the later phases do, and keep their caches
    const bool refresh = !(N.mode == SC_EPS && N.k < SC_COARSE);
    int retired = 0, dense = 0;                                   // (lane 0 of every wave)
    int *ocnt = &sc->rnd[rc].cnt;
    auto record = [&](int oslot, int i, const Top2 &t) {
        int jt, i0; float pt, ct;
        sc_decide(t, eps, jt, pt, ct, i0);
        if (lane == 0) {
            if (jt < 0) retired++;
            else {
                atomicMin(wdst + jt, bidkey(tag, pt, i));
                atomicMin(pw + jt, f2ord(pt));
            }
            *reinterpret_cast<int4 *>(rdst + oslot) = make_int4(i, jt, __float_as_int(pt), i0);
            cdst[oslot] = ct;
        }
    };
    // rows whose caches could not certify (queued in LDS): the whole workgroup reads each in full, one after the other
    auto full_rows = [&](int obase) {
        __syncthreads();
        const int nq = ss.nq;
        __syncthreads();
        if (nq) {
            if (threadIdx.x == 0) ss.nq = 0;
            for (int q = 0; q < nq; q++) {
                const int i = ss.qrow[q], oslot = obase + ss.qslot[q];
                const Top2 t = sc_top2_block<U>(a, vw, ss, i, refresh);
                if (w == 0) { record(oslot, i, t); dense++; }
            }
            __syncthreads();
        }
    };
    if (small) {
        // ---- a wave per bid of the launch before.  Everything the outcome decides between is requested with the word that decides
        // it: the caches of the row (it bids again if it lost) and of the owner it displaces (who bids if it won); the slot of the
        // new record comes from one atomic per workgroup that is in flight while the bid is made.
        const int ri = uni(rr_s.x), rj = uni(rr_s.y), r0 = uni(rr_s.w);
        int nxt = -1, rank = 0;
        uint32_t col = COLSENT; float val = 0.0f;
        if (slot_s < np && rj >= 0) {                               // (a retired row stays free and bids no more)
            if (lane == 0) atomicMin(pw + rj, f2ord(__int_as_float(rr_s.z)));     // (every bid of the round, won or lost: the lowest is the price)
            const unsigned long long word = vw.wsrc[rj];
            const uint32_t colA = a.cache_col[(int64_t)ri * KC + lane];
            const float valA = a.cache_val[(int64_t)ri * KC + lane];
            uint32_t colB = COLSENT; float valB = 0.0f;
            if (r0 >= 0) { colB = a.cache_col[(int64_t)r0 * KC + lane]; valB = a.cache_val[(int64_t)r0 * KC + lane]; }
            if (bid_won(uni(word), ri)) {
                if (lane == 0) {
                    a.v[rj] = __int_as_float(rr_s.z); a.colsol[rj] = ri; a.rowsol[ri] = rj; a.cassign[rj] = rct_s;
                    if (r0 >= 0) a.rowsol[r0] = -1;
                }
                nxt = r0; col = colB; val = valB;
            } else { nxt = ri; col = colA; val = valA; }
            if (nxt >= 0) { if (lane == 0) rank = atomicAdd(&ss.cnt, 1); rank = uni(rank); }
        }
        __syncthreads();
        int obase_r = 0;
        const int nl = ss.cnt;
        if (threadIdx.x == 0 && nl) obase_r = atomicAdd(ocnt, nl);
        Top2 t;
        bool have = false;
        if (nxt >= 0) {
            have = sc_top2_cached<true>(a, vw, lane, col, val, t);
            if (!have && lane == 0) { const int q = atomicAdd(&ss.nq, 1); ss.qrow[q] = nxt; ss.qslot[q] = rank; }
        }
        if (threadIdx.x == 0) ss.obase = obase_r;
        __syncthreads();
        const int obase = ss.obase;
        if (have) record(obase + rank, nxt, t);
        full_rows(obase);
    } else {
        // ---- every workgroup takes a contiguous run of the work (bids to resolve; rows at a phase boundary) in tiles of HEADB.
        // ROUND, per tile: first a THREAD per bid resolves it (record and word: coalesced / one gather; a winner's thread writes price,
        // owner, displaced owner) and the rows that bid next are gathered in LDS -- their records' slots come from ONE atomic on the
        // launch's counter per tile (a wave per bid with an atomic each: 20 000 atomics on one address made a phase's first launches
        // 130-170 us) -- then a WAVE per gathered row makes its bid.
        const int nwork = act == SC_ACT_RESET ? n : np;
        // (a run of at least 8 bids -- fewer atomics on the launch's counter -- unless the instance bids from full rows a lot: those a
        //  workgroup reads one after the other, and the launch waits for the workgroup with the most of them)
        const bool heavy = S.heavy != 0;
        const int chunk = std::max(act == SC_ACT_ROUND && !heavy ? 8 : (heavy && HEAVY_SPREAD && gridDim.x >= 512 && act == SC_ACT_ROUND ? 1 : HEADB / 64), (nwork + (int)gridDim.x - 1) / (int)gridDim.x);
        const int c_lo = std::min(nwork, (int)blockIdx.x * chunk), c_hi = std::min(nwork, c_lo + chunk);
        for (int t0 = c_lo; t0 < c_hi; t0 += HEADB) {
            const int tn = std::min(HEADB, c_hi - t0);           // work items of this tile
            int nlist = tn;                                       // rows that bid out of this tile
            if (act == SC_ACT_ROUND) {
                if (threadIdx.x == 0) ss.cnt = 0;
                __syncthreads();
                int nxt = -1;
                if ((int)threadIdx.x < tn) {
                    const int slot = t0 + (int)threadIdx.x;
                    const int4 rr = *reinterpret_cast<const int4 *>(rsrc + slot);
                    if (rr.y >= 0) {                                // (a retired row stays free and bids no more)
                        atomicMin(pw + rr.y, f2ord(__int_as_float(rr.z)));      // (every bid of the round, won or lost: the lowest is the price)
                        const unsigned long long word = vw.wsrc[rr.y];
                        const float rct = csrc[slot];
                        if (bid_won(word, rr.x)) {
                            a.v[rr.y] = __int_as_float(rr.z); a.colsol[rr.y] = rr.x; a.rowsol[rr.x] = rr.y; a.cassign[rr.y] = rct;
                            if (rr.w >= 0) a.rowsol[rr.w] = -1;
                            nxt = rr.w;                             // the displaced owner bids next (or nobody)
                        } else nxt = rr.x;
                    }
                }
                const uint64_t mb = __ballot(nxt >= 0);
                int wbase = 0;
                if (lane == 0 && mb) wbase = atomicAdd(&ss.cnt, __popcll(mb));
                wbase = __shfl(wbase, 0);
                const int lpos = wbase + __popcll(mb & lanemask_lt());
                if (nxt >= 0) ss.lrow[lpos] = nxt;
                __syncthreads();
                nlist = ss.cnt;
                if (threadIdx.x == 0 && nlist) ss.obase = atomicAdd(ocnt, nlist);
                __syncthreads();
            }
            const int obase = act == SC_ACT_ROUND ? ss.obase : t0;
            for (int e0 = 0; e0 < nlist; e0 += HEADB / 64) {     // (the same trips for every wave of the workgroup)
                const int e = e0 + w;
                if (e < nlist) {
                    const int i = act == SC_ACT_ROUND ? uni(ss.lrow[e]) : (act == SC_ACT_RESET ? t0 + e : uni(a.act0[t0 + e]));
                    const uint32_t col = a.cache_col[(int64_t)i * KC + lane];
                    const float val = a.cache_val[(int64_t)i * KC + lane];
                    Top2 t;
                    if (sc_top2_cached(a, vw, lane, col, val, t)) record(obase + e, i, t);
                    else if (lane == 0) { const int q = atomicAdd(&ss.nq, 1); ss.qrow[q] = i; ss.qslot[q] = e; }
                }
                full_rows(obase);
            }
        }
    }
    if (lane == 0) { if (retired) atomicAdd(&sc->rnd[rc].retired, retired); if (dense) atomicAdd(&sc->dense, dense); }
}

// before launch L: the bid words of ITS buffer start over (the other buffer holds the bids launch L resolves)
__global__ __launch_bounds__(HEADB) void wide_sc_wipe(const WideArgs *__restrict__ batch, int L) {
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    ScCtl *sc = reinterpret_cast<ScCtl *>(a.sc);
    unsigned long long *wd = sc_words((char *)a.scx, a.n, L & 1);
    for (int j = blockIdx.x * HEADB + threadIdx.x; j < a.n; j += gridDim.x * HEADB) wd[j] = ~0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->wbase[L & 1] = (unsigned long long)L;
}

// the machine is through (the next launch would be L): what wide_arr picks up -- the rows that were active when it stopped (for the
// chain rounds of a LEGACY problem: on the list its round's parity names), the counters; no_more: the rounds are over (budget, or a
// scaled problem's last phase)
__global__ void wide_sc_finish(const WideArgs *__restrict__ batch, int L) {
    const WideArgs a = load_wide_args(batch, blockIdx.x);
    ScCtl *sc = reinterpret_cast<ScCtl *>(a.sc);
    ArrHead *h = reinterpret_cast<ArrHead *>(a.misc + 384);
    const ScSlot S = sc->slot[L & 1];
    const int na = sc->fin_cnt, want = (int)(S.total & 1);
    const bool chain = S.mode == SC_HANDOVER;
    if (chain) {
        const ScRec *R = sc_recs((char *)a.scx, a.n, sc->fin_buf);
        int32_t *B = want ? a.act1 : a.act0;
        for (int q = threadIdx.x; q < na; q += blockDim.x) B[q] = R[q].i;
    }
    if (threadIdx.x == 0) {
        h->cnt[want] = na; h->cnt[want ^ 1] = 0; h->started = 1; h->free_cr = sc->free_cr; h->round = S.total; h->bids = S.bids;
        h->retired = sc->retired; h->dense = sc->dense; h->done = 0; h->launches = 0; h->list_rounds = S.total; h->no_more = chain ? 0 : 1;
        long long *dbg = reinterpret_cast<long long *>(a.misc + 256);
        dbg[13] = sc->phases > 0 ? 1 : 0; dbg[14] = sc->phases;
    }
}

template <bool VLDS, bool CLDS>
__global__ __launch_bounds__(WT) void wide_arr(const WideArgs *__restrict__ batch) {
    extern __shared__ __align__(16) unsigned char w_smem[];
    ArrShared &s = *reinterpret_cast<ArrShared *>(w_smem);
    ArrCtx<VLDS, CLDS> cx;
    cx.a = load_wide_args(batch, blockIdx.x);
    const WideArgs &a = cx.a;
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, w = uni((int)(threadIdx.x >> 6));
    cx.s_v = reinterpret_cast<float *>(w_smem + ARR_SHARED_BYTES);
    cx.s_cs = reinterpret_cast<uint16_t *>(cx.s_v + (VLDS ? ((n + 3) & ~3) : 0));
    cx.s = &s; cx.lane = lane;
    const long long t_kernel0 = wall_clock64();
    // The rounds may take several launches (the control block at misc + 384 carries the state): like the searches of wide_aug, the
    // list rounds return to the driver when the row caches have gone stale -- a row whose cache cannot certify its bid reads its
    // full row and rebuilds its own cache, three sweeps on one wave; once those have cost what a rebuild of ALL caches by the
    // whole chip costs (a.arr_waste of them in this launch; or a.seg_quorum workgroups of the launch have asked), that is cheaper.
    ArrHead *h = reinterpret_cast<ArrHead *>(a.misc + 384);
    if (h->done) {                                               // finished in an earlier launch
        if (tid == 0 && a.seg_sync) a.seg_sync[1 + blockIdx.x] = a.seg_sync[1 + blockIdx.x] & 2;
        return;
    }
    if (tid == 0) { s.cnt[0] = 0; s.cnt[1] = 0; s.retired = 0; s.dense = 0; s.ndeal = 0; s.pause = 0; }
    if (CLDS)
        for (int j = tid; j < n; j += WT) {
            if (VLDS) cx.s_v[j] = a.v[j];
            const int o = a.colsol[j]; cx.s_cs[j] = o < 0 ? (uint16_t)0xFFFFu : (uint16_t)o;
        }
    __syncthreads();

    // the active list: what the phase machine on the whole chip left (wide_sc_finish), else every free row (in any order -- a
    // round does not depend on it)
    const bool headed = h->started != 0;
    int cur = 0;
    long long round = 0, bids = 0;
    if (headed) {
        round = h->round; bids = h->bids; cur = (int)(round & 1);
        if (tid == 0) { s.cnt[cur] = h->cnt[cur]; s.retired = h->retired; s.dense = h->dense; }
    } else {
        for (int i0 = 0; i0 < n; i0 += WT) {
            const int i = i0 + tid;
            const bool fr = i < n && a.rowsol[i] < 0;
            const uint64_t m = __ballot(fr);
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&s.cnt[0], __popcll(m));
            base = __shfl(base, 0);
            if (fr) a.act0[base + __popcll(m & lanemask_lt())] = i;
        }
    }
    __syncthreads();
    const int free_cr = headed ? h->free_cr : s.cnt[0];
    const long long t_start = wall_clock64();
    if (tid == 0) reinterpret_cast<long long *>(a.misc + 256)[5] = t_start - t_kernel0;
    long long t_list = 0, t_chain = 0, n_list = 0, n_chain = 0, n_deal = 0;
    int32_t *A = cur ? a.act1 : a.act0, *B = cur ? a.act0 : a.act1;
    int na = free_cr;
    const int dense0 = s.dense;
    const long long round0 = round;
    const bool no_more = h->no_more != 0;
    bool paused = false, announced = false;
    // ================= LIST rounds =================
    for (;;) {
        na = uni(s.cnt[cur]);
        if (na <= ASL || round >= a.max_rounds || no_more) break;
        if (uni(s.pause)) { paused = true; break; }
        if (round == round0 || (round & 0xFFF) == 0) {               // the bid words: wiped at the start and whenever their round tag wraps
            for (int j = tid; j < n; j += WT) a.bid[j] = ~0ull;
            __syncthreads();
        }
        for (int base = 0; base < na; base += ASL) {
            int ri[ACS]; uint32_t col[ACS]; float val[ACS];
#pragma unroll
            for (int q = 0; q < ACS; q++) { const int slot = base + q * WNW + w; ri[q] = slot < na ? uni(ld_sc1(A + slot)) : -1; }
#pragma unroll
            for (int q = 0; q < ACS; q++)
                if (ri[q] >= 0) { col[q] = a.cache_col[(int64_t)ri[q] * KC + lane]; val[q] = a.cache_val[(int64_t)ri[q] * KC + lane]; }
#pragma unroll
            for (int q = 0; q < ACS; q++) {
                if (ri[q] < 0) continue;
                const int slot = base + q * WNW + w;
                int jt, i0; float pt, ct;
                cx.bid_of(ri[q], w, col[q], val[q], jt, pt, ct, i0);
                if (lane == 0) {
                    if (jt < 0) atomicAdd(&s.retired, 1);
                    else atomicMin(a.bid + jt, bidkey(round, pt, ri[q]));
                    a.slot_j[slot] = jt; a.slot_p[slot] = pt; a.slot_c[slot] = ct;
                }
            }
        }
        bids += na;
        __syncthreads();
        for (int slot = tid; slot < na; slot += WT) {
            const int jt = ld_sc1(a.slot_j + slot);
            if (jt < 0) continue;                                // retired: stays free, bids no more
            const int i = ld_sc1(A + slot);
            if (bid_won(ld_sc1(a.bid + jt), i)) {
                const int i0 = cx.getcs(jt);
                cx.apply(i, jt, ld_sc1(a.slot_p + slot), ld_sc1(a.slot_c + slot), i0);
                if (i0 >= 0) B[atomicAdd(&s.cnt[cur ^ 1], 1)] = i0;
            } else {
                B[atomicAdd(&s.cnt[cur ^ 1], 1)] = i;
            }
        }
        if (tid == 0) {
            s.cnt[cur] = 0;
            if (a.aug_seg == 0 && a.seg_sync && round > round0) {     // (s.dense: complete since the barrier behind the bids)
                if (s.dense - dense0 >= a.arr_waste) { s.pause = 1; if (!announced) atomicAdd(a.seg_sync, 1); announced = true; }
                else if (a.seg_quorum > 0 && ld_sc1(a.seg_sync) >= a.seg_quorum) s.pause = 1;
            }
        }
        cur ^= 1;
        { int32_t *t_ = A; A = B; B = t_; }
        round++;
        __syncthreads();
    }
    t_list = wall_clock64() - t_start;
    n_list = (headed && h->launches > 0) ? h->list_rounds + (round - round0) : round;      // (the machine's rounds count as list rounds)
    const long long t_chain0 = wall_clock64();
    // ================= CHAIN rounds: wave w holds the rows of slots q * 16 + w =================
    int left = na;                                               // rows still active when the rounds end
    if (!paused && !no_more && na > 0 && na <= ASL && round < a.max_rounds) {
        int my[ACS], jt[ACS], i0[ACS];
        uint32_t col[ACS], ncol[ACS];
        float val[ACS], nval[ACS], pt[ACS], ct[ACS];
#pragma unroll
        for (int q = 0; q < ACS; q++) {
            const int e = q * WNW + w;
            my[q] = e < na ? uni(ld_sc1(A + e)) : -1;
            col[q] = COLSENT; val[q] = 0.0f; ncol[q] = COLSENT; nval[q] = 0.0f;
            if (my[q] >= 0) { col[q] = a.cache_col[(int64_t)my[q] * KC + lane]; val[q] = a.cache_val[(int64_t)my[q] * KC + lane]; }
        }
        int dealt = na;
#ifdef CYTO_WIDE_PROF
        long long tp[5] = {0, 0, 0, 0, 0}, tm = wall_clock64();
#define CH_LAP(k) { const long long now_ = wall_clock64(); tp[k] += now_ - tm; tm = now_; }
#endif
        for (;;) {
#ifdef CYTO_WIDE_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CH_LAP(0)
#endif
#pragma unroll
            for (int q = 0; q < ACS; q++) {
                jt[q] = -2; i0[q] = -1; pt[q] = 0.0f; ct[q] = 0.0f;
                if (my[q] >= 0) {
                    cx.bid_of(my[q], w, col[q], val[q], jt[q], pt[q], ct[q], i0[q]);
                    if (i0[q] >= 0) { ncol[q] = a.cache_col[(int64_t)i0[q] * KC + lane]; nval[q] = a.cache_val[(int64_t)i0[q] * KC + lane]; }   // in flight across the barrier
                    if (jt[q] < 0 && lane == 0) atomicAdd(&s.retired, 1);
                }
                if (lane == 0) { s.sm_j[q * WNW + w] = jt[q]; s.sm_i[q * WNW + w] = my[q]; s.sm_p[q * WNW + w] = pt[q]; }
            }
            bids += na;
#ifdef CYTO_WIDE_PROF
            CH_LAP(1)
#endif
            lds_barrier();
#ifdef CYTO_WIDE_PROF
            CH_LAP(2)
#endif
            int nact = 0;
            {
                const int oj = s.sm_j[lane], oi = s.sm_i[lane];
                const float op = s.sm_p[lane];
#pragma unroll
                for (int q = 0; q < ACS; q++) {
                    if (my[q] < 0) continue;
                    if (jt[q] < 0) { my[q] = -1; continue; }      // retired
                    const bool better = oj == jt[q] && lane != q * WNW + w && (op < pt[q] || (op == pt[q] && oi < my[q]));
                    if (!__ballot(better)) {
                        if (lane == 0) cx.apply(my[q], jt[q], pt[q], ct[q], i0[q]);
                        my[q] = i0[q]; col[q] = ncol[q]; val[q] = nval[q];     // the displaced owner bids in the next round (or nobody)
                    }
                    nact += my[q] >= 0;
                }
            }
            if (lane == 0) s.wcnt[w] = nact;
            round++;
            // (a full-row bid with its cache refresh holds up the whole round here: 16 waves wait for the one that sweeps -- a
            //  quarter of the list regime's allowance, then the rows go back to the list and the rounds pause for fresh caches)
            if (tid == 0 && a.aug_seg == 0 && a.seg_sync) {
                if (s.dense - dense0 >= (a.arr_waste + 3) / 4) { s.pause = 1; if (!announced) atomicAdd(a.seg_sync, 1); announced = true; }
                else if (a.seg_quorum > 0 && ld_sc1(a.seg_sync) >= a.seg_quorum) s.pause = 1;
            }
#ifdef CYTO_WIDE_PROF
            CH_LAP(3)
#endif
            if (VLDS) lds_barrier();          // prices and owners are exchanged through LDS: the global stores may still be in flight
            else __syncthreads();             // ... through L2: the round's stores are acknowledged before anyone bids again
            na = 0;
#pragma unroll
            for (int k = 0; k < WNW; k++) na += s.wcnt[k];
            na = uni(na);
            left = na;
#ifdef CYTO_WIDE_PROF
            CH_LAP(4)
#endif
            if (na == 0 || round >= a.max_rounds) break;
            if (uni(s.pause)) {
                cur = (int)(round & 1);                                // the list the next launch reads: by the round's parity
                A = cur ? a.act1 : a.act0;
                if (tid == 0) s.cnt[cur] = 0;
                lds_barrier();
#pragma unroll
                for (int q = 0; q < ACS; q++)
                    if (my[q] >= 0 && lane == 0) st_sc1(A + atomicAdd(&s.cnt[cur], 1), (int32_t)my[q]);
                paused = true;
                break;
            }
            if (na * 2 <= dealt && dealt > WNW) {
                // half of the rows have dropped out: deal the rest out again, round-robin over the waves
#pragma unroll
                for (int q = 0; q < ACS; q++)
                    if (my[q] >= 0 && lane == 0) s.deal[atomicAdd(&s.ndeal, 1)] = my[q];
                lds_barrier();
#pragma unroll
                for (int q = 0; q < ACS; q++) {
                    const int e = q * WNW + w;
                    my[q] = e < na ? uni(s.deal[e]) : -1;
                    if (my[q] >= 0) { col[q] = a.cache_col[(int64_t)my[q] * KC + lane]; val[q] = a.cache_val[(int64_t)my[q] * KC + lane]; }
                }
                dealt = na; n_deal++;
                lds_barrier();
                if (tid == 0) s.ndeal = 0;
            }
        }
#ifdef CYTO_WIDE_PROF
        if (tid == 0) { long long *dbg = reinterpret_cast<long long *>(a.misc + 256); for (int k = 0; k < 5; k++) dbg[12 + k > 15 ? 15 : 12 + k] = tp[k]; dbg[7] = tp[4]; }
#endif
    }
    __syncthreads();
    t_chain = wall_clock64() - t_chain0; n_chain = round - n_list;
    const long long t_tail0 = wall_clock64();
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
        wc[WC_ROUNDS] = round; wc[WC_BIDS] = bids; wc[WC_RETIRED] = s.retired; wc[WC_ACTIVE_LEFT] = left;
        wc[WC_FREE_ARR] = numfree; wc[WC_DENSE_ARR] = s.dense;
        *reinterpret_cast<int *>(a.misc + 128) = numfree;
        long long *dbg = reinterpret_cast<long long *>(a.misc + 256);      // (100 MHz ticks)
        dbg[0] = n_list; dbg[1] += t_list; dbg[2] = n_chain; dbg[3] += t_chain; dbg[4] = n_deal; dbg[6] = wall_clock64() - t_tail0;
        // the state for the next launch, if the rounds paused (cur == round & 1: both flip together)
        h->started = 1; h->round = round; h->bids = bids; h->retired = s.retired; h->dense = s.dense; h->cnt[cur] = paused ? na : 0;
        h->cnt[cur ^ 1] = 0; h->free_cr = free_cr; h->done = paused ? 0 : 1; h->launches += 1; h->list_rounds = n_list;
#ifndef CYTO_WIDE_PROF
        dbg[12] = h->launches;
#endif
        // bit 0: the rounds paused for fresh row caches; bit 1: the row reduction hardly ever had to read a full row (<= n / 64 bids):
        // the caches are evidently in good shape, the driver skips the rebuild before the searches (floors stay valid bounds while
        // prices only fall; n = 50 000: 3.3 of 50 ms)
        if (a.seg_sync) a.seg_sync[1 + blockIdx.x] = (paused ? 1 : 0) | ((!paused && s.dense <= n / 64) ? 2 : 0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// AUGMENTATION (oracle/jv_oracle_impl.h, wide mode): shortest paths with (distance, tight-hop count) labels, run speculatively.
//
// Per column (global, L2): label = (ordered distance << 32 | tight hops << 20 | predecessor row), all-ones = unlabelled; an atomic min on it
// keeps the smallest distance and, among equal distances, the lowest row -- the oracle's pred.
// In LDS: per 64-column block the smallest (distance, column) among its DIRTY columns (labelled, assigned, not settled at
// their current label), a dirty bit and an assigned bit per column, a dense bit per row (its cache could not certify the
// search: relaxed from the full cost row), the best unassigned column so far s_T = (distance, column).
// A round: every wave takes the best dirty column of the blocks it owns (block b belongs to wave b % 16), if its distance
// is below s_T's; clears the dirty bit, reads the label, relaxes the owner row's cached columns.  Barrier.  The block
// minimum of the block a wave took from is rebuilt.  Barrier.  No wave found work: converged.
// ------------------------------------------------------------------------------------------------------------------
// (scratch of the one-edge searches: the machine's arrays a.scx, free once the row reduction is over -- wide_claim_* below, and wide_aug's loop)
struct ClaimCtl { int changed, blocked, rounds, pad_; };
__device__ __forceinline__ int *cl_claim(const WideArgs &a) { return reinterpret_cast<int *>(a.scx); }
__device__ __forceinline__ unsigned long long *cl_mask(const WideArgs &a) { return reinterpret_cast<unsigned long long *>(a.scx + sc_np(a.n) * 8); }
__device__ __forceinline__ int2 *cl_state(const WideArgs &a) { return reinterpret_cast<int2 *>(a.scx + sc_np(a.n) * 16); }
__device__ __forceinline__ ClaimCtl *cl_ctl(const WideArgs &a) { return reinterpret_cast<ClaimCtl *>(a.scx + sc_np(a.n) * 24); }
// per free-list position: how many rows of its run of identical rows are left from it on (itself included)
__device__ __forceinline__ int *cl_rem(const WideArgs &a) { return reinterpret_cast<int *>(a.scx + sc_np(a.n) * 32); }
constexpr int AP = 2;              // columns a wave settles per round (their loads are in flight together)

size_t wide_aug_lds_bytes(int n, bool vlds, bool clds) {
    const size_t nblk = ((size_t)n + 63) / 64, nw32 = ((size_t)n + 31) / 32, npad = ((size_t)n + 3) & ~(size_t)3;
    return ((nblk * 8 + 3 * nw32 * 4 + npad * ((vlds ? 4 : 0) + (clds ? 2 : 0)) + 15) / 16) * 16;
}
bool wide_aug_vlds(int n) { return n <= 65534 && wide_aug_lds_bytes(n, true, true) + 4096 <= (size_t)LDS_DYNAMIC_MAX; }
bool wide_aug_clds(int n) { return n <= 65534 && wide_aug_lds_bytes(n, false, true) + 4096 <= (size_t)LDS_DYNAMIC_MAX; }
size_t wide_aug_lds_bytes(int n) { return wide_aug_lds_bytes(n, wide_aug_vlds(n), wide_aug_clds(n)); }

struct AugShared {
    unsigned long long T;          // best unassigned column: (ordered distance << 32 | tight hops << 20 | column)
    int ntouch, any[3], npk[3], fail, anydense, rootdense, doroot, f, err;
    int waste, stop;               // full-row relaxations of this launch; "return to the driver for fresh caches"
    int nhop;                      // edges of the path being flipped
    int conflict;                  // PAR: this search met a column an earlier search of the batch claimed
    int scans;
    int st_row[64], st_col[64];    // one-edge searches: their results, stored to global memory 64 at a time
    float st_val[64];
    // full-row relaxations of a round (an owner whose cache could not certify): queued by the wave that settled the column, done by the
    // WHOLE workgroup behind the round's barrier -- a 200-KB row at n = 50 000 took its one wave ~60 us while fifteen others waited at the
    // barrier (config c3: 40 of them were 2.4 of its searches' 6.1 ms)
    // (the queue lives in st_row / st_col / st_val: the one-edge loop between two searches and the rounds of a search never overlap, and
    //  the kernel's static LDS has no room for another 640 bytes beside the dynamic region; their count: npk[] >> 16)
};
static_assert(WNW * 2 <= 32, "the queue of a round's full-row relaxations: two halves of the 64-entry staging arrays");

// ---- SEVERAL SEARCHES AT ONCE (one problem, PAR): the free rows' searches are independent until they are committed, and most of
// them are shallow (a few dozen settled columns after the scaled row reduction), so G workgroups run the searches of free rows
// f, f + 1, ..., f + G - 1 from ONE state, each with its own labels, and then commit IN ROW ORDER the longest prefix whose settled
// sets (and sinks) are pairwise disjoint -- which is exactly what running them one after the other gives:
//   a committed search lowers the prices of the columns it settled below its final distance and flips owners along its path; a later
//   search that settled none of those columns, did not end at that sink and whose own sink is not among them read nothing that
//   changed in a way it can observe (a lower price of a column only RAISES the label the search would give it, and that label was
//   already at or above the search's end; unassigned columns other than the sink keep their prices) -- its labels, sink and path are
//   the ones it would have computed after the commit.  The first search of a batch that meets an earlier one's set, and every one
//   after it, is discarded and runs again in the next batch.  Nothing of the restatement changes: same searches, same order, same bits.
// Conflicts are found without a serial pass: every search claims its columns with an atomic max of (batch << 8 | 255 - g); after a
// grid barrier a search whose columns all still carry its own key met no smaller g.  The committing workgroups log their price and
// owner changes; every workgroup applies the log to its LDS copies before the next batch.  Grid barriers: three per batch.
struct ParCtl {
    unsigned int bar_arrive, bar_gen;
    int P[2], nplog[2], nolog[2];          // per batch parity: first conflicting search, entries of the two change logs
    int err, pad_;
    long long c_relax, c_hops, c_proc, c_dense, c_rounds, c_verify, c_batches, c_discarded;
};
static_assert(sizeof(ParCtl) <= 256, "control block");
constexpr int PAR_GMAX = 64;
size_t wide_par_state_bytes(int n, int G) {
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    // control block | labels G x n | touched G x n | hops G x 2n | claim n | price log (col, val) | owner log (col, owner)
    return 256 + (size_t)G * np * 8 + (size_t)G * np * 4 + (size_t)G * np * 8 + np * 4 + np * 8 + np * 8;
}
__device__ __forceinline__ void par_barrier(ParCtl *c, int G, unsigned &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned arrived = atomicAdd(&c->bar_arrive, 1u) + 1u;
        if (arrived == (gen + 1u) * (unsigned)G) st_sc1(&c->bar_gen, gen + 1u);
        else {
            long long spins = 0;
            while (ld_sc1(&c->bar_gen) <= gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1ll << 26)) { atomicExch(&c->err, 2); break; }      // (a lost workgroup must not hang the device)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    gen++;
    __syncthreads();
}

// The full-row relaxations a round of wide_aug queued (nd of them, in st_row / st_col / st_val), by the whole workgroup behind
// the round's barrier: a slice of the row per wave.  A round that queued one has settled a column: the rounds go on, and what these
// relaxations label is picked up.  Their count travels in the upper half of the round's pick counter (AugShared::npk), which every wave
// reads behind the barrier anyway: a counter of its own, read before the other two, cost every round of every search a serial LDS round
// trip (10 000-cell chunk: 10.8 -> 11.7 ms of searches that never come here).
// Rows shorter than this stay with the wave that settled the column, inside the round (a 40-KB row is ~12 us of one wave; and with that
// sweep gone from the round loop the compiler's allocation made every phase of wide_aug 7-10 % slower on the instances that have no
// full-row relaxation at all -- 10 000-cell chunk 10.8 -> 11.7 ms, profiles/r06q_coop_ab.txt).  -DCYTO_COOP_MIN_N=0: a developer build
// that takes every full row through the cooperative path (tools/run/r06r.sh runs the randomised stress on it).
#ifndef CYTO_COOP_MIN_N
#define CYTO_COOP_MIN_N 16384
#endif
constexpr int COOP_MIN_N = CYTO_COOP_MIN_N;
template <bool VLDS>
__device__ __forceinline__ void coop_dense_rows(AugShared *sp, int nd, const float *__restrict__ cost, const int32_t *__restrict__ rowmap, int64_t ld,
                                             int n, int w, int lane, unsigned long long *lbl, int32_t *tch, uint32_t *dirty,
                                             unsigned long long *bmin, const uint32_t *asg, const float *s_v, const float *v) {
    AugShared &s = *sp;
    const int nq = (n + 3) >> 2, per = (nq + WNW - 1) / WNW, q_lo = w * per, q_hi = min(nq, q_lo + per);
    for (int e = 0; e < nd; e++) {
        const int oiq = uni(s.st_row[e]), pjq = uni(s.st_row[32 + e]);
        const uint32_t dord = (uint32_t)uni(s.st_col[e]), kq = (uint32_t)uni(s.st_col[32 + e]);
        const float h = uni(s.st_val[e]);
        const float *__restrict__ row = cost + wrow_off(rowmap, oiq, ld);
        // (a full row offers hundreds of candidates below the best unassigned distance when many columns are near-equal:
        //  the label is read first -- a plain L2 load -- and the atomic follows only where it would change something)
        wave_row_sweep_q(row, q_lo, q_hi, n, lane, [&](int c, float x) {
            const float vc = VLDS ? s_v[c] : ld_sc1(v + c);
            const unsigned long long lv = edge_lv(f2ord((x - vc) - h), dord, kq);
            const unsigned long long key = lkey(lv, (uint32_t)oiq);
            if (c == pjq || lv > lv_of(s.T) || !(key < ld_sc1(lbl + c))) return;
            const unsigned long long old = atomicMin(lbl + c, key);                 // (as relax_to in wide_aug)
            if (key < old) {
                if (old == ~0ull) tch[atomicAdd(&s.ntouch, 1)] = c;
                if (lv_of(old) > lv) {
                    const unsigned long long ck = lkey(lv, (uint32_t)c);
                    if ((asg[c >> 5] >> (c & 31)) & 1u) { atomicOr(&dirty[c >> 5], 1u << (c & 31)); atomicMin(&bmin[c >> 6], ck); }
                    else atomicMin(&s.T, ck);
                }
            }
        });
    }
    __syncthreads();                                                                // (the next round reads the block minima these set)
}

// VLDS: prices (f32) and column owners (u16) also in LDS (every update goes to both copies), as in wide_arr.
template <bool VLDS, bool CLDS, bool PAR>
__global__ __launch_bounds__(WT) void wide_aug(const WideArgs *__restrict__ batch) {
    extern __shared__ __align__(16) unsigned char w_smem[];
    __shared__ AugShared s;
    // (PAR: the launch has par_groups * stride workgroups of which every stride-th takes part: stride 1 = spread over all XCDs, 8 = one XCD)
    const int pstride = PAR ? (int)(gridDim.x / (unsigned)load_wide_args(batch, 0).par_groups) : 1;
    if (PAR && (blockIdx.x % pstride)) return;
    const int g = PAR ? uni((int)(blockIdx.x / pstride)) : 0;
    const WideArgs a = load_wide_args(batch, PAR ? 0 : blockIdx.x);
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, w = uni((int)(threadIdx.x >> 6));
    const int nblk = (n + 63) / 64, nw32 = (n + 31) / 32;
    // the search's own arrays: the problem's (one search at a time) or this workgroup's (PAR)
    const int G = PAR ? a.par_groups : 1;
    const size_t np_ = ((size_t)n + 63) & ~(size_t)63;
    ParCtl *pc = reinterpret_cast<ParCtl *>(a.par);
    unsigned long long *lbl = PAR ? reinterpret_cast<unsigned long long *>(a.par + 256) + (size_t)g * np_ : a.label;
    int32_t *tch = PAR ? reinterpret_cast<int32_t *>(a.par + 256 + (size_t)G * np_ * 8) + (size_t)g * np_ : a.touched;
    int32_t *hop_i = PAR ? reinterpret_cast<int32_t *>(a.par + 256 + (size_t)G * np_ * 12) + (size_t)g * 2 * np_ : a.act0;
    int32_t *hop_j = PAR ? hop_i + np_ : a.act1;
    uint32_t *claim = reinterpret_cast<uint32_t *>(a.par + 256 + (size_t)G * np_ * 20);
    int32_t *plog_col = reinterpret_cast<int32_t *>(claim + np_);
    float *plog_val = reinterpret_cast<float *>(plog_col + np_);
    int32_t *olog_col = reinterpret_cast<int32_t *>(plog_val + np_);
    int32_t *olog_own = olog_col + np_;
    unsigned pgen = 0;
    unsigned long long *bmin = reinterpret_cast<unsigned long long *>(w_smem);
    uint32_t *dirty = reinterpret_cast<uint32_t *>(bmin + nblk);
    uint32_t *asg = dirty + nw32;
    uint32_t *dense = asg + nw32;
    float *s_v = reinterpret_cast<float *>(dense + nw32);
    uint16_t *s_cs = reinterpret_cast<uint16_t *>(s_v + (VLDS ? ((n + 3) & ~3) : 0));
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    for (int b = tid; b < nblk; b += WT) bmin[b] = ~0ull;
    for (int q = tid; q < nw32; q += WT) { dirty[q] = 0; dense[q] = 0; }
    for (int c0 = 0; c0 < nw32 * 32; c0 += WT) {                 // assigned bits, 64 columns per wave and step
        const int c = c0 + tid;
        const uint64_t m = __ballot(c < n && a.colsol[c] >= 0);
        if (lane == 0 && (c >> 5) < nw32) { asg[c >> 5] = (uint32_t)m; if ((c >> 5) + 1 < nw32) asg[(c >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    if (CLDS)
        for (int j = tid; j < n; j += WT) {
            if (VLDS) s_v[j] = a.v[j];
            const int o = a.colsol[j]; s_cs[j] = o < 0 ? (uint16_t)0xFFFFu : (uint16_t)o;
        }
    // The searches may take several launches, with the row caches rebuilt by the whole chip in between (misc + 132 = searches
    // done so far).  A cache floor bounds c - v in absolute terms, as of the build; every search lowers the prices of the columns it
    // settles and raises their rows' duals with them, so after a few DEEP searches (few-cell-type chunks: a search settles most
    // columns) the floors lie below what the next search asks for and row after row falls back to its full cost row -- while
    // fresh caches certify practically everything (10 000-cell c4 chunk: 1.14 M full-row relaxations without a rebuild, 18 000
    // with one every 32 searches).  a.aug_seg: -1 never return early; k > 0 after k searches; 0: when the full-row relaxations of
    // this launch reach a.aug_waste (what a rebuild costs), or when a.seg_quorum workgroups of the launch have asked for one.
    const int f0 = PAR ? 0 : *reinterpret_cast<const int *>(a.misc + 132);
    if (!PAR) {
        long long *wc = reinterpret_cast<long long *>(a.misc + 160);
        if (wc[WC_AUG_LAUNCHES] > 0 && f0 >= numfree) {          // finished in an earlier launch
            if (tid == 0 && a.seg_sync) a.seg_sync[1 + blockIdx.x] = 0;
            return;
        }
    }
    if (tid == 0) { s.waste = 0; s.stop = 0; }
    if (tid == 0) { s.T = ~0ull; s.ntouch = 0; s.any[0] = 0; s.any[1] = 0; s.any[2] = 0; s.npk[0] = 0; s.npk[1] = 0; s.npk[2] = 0; s.fail = 0; s.anydense = 0; s.rootdense = 0; s.doroot = 0; s.f = f0; s.err = 0; s.scans = 0; }
    __syncthreads();
    auto getv = [&](int j) -> float { return VLDS ? s_v[j] : ld_sc1(a.v + j); };
    auto getcs = [&](int j) -> int {
        if (CLDS) { const uint16_t x = s_cs[j]; return x == 0xFFFFu ? -1 : (int)x; }
        return ld_sc1(a.colsol + j);
    };
    auto is_asg = [&](int c) -> bool { return (asg[c >> 5] >> (c & 31)) & 1u; };

    long long c_relax = 0, c_hops = 0, c_rounds = 0, c_proc = 0, c_trivial = 0, c_dense = 0, c_verify = 0;   // (thread 0 / wave leaders)
    long long t_rounds = 0, t_verify = 0, t_finish = 0, t_triv = 0, t_mark = wall_clock64();

#define AUG_LAP(acc) { const long long now_ = wall_clock64(); acc += now_ - t_mark; t_mark = now_; }

    // a relaxation in two halves: the offer (column `col` is offered the ordered distance `co` by row `row`; returns the label
    // it replaced or lost against) and what follows from it -- first touch, dirty column, best unassigned column
    auto offer = [&](int col, unsigned long long lv, int row) -> unsigned long long {
        return atomicMin(lbl + col, lkey(lv, (uint32_t)row));
    };
    auto after_offer = [&](int col, unsigned long long lv, int row, unsigned long long old) {
        const unsigned long long key = lkey(lv, (uint32_t)row);
        if (key < old) {
            if (old == ~0ull) tch[atomicAdd(&s.ntouch, 1)] = col;
            if (lv_of(old) > lv) {                              // the label value itself dropped (not only the row of a tie)
                const unsigned long long ck = lkey(lv, (uint32_t)col);
                if (is_asg(col)) { atomicOr(&dirty[col >> 5], 1u << (col & 31)); atomicMin(&bmin[col >> 6], ck); }
                else atomicMin(&s.T, ck);
            }
        }
    };
    auto relax_to = [&](int col, unsigned long long lv, int row) { after_offer(col, lv, row, offer(col, lv, row)); };
    // behind a round's barrier (every wave of the workgroup, the idle ones too): the full-row relaxations the round queued
    // (coop_dense_rows, out of line: the rounds that queued none -- nearly all -- pay one LDS read for it)

    int f = f0;
    int ph = 0;                                                  // round flag in use (three take turns: one barrier per round)
    int rb[AP];                                                  // blocks this wave took from in the last round: their minima are due
#pragma unroll
    for (int q = 0; q < AP; q++) rb[q] = -1;
    int seg_done = 0;                                            // searches (other than one-edge ones) of this launch
    int wact = 4;                                                // waves that take part in the next round of a search (4 or all 16)
    bool announced = false;
    int fbase = 0, batchno = 1, perr = 0;                        // PAR: first free row of the batch, batch number (claim keys), error seen
    long long c_batches = 0, c_discarded = 0;
    for (;;) {
        // ---- wave 0 disposes of the searches that end at once: the free row's best cached column is unassigned and its
        // cache certifies that (no column settled, no price changes: the path is one edge) ----
        if (!PAR && w == 0) {
            // A pipeline over the free list (35 000 such searches at c3), three steps deep: while step k is decided, the prices of step
            // k + 1's cached columns, the cache row of step k + 2 and the position / row / length of step k + 3 are in flight --
            // prices do not change in this loop.  A STEP is a run of identical rows (ten slots per spot at c3: free together, identical
            // caches, consecutive in the free list; their lengths come from the whole-chip pass before the kernel, cl_rem): with the
            // prices fixed, the t-th row of the run takes the t-th lowest of the unassigned columns that tie at the minimum -- the whole
            // run is one step, and the step after it is known before this one is decided.  (Rounds 3-4 found a run's end inside the
            // step and started the pipeline again behind every run: four dependent loads per run, 1.5 us.)
            auto flush_one_edge = [&](int cnt) {
                if (lane < cnt) {
                    const int r = s.st_row[lane], cc = s.st_col[lane];
                    a.rowsol[r] = cc; a.colsol[cc] = r; a.cassign[cc] = s.st_val[lane];
                }
            };
            const int *remp = (a.same_prev && a.scx) ? cl_rem(a) : nullptr;
            int nst = 0;
            // stage 0: decided now; stage 1: cache row here, prices in flight; stage 2: row known, cache row in flight; stage 3: position
            // known, row and length in flight (as loaded: made uniform when the stage moves up)
            int p0, l0, r0, p1, l1, r1, p2, l2, r2, p3, l3v, r3v;
            uint32_t col0, col1, col2; float val0, val1, val2, vv0, vv1;
            auto start = [&](int ff) {                               // (cold: a chain of loads -- at the start and behind a run cut short)
                auto len_of = [&](int pp) -> int { return pp < numfree ? (remp ? uni(remp[pp]) : 1) : 1; };
                auto row_at = [&](int pp) -> int { return pp < numfree ? uni(a.freerows[pp]) : -1; };
                p0 = ff; l0 = len_of(p0); r0 = row_at(p0);
                p1 = p0 + l0; l1 = len_of(p1); r1 = row_at(p1);
                p2 = p1 + l1; l2 = len_of(p2); r2 = row_at(p2);
                p3 = p2 + l2; l3v = p3 < numfree ? (remp ? remp[p3] : 1) : 1; r3v = p3 < numfree ? a.freerows[p3] : -1;
                col0 = COLSENT; col1 = COLSENT; col2 = COLSENT; val0 = 0.0f; val1 = 0.0f; val2 = 0.0f;
                if (r0 >= 0) { col0 = a.cache_col[(int64_t)r0 * KC + lane]; val0 = a.cache_val[(int64_t)r0 * KC + lane]; }
                if (r1 >= 0) { col1 = a.cache_col[(int64_t)r1 * KC + lane]; val1 = a.cache_val[(int64_t)r1 * KC + lane]; }
                if (r2 >= 0) { col2 = a.cache_col[(int64_t)r2 * KC + lane]; val2 = a.cache_val[(int64_t)r2 * KC + lane]; }
                vv0 = (r0 >= 0 && lane < KCU && col0 != COLSENT) ? getv((int)col0) : 0.0f;
                vv1 = (r1 >= 0 && lane < KCU && col1 != COLSENT) ? getv((int)col1) : 0.0f;
            };
            start(f);
            while (f < numfree) {
                const int fr = r0, L = l0 < 1 ? 1 : l0;
                const uint32_t col = col0;
                const float val = val0, vv = vv0;
                // the stages move up (as if this step took its whole run: else the pipeline starts again below)
                const int n_p3 = p3, n_l3 = uni(l3v), n_r3 = uni(r3v);          // stage 3's row and length have arrived
                p0 = p1; l0 = l1; r0 = r1; col0 = col1; val0 = val1; vv0 = vv1;
                p1 = p2; l1 = l2; r1 = r2; col1 = col2; val1 = val2;
                vv1 = (r1 >= 0 && lane < KCU && col1 != COLSENT) ? getv((int)col1) : 0.0f;
                p2 = n_p3; l2 = n_l3 < 1 ? 1 : n_l3; r2 = n_r3;
                col2 = COLSENT; val2 = 0.0f;
                if (r2 >= 0) { col2 = a.cache_col[(int64_t)r2 * KC + lane]; val2 = a.cache_val[(int64_t)r2 * KC + lane]; }
                p3 = p2 + l2;
                l3v = p3 < numfree ? (remp ? remp[p3] : 1) : 1; r3v = p3 < numfree ? a.freerows[p3] : -1;
                const float tau = rdlane(val, KCU);
                const bool valid = lane < KCU && col != COLSENT;
                const uint32_t od = valid ? f2ord(val - vv) : 0xFFFFFFFFu;
                const uint32_t omin = wave_min_u32(od);
                const bool un = valid && od == omin && !is_asg((int)col);
                const uint64_t mu = __ballot(un);
                if (!(omin != 0xFFFFFFFFu && mu && tau > ord2f(omin))) break;
                // this row and the identical free rows right behind it, as many as there are tied unassigned columns
                const int cntmu = __popcll(mu);
                const int m = L < cntmu ? L : cntmu;
                if (nst + m > 64) { flush_one_edge(nst); nst = 0; }
                const int rank = __popcll(mu & lanemask_lt());           // cache rows are sorted by column: ranks go by column
                if (un && rank < m) {
                    // (no global store in the loop: a load is only returned after the stores issued before it are acknowledged, so
                    //  three stores per search made every search wait for the previous one's)
                    s.st_row[nst + rank] = fr + rank; s.st_col[nst + rank] = (int)col; s.st_val[nst + rank] = val;
                    if (CLDS) s_cs[col] = (uint16_t)(fr + rank);
                    atomicOr(&asg[col >> 5], 1u << (col & 31));
                }
                c_trivial += m; c_hops += m;
                f += m;
                nst += m;
                if (nst == 64) { flush_one_edge(nst); nst = 0; }
                if (m != L && f < numfree) start(f);                     // (the run's other rows are not one-edge searches: most likely the loop ends with the next step)
            }
            flush_one_edge(nst);
            if (lane == 0) {
                s.f = f;
                if (seg_done > 0 && f < numfree) {               // back to the driver?  (never before one search is done)
                    bool stop = a.aug_seg > 0 && seg_done >= a.aug_seg;
                    if (a.aug_seg == 0) {
                        if (s.waste >= a.aug_waste) { stop = true; if (!announced && a.seg_sync) atomicAdd(a.seg_sync, 1); announced = true; }
                        else if (a.seg_sync && a.seg_quorum > 0 && ld_sc1(a.seg_sync) >= a.seg_quorum) stop = true;
                    }
                    if (stop) s.stop = 1;
                }
            }
        }
        __syncthreads();
        AUG_LAP(t_triv)
        if (PAR) {
            if (fbase >= numfree || perr) break;                 // (the same decision in every workgroup)
            f = fbase + g;
        } else {
            f = uni(s.f);
            if (f >= numfree || uni(s.stop)) break;
        }
        const bool active = !PAR || f < numfree;                  // (PAR: the last batch may have fewer searches than workgroups)
        seg_done++;
        const int fr = a.freerows[active ? f : 0];
        const float *__restrict__ frow = a.cost + wrow_off(a.rowmap, fr, a.ld);
        const float ftau = a.cache_val[(int64_t)fr * KC + KCU];

        if (active) {
        // ---- root: d[j] = c[fr][j] - v[j] for the cached columns (pred = fr) ----
        if (w == 0) {
            const uint32_t col = a.cache_col[(int64_t)fr * KC + lane];
            const float val = a.cache_val[(int64_t)fr * KC + lane];
            if (lane < KCU && col != COLSENT) relax_to((int)col, (unsigned long long)f2ord(val - getv((int)col)) << 12, fr);
        }
        __syncthreads();

        for (;;) {
            // ================= rounds until no wave finds work =================
            int nd = 0;                                              // full-row relaxations the last round queued (leaves the rounds for them)
            for (;;) {
                const unsigned long long Tlv = uni(lv_of(s.T));          // label value of the best unassigned column so far
                // the two best dirty columns among the wave's blocks (one key per lane: the smallest of the blocks it looks at)
                // How many waves take part in a round: a round is bound by the CU's instruction issue (16 waves on 4 SIMDs: ~500
                // instructions each), so while the frontier is narrow -- shallow searches, the first and last rounds of deep ones --
                // four waves (one per SIMD, eight settlements at most) make a round several times shorter.  Block b belongs to wave
                // b mod W in this round; which wave settles what is a matter of efficiency only.
                const int W = wact;
                if (w >= W && rb[0] < 0 && rb[1] < 0) {              // nothing to pick from, no minimum to rebuild: straight to the barrier
                    __syncthreads();
                    const int any0 = uni(s.any[ph]), npd0 = uni(s.npk[ph]), np0 = npd0 & 0xFFFF;
                    ph = (ph + 1) % 3;
                    c_rounds++;
                    wact = np0 > 6 ? WNW : 4;
                    nd = npd0 >> 16;
                    if (!any0 || nd) break;
                    continue;
                }
                unsigned long long mk = ~0ull;
                if (w < W) for (int b = w + W * lane; b < nblk; b += W * 64) mk = umin64(mk, bmin[b]);
                // Which dirty column a wave settles next is a matter of efficiency only (any schedule reaches the same labels), what
                // must be exact is WHETHER a block holds work: a lane is eligible if its best key is below the best unassigned
                // column's label; among the eligible lanes the smallest distance goes first (one 32-bit reduction and a ballot per
                // pick -- the tight-hop counts do not order the picks).
                bool pk[AP]; int pj[AP], oi[AP];
                unsigned long long lab[AP];
                float ca[AP], vp[AP], val[AP];
                uint32_t col[AP];
                {
                    uint32_t dk = lv_of(mk) < Tlv ? (uint32_t)(mk >> 32) : 0xFFFFFFFFu;
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        const uint32_t m = wave_min_u32(dk);
                        pk[q] = m != 0xFFFFFFFFu;
                        const int l = pk[q] ? __ffsll((unsigned long long)__ballot(dk == m)) - 1 : 0;
                        pj[q] = (int)(rdlane((uint32_t)mk, l) & 0xFFFFFu);
                        if (lane == l) dk = 0xFFFFFFFFu;
                    }
                }
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    // (the block's minimum is void from here on; it is rebuilt in the NEXT round, see below)
                    if (pk[q] && lane == 0) { atomicAnd(&dirty[pj[q] >> 5], ~(1u << (pj[q] & 31))); bmin[pj[q] >> 6] = ~0ull; }
                }
                if (lane == 0 && (pk[0] || pk[1] || rb[0] >= 0 || rb[1] >= 0)) { s.any[ph] = 1; if (pk[0]) atomicAdd(&s.npk[ph], (int)pk[0] + (int)pk[1]); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the bits are cleared before the labels are read
                // Everything the two settlements read is requested in ONE go, without a branch in between: label and owner entry of
                // both columns, then (the owners come from LDS) both cache rows -- a wave without a pick reads column 0's, harmlessly.
                // (With the loads inside `if (picked)` blocks the compiler drains the memory counter at every join: label 0, label 1,
                //  cache rows, offer 0, offer 1 were five L2 round trips in a row, most of a round's 4.1 us.)
                int pjx[AP];
                unsigned long long lab_r[AP];
                float ca_r[AP], vp_r[AP];
                int oi_r[AP];
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    pjx[q] = pk[q] ? pj[q] : 0;
                    lab_r[q] = ld_sc1(lbl + pjx[q]); ca_r[q] = ld_sc1(a.cassign + pjx[q]);
                }
                // ... and with them the labels of the dirty columns of the blocks the wave took from in the LAST round: the new
                // minimum of such a block is not on any round's critical path this way (it used to be a third round trip, behind a
                // second barrier).  Safe against the offers that go on meanwhile: the minimum was voided at the pick (store), the
                // dirty bits are read here, the labels after them, the result is merged with an atomic min -- an offer either set
                // its dirty bit before this read (then its label, completed before the bit, is seen) or lowers the minimum itself
                // after the void store.  The block yields no pick for one round; no round ends the search while a rebuild is due.
                // (When the number of waves taking part changes between two rounds -- 4 <-> 16 -- a block's NEW owner may pick from it in the very
                //  round its old owner's rebuild is merged: the rebuild can then put back a column that is being settled.  Labels are a
                //  fixed point, so results do not depend on it; the column is settled once more, which is why wide_aug_rounds and
                //  wide_aug_settled -- and nothing else -- may differ from run to run by a few units.)
                bool db[AP]; unsigned long long lbr[AP];
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    const int c = (rb[q] < 0 ? 0 : rb[q]) * 64 + lane;
                    db[q] = rb[q] >= 0 && c < n && ((dirty[c >> 5] >> (c & 31)) & 1u);
                    lbr[q] = ld_sc1(lbl + (db[q] ? c : pjx[q]));
                }
#pragma unroll
                for (int q = 0; q < AP; q++) { vp_r[q] = getv(pjx[q]); oi_r[q] = getcs(pjx[q]); }
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    oi[q] = uni(oi_r[q]);
                    const int oix = oi[q] < 0 ? 0 : oi[q];
                    col[q] = a.cache_col[(int64_t)oix * KC + lane]; val[q] = a.cache_val[(int64_t)oix * KC + lane];
                }
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    lab[q] = pk[q] ? uni(lab_r[q]) : ~0ull; ca[q] = uni(ca_r[q]); vp[q] = uni(vp_r[q]);
                    if (!pk[q]) { col[q] = COLSENT; oi[q] = 0; }
                }
                unsigned long long old[AP], co[AP];
                bool off[AP], dn[AP];
                float vc[AP];
#pragma unroll
                for (int q = 0; q < AP; q++) {                       // (the cached columns' prices: LDS, or one more request for both)
                    const bool valid = lane < KCU && col[q] != COLSENT;
                    vc[q] = getv(valid ? (int)col[q] : pjx[q]);
                }
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    off[q] = false; dn[q] = false; old[q] = 0; co[q] = 0;
                    const uint32_t dord = (uint32_t)(lab[q] >> 32), kq = (uint32_t)(lab[q] >> 20) & LKMAX;
                    if (pk[q] && lv_of(lab[q]) < Tlv) {
                        c_proc++;
                        if ((dense[oi[q] >> 5] >> (oi[q] & 31)) & 1u) { dn[q] = true; continue; }
                        const float h = (ca[q] - vp[q]) - ord2f(dord);
                        const unsigned long long lv = edge_lv(f2ord((val[q] - vc[q]) - h), dord, kq);
                        off[q] = lane < KCU && col[q] != COLSENT && (int)col[q] != pj[q] && lv <= Tlv;
                        co[q] = lv;
                    }
                }
                {   // both columns' offers (a returning atomic min per lane that has one) in flight together: the lanes are masked by
                    // hand -- as two `if` blocks each atomic was followed by a full wait at the block's end
                    static_assert(AP == 2, "two offers per wave and round");
                    const uint64_t m0 = __ballot(off[0]), m1 = __ballot(off[1]);
                    if (m0 | m1) {
                        unsigned long long *p0 = lbl + (off[0] ? (int)col[0] : 0), *p1 = lbl + (off[1] ? (int)col[1] : 0);
                        const unsigned long long k0 = lkey(co[0], (uint32_t)oi[0]), k1 = lkey(co[1], (uint32_t)oi[1]);
                        unsigned long long r0, r1;
                        uint64_t sv;
                        asm volatile("s_mov_b64 %[sv], exec\n\t"
                                     "s_and_b64 exec, %[sv], %[m0]\n\t"
                                     "global_atomic_umin_x2 %[r0], %[a0], %[d0], off sc0\n\t"
                                     "s_and_b64 exec, %[sv], %[m1]\n\t"
                                     "global_atomic_umin_x2 %[r1], %[a1], %[d1], off sc0\n\t"
                                     "s_mov_b64 exec, %[sv]\n\t"
                                     "s_waitcnt vmcnt(0)"
                                     : [r0] "=&v"(r0), [r1] "=&v"(r1), [sv] "=&s"(sv)
                                     : [a0] "v"(p0), [d0] "v"(k0), [a1] "v"(p1), [d1] "v"(k1), [m0] "s"(m0), [m1] "s"(m1)
                                     : "memory");
                        old[0] = off[0] ? r0 : 0; old[1] = off[1] ? r1 : 0;
                    }
                }
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    const bool better = off[q] && (lkey(co[q], (uint32_t)oi[q]) < old[q]);
                    if (__ballot(better) && better) after_offer((int)col[q], co[q], oi[q], old[q]);
                }
#pragma unroll
                for (int q = 0; q < AP; q++) {
                    if (!dn[q]) continue;                          // the owner's cache could not certify: its whole cost row
                    const uint32_t dord = (uint32_t)(lab[q] >> 32), kq = (uint32_t)(lab[q] >> 20) & LKMAX;
                    const float h = (ca[q] - vp[q]) - ord2f(dord);
                    if (n >= COOP_MIN_N) {                         // a long row: by the whole workgroup, behind the round's barrier (coop_dense_rows)
                        if (lane == 0) {
                            const int e = atomicAdd(&s.npk[ph], 1 << 16) >> 16;
                            s.st_row[e] = oi[q]; s.st_row[32 + e] = pj[q]; s.st_col[e] = (int)dord; s.st_col[32 + e] = (int)kq; s.st_val[e] = h;
                        }
                    } else {
                        const float *__restrict__ row = a.cost + wrow_off(a.rowmap, oi[q], a.ld);
                        const int pjq = pj[q], oiq = oi[q];
                        // (a full row offers hundreds of candidates below the best unassigned distance when many columns are near-equal:
                        //  the label is read first -- a plain L2 load -- and the atomic follows only where it would change something)
                        wave_row_sweep(row, n, lane, [&](int c, float x) {
                            const unsigned long long lv = edge_lv(f2ord((x - getv(c)) - h), dord, kq);
                            if (c != pjq && lv <= lv_of(s.T) && (lkey(lv, (uint32_t)oiq) < ld_sc1(lbl + c))) relax_to(c, lv, oiq);
                        });
                    }
                    c_dense++;
                    if (lane == 0) atomicAdd(&s.waste, 1);
                }
#pragma unroll
                for (int q = 0; q < AP; q++) {                       // last round's blocks: smallest (distance, k) among their dirty columns
                    if (rb[q] >= 0) {
                        const int c = rb[q] * 64 + lane;
                        // (the smallest distance with hop count 0 and one of its columns: a lower bound of the block's smallest label
                        //  is all a minimum has to be -- a pick re-reads the label)
                        const uint32_t dk = db[q] ? (uint32_t)(lbr[q] >> 32) : 0xFFFFFFFFu;
                        const uint32_t m = wave_min_u32(dk);
                        if (m != 0xFFFFFFFFu) {
                            const int l = __ffsll((unsigned long long)__ballot(dk == m)) - 1;
                            if (lane == 0) atomicMin(&bmin[rb[q]], ((unsigned long long)m << 32) | (uint32_t)(c - lane + l));
                        }
                    }
                    rb[q] = pk[q] ? (pj[q] >> 6) : -1;
                }
                __syncthreads();
                const int any = uni(s.any[ph]), npd = uni(s.npk[ph]), np = npd & 0xFFFF;     // picks of the round | full rows it queued << 16
                if (tid == 0) { s.any[(ph + 2) % 3] = 0; s.npk[(ph + 2) % 3] = 0; }   // (last read before this barrier, next set after the next one)
                ph = (ph + 1) % 3;
                c_rounds++;
                wact = np > 6 ? WNW : 4;
                nd = npd >> 16;
                if (!any || nd) break;
            }
            if (nd) {                                                // (outside the round loop: its code must not weigh on the rounds)
                coop_dense_rows<VLDS>(&s, nd, a.cost, a.rowmap, a.ld, n, w, lane, lbl, tch, dirty, bmin, asg, s_v, a.v);
                continue;
            }
            // ================= converged: do the caches certify what was skipped? =================
            AUG_LAP(t_rounds)
            const unsigned long long Tk = s.T;
            const uint32_t Dord = (uint32_t)(Tk >> 32);
            const float D = Tk == ~0ull ? INFINITY : ord2f(Dord);
            const int nt = s.ntouch;
            for (int q = tid; q < nt; q += WT) {
                const int k = ld_sc1(tch + q);
                const unsigned long long lbk = ld_sc1(lbl + k);
                const uint32_t dord = (uint32_t)(lbk >> 32);
                // every settled column: label below the end's -- also at the end's DISTANCE with fewer tight hops: an uncached
                // tight edge out of such a column would give (distance, k + 1), possibly below the end's label
                if (lv_of(lbk) < lv_of(Tk) && is_asg(k)) {
                    const int i = getcs(k);
                    if (!((dense[i >> 5] >> (i & 31)) & 1u)) {
                        const float h = (ld_sc1(a.cassign + k) - getv(k)) - ord2f(dord);
                        const float bound = a.cache_val[(int64_t)i * KC + KCU] - h;
                        if (!(bound > D)) {
                            atomicOr(&dense[i >> 5], 1u << (i & 31));
                            atomicOr(&dirty[k >> 5], 1u << (k & 31));
                            atomicMin(&bmin[k >> 6], lkey(lv_of(lbk), (uint32_t)k));
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
                    const unsigned long long lv = (unsigned long long)f2ord(frow[c] - getv(c)) << 12;
                    if (lv <= lv_of(s.T) && (lkey(lv, (uint32_t)fr) < ld_sc1(lbl + c))) relax_to(c, lv, fr);
                }
            __syncthreads();
            if (tid == 0) { if (fail) s.anydense = 1; s.fail = 0; s.doroot = 0; }
            __syncthreads();
            AUG_LAP(t_verify)
            if (!fail) break;
        }

        }   // (active)

        // ---- the search has ended at s.T: [PAR: claim, find the conflict-free prefix of the batch;] price update, path flip, reset ----
        const unsigned long long Tk = active ? s.T : ~0ull;
        if (!PAR && Tk == ~0ull) { if (tid == 0) s.err = 1; __syncthreads(); break; }
        const uint32_t Dord = (uint32_t)(Tk >> 32);
        const float D = ord2f(Dord);
        const int sink = (int)lid_of(Tk);
        const int nt = active ? s.ntouch : 0;
        bool commit = true;
        int Pb = G;
        const int par = batchno & 1;
        if (PAR) {
            const bool have = Tk != ~0ull;
            if (active && !have && tid == 0) atomicExch(&pc->err, 1);
            // every column this search settled (label below its end's) and its sink: claimed with (batch, workgroup) -- the atomic max
            // keeps the LOWEST workgroup of the LATEST batch
            const uint32_t key = ((uint32_t)batchno << 8) | (uint32_t)(255 - g);
            if (have) {
                for (int q = tid; q < nt; q += WT) {
                    const int k = ld_sc1(tch + q);
                    if (lv_of(ld_sc1(lbl + k)) < lv_of(Tk) && is_asg(k)) atomicMax(claim + k, key);
                }
                if (tid == 0) atomicMax(claim + sink, key);
            }
            if (tid == 0) s.conflict = 0;
            par_barrier(pc, G, pgen);
            if (have) {
                int c = 0;
                for (int q = tid; q < nt; q += WT) {
                    const int k = ld_sc1(tch + q);
                    if (lv_of(ld_sc1(lbl + k)) < lv_of(Tk) && is_asg(k) && ld_sc1(claim + k) != key) c = 1;
                }
                if (tid == 0 && ld_sc1(claim + sink) != key) c = 1;
                if (c) s.conflict = 1;
            }
            __syncthreads();
            if (tid == 0 && active && (s.conflict || !have)) atomicMin(&pc->P[par], g);
            par_barrier(pc, G, pgen);
            Pb = uni(ld_sc1(&pc->P[par]));
            Pb = Pb < G ? Pb : G;
            // (the other parity's words are free: their last readers passed this batch's first barrier)
            if (g == 0 && tid == 0) { st_sc1(&pc->P[par ^ 1], PAR_GMAX + 1); st_sc1(&pc->nplog[par ^ 1], 0); st_sc1(&pc->nolog[par ^ 1], 0); }
            commit = active && have && g < Pb;
            if (active && !commit) c_discarded++;
        }
#ifdef CYTO_AUG_FIN_SPLIT                                        // (developer build: the finish by parts -- the wait for the batch's slowest search, the
        AUG_LAP(t_verify)                                        //  claims and the conflict check go into the certificate timer, ...
#endif
        if (commit) {
            int myscans = 0;
            for (int q = tid; q < nt; q += WT) {
                const int k = ld_sc1(tch + q);
                const uint32_t dord = (uint32_t)(ld_sc1(lbl + k) >> 32);
                if (dord < Dord && is_asg(k)) {
                    const float vk = getv(k);
                    const float nv = (vk + ord2f(dord)) - D;
                    if (nv < vk) {
                        a.v[k] = nv; if (VLDS) s_v[k] = nv;
                        if (PAR) { const int e = atomicAdd(&pc->nplog[par], 1); plog_col[e] = k; plog_val[e] = nv; }
                    }
                    myscans++;
                }
            }
            if (myscans) atomicAdd(&s.scans, myscans);
            if (tid == 0) {
                // (the new owner entries' costs are fetched afterwards, by everybody: inside this walk every hop waited for its cost --
                //  an HBM access, and loads return in order -- before the next hop's label arrived: 0.7 us per hop, 37 824 hops on a c4 chunk)
                int j = sink, nh = 0;
                for (;;) {
                    const int i = (int)lid_of(ld_sc1(lbl + j));
                    const int jn = ld_sc1(a.rowsol + i);
                    a.colsol[j] = i; a.rowsol[i] = j;
                    st_sc1(hop_i + nh, (int32_t)i); st_sc1(hop_j + nh, (int32_t)j); nh++;
                    if (CLDS) s_cs[j] = (uint16_t)i;
                    c_hops++;
                    if (i == fr) break;
                    j = jn;
                }
                s.nhop = nh;
                atomicOr(&asg[sink >> 5], 1u << (sink & 31));
            }
            __syncthreads();
            for (int q = tid; q < s.nhop; q += WT) {
                const int i = ld_sc1(hop_i + q), j = ld_sc1(hop_j + q);
                a.cassign[j] = a.cost[wrow_off(a.rowmap, i, a.ld) + j];
                if (PAR) { const int e = atomicAdd(&pc->nolog[par], 1); olog_col[e] = j; olog_own[e] = i; }
            }
        }
        for (int q = tid; q < nt; q += WT) {
            const int k = ld_sc1(tch + q);
            lbl[k] = ~0ull;
            atomicAnd(&dirty[k >> 5], ~(1u << (k & 31)));
            bmin[k >> 6] = ~0ull;
        }
#ifdef CYTO_AUG_FIN_SPLIT                                        //  ... update + flip stay, reset + logs + the batch's last barrier into the one-edge timer)
        AUG_LAP(t_finish)
#endif
        // (keeping the dense bits across searches was measured: fewer rounds, but more full-row relaxations -- slower)
        if (s.anydense) for (int q = tid; q < nw32; q += WT) dense[q] = 0;
        __syncthreads();
        if (tid == 0) { if (commit) c_relax += s.scans; s.scans = 0; s.T = ~0ull; s.ntouch = 0; s.anydense = 0; s.rootdense = 0; s.f = f + 1; }
        f++;
        __syncthreads();
        if (PAR) {
            // the batch's changes, into this workgroup's copies (its own included: harmless); then the next batch
            par_barrier(pc, G, pgen);
            perr = uni(ld_sc1(&pc->err));
            const int npl = uni(ld_sc1(&pc->nplog[par])), nol = uni(ld_sc1(&pc->nolog[par]));
            if (VLDS) for (int e = tid; e < npl; e += WT) s_v[ld_sc1(plog_col + e)] = ld_sc1(plog_val + e);
            for (int e = tid; e < nol; e += WT) {
                const int col = ld_sc1(olog_col + e);
                if (CLDS) s_cs[col] = (uint16_t)ld_sc1(olog_own + e);
                atomicOr(&asg[col >> 5], 1u << (col & 31));
            }
            __syncthreads();
            const int left = numfree - fbase;
            fbase += Pb < left ? Pb : left;                       // (Pb >= 1: the first search of a batch never conflicts)
            if (Pb < 1) perr = 1;
            batchno++; c_batches++;
        }
#ifdef CYTO_AUG_FIN_SPLIT
        AUG_LAP(t_triv)
#else
        AUG_LAP(t_finish)
#endif
    }

    // ---- duals, total, counters ----
    __syncthreads();
    __shared__ double s_tot[WNW];
    __shared__ long long s_wc[WNW][4];
    if (lane == 0) { s_wc[w][0] = c_proc; s_wc[w][1] = c_dense; s_wc[w][2] = c_trivial; s_wc[w][3] = (w == 0) ? c_hops : 0; }
    __syncthreads();
    long long proc = 0, dn = 0, triv = 0, hops0 = 0;
    for (int k = 0; k < WNW; k++) { proc += s_wc[k][0]; dn += s_wc[k][1]; triv += s_wc[k][2]; hops0 += s_wc[k][3]; }
    if (PAR) {                                                   // the workgroups' counters meet in the control block; workgroup 0 finishes
        if (tid == 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&pc->c_relax), (unsigned long long)c_relax);
            atomicAdd(reinterpret_cast<unsigned long long *>(&pc->c_hops), (unsigned long long)hops0);
            atomicAdd(reinterpret_cast<unsigned long long *>(&pc->c_proc), (unsigned long long)proc);
            atomicAdd(reinterpret_cast<unsigned long long *>(&pc->c_dense), (unsigned long long)dn);
            atomicAdd(reinterpret_cast<unsigned long long *>(&pc->c_discarded), (unsigned long long)c_discarded);
            if (g == 0) { pc->c_rounds = c_rounds; pc->c_verify = c_verify; pc->c_batches = c_batches; }
        }
        par_barrier(pc, G, pgen);
        if (g != 0) return;
        c_relax = (long long)ld_sc1(reinterpret_cast<unsigned long long *>(&pc->c_relax));
        hops0 = (long long)ld_sc1(reinterpret_cast<unsigned long long *>(&pc->c_hops));
        proc = (long long)ld_sc1(reinterpret_cast<unsigned long long *>(&pc->c_proc));
        dn = (long long)ld_sc1(reinterpret_cast<unsigned long long *>(&pc->c_dense));
        f = perr || ld_sc1(&pc->err) ? f : numfree;
    }
    double tot = 0.0;
    for (int i = tid; i < n; i += WT) {
        const int j = ld_sc1(a.rowsol + i);
        if (j >= 0) {
            const float cij = ld_sc1(a.cassign + j);
            a.u[i] = cij - ld_sc1(a.v + j);
            tot += (double)cij;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if (lane == 0) s_tot[w] = tot;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < WNW; k++) t += s_tot[k];
        const bool err = s.err || (PAR && (perr || ld_sc1(&pc->err)));
        *reinterpret_cast<double *>(a.misc + 8) = t;
        long long *ctr = reinterpret_cast<long long *>(a.misc + 16);
        long long *wc = reinterpret_cast<long long *>(a.misc + 160);
        ctr[C_AUG_INIT] = numfree; ctr[C_AUG_RELAX] += c_relax; ctr[C_AUGS] = numfree; ctr[C_HOPS] += hops0;
        wc[WC_DENSE_AUG] += dn; wc[WC_AUG_LAUNCHES] += 1; wc[WC_AUG_ROUNDS] += c_rounds; wc[WC_AUG_PROCESSED] += proc; wc[WC_TRIVIAL] += triv; wc[WC_VERIFY_PASSES] += c_verify;
        if (err) *reinterpret_cast<int *>(a.misc + 4) = 1;
        *reinterpret_cast<int *>(a.misc + 132) = err ? numfree : f;      // searches done (an error ends them)
        if (a.seg_sync) a.seg_sync[1 + (PAR ? 0 : blockIdx.x)] = err ? 0 : numfree - f;
        long long *dbg = reinterpret_cast<long long *>(a.misc + 256);      // (100 MHz ticks)
        dbg[8] += t_rounds; dbg[9] += t_verify; dbg[10] += t_finish; dbg[11] += t_triv;
        if (PAR) { dbg[15] = ld_sc1(reinterpret_cast<unsigned long long *>(&pc->c_batches)); dbg[7] = ld_sc1(reinterpret_cast<unsigned long long *>(&pc->c_discarded)); }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// AUGMENTATION on SEVERAL workgroups (one problem): the same searches, the same labels -- the fixed point of §"AUGMENTATION"
// above does not care who relaxes what when -- with the search state in global memory (L2) instead of LDS and NO barrier
// inside a search's steady state: every wave of every workgroup loops on its own { take the best dirty columns of the blocks
// it owns, settle them, rebuild those blocks' minima }.  Exchange is by agent-scope atomics only (labels, dirty bits, block
// minima, best unassigned column); everything mutable is read with sc1 loads and written with sc1 stores / atomics.
//   * block minimum rebuilt without losing a concurrent update: the owner stores all-ones, WAITS, reads the block's dirty bits
//     and labels, then atomic-mins what it found; an updater completes its label atomic, then its dirty bit (waits), then
//     atomic-mins the block minimum -- whichever way they interleave, the owner either sees the dirty column or the
//     updater's atomic min lands after the owner's store;
//   * settling: dirty bit cleared, WAIT, label read -- an update in between sets the bit again;
//   * termination: a wave without work leaves the `active` count and polls; all waves idle -> grid barrier -> did anyone
//     settle anything since the last barrier?  No: converged (nothing can appear while nobody runs).  Yes: once more.
// Grid barriers (a monotonic arrival counter, bounded spins) separate only the phases of a search: root, rounds, certificate
// pass, price update + path flip, reset.  Workgroups 0, 8, 16 ... of the launch take part (observed: block b runs on XCD b % 8,
// so they share one L2 -- a speed matter only, nothing here depends on placement).
// ------------------------------------------------------------------------------------------------------------------
struct McCtl {
    unsigned long long T;
    int ntouch, active, progress[2], fail, doroot, rootdense, anydense, f, err, scans;
    unsigned int bar_arrive, bar_gen;
    int pad_;
    long long c_relax, c_hops, c_macro, c_proc, c_dense, c_trivial, c_verify;
};
static_assert(sizeof(McCtl) <= 256, "control block");

// per 64-column block ONE 128-byte record (its own cache line: the block minimum and the dirty bits take atomics from every
// workgroup, and atomics to one line serialise): [0] smallest dirty label, [1] dirty bits
constexpr int MC_REC = 16;             // 64-bit words per record
size_t wide_mc_state_bytes(int n) {
    const size_t nblk = ((size_t)n + 63) / 64, nw32 = ((size_t)n + 31) / 32;
    return ((nblk * 128 + 2 * nw32 * 4 + 255) / 256) * 256 + 256;
}
int wide_mc_groups(int nb, int n) {
    // Measured (round 3, tools/wide_large.py --groups G): on uniform instances the one-workgroup kernel is faster (n = 20 000: 39 ms
    // against 46 ms with 4-8 groups, 55 ms with 16) -- 4-8x the waves settle 1.8-2.6x the columns (speculation further from the
    // frontier) and every exchange is a global atomic instead of an LDS one; on few-cell-type chunks, where full-row relaxations
    // dominate, 8 groups are 1.4x faster (10 000-cell chunk: 2.55 -> 1.79 s).  So: on request only (cyto_lap_opts.wide_groups).
    (void)nb; (void)n;
    return 0;
}

#define MC_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

__device__ __forceinline__ void mc_barrier(McCtl *c, int G, unsigned &gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        MC_WAIT_VM();
        const unsigned arrived = atomicAdd(&c->bar_arrive, 1u) + 1u;
        if (arrived == (gen + 1u) * (unsigned)G) st_sc1(&c->bar_gen, gen + 1u);
        else {
            long long spins = 0;
            while (ld_sc1(&c->bar_gen) <= gen) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1ll << 25)) { atomicExch(&c->err, 2); break; }      // (a lost workgroup must not hang the device)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    gen++;
    __syncthreads();
}

__global__ __launch_bounds__(WT) void wide_aug_mc(const WideArgs *__restrict__ batch) {
    if (blockIdx.x & 7) return;
    const int g = uni((int)(blockIdx.x >> 3));
    const WideArgs a = load_wide_args(batch, 0);
    const int G = a.mc_groups;
    McCtl *c = reinterpret_cast<McCtl *>(a.ctl);
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, w = uni((int)(threadIdx.x >> 6));
    const int gw = g * WNW + w, W = G * WNW, gtid = g * WT + tid, GT = G * WT;
    const int nblk = (n + 63) / 64, nw32 = (n + 31) / 32;
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    unsigned gen = 0;
    __shared__ double s_tot[WNW];
    // assigned bits (block minima, dirty and dense bits, the control block: initialised by the host)
    for (int c0 = gw * 64; c0 < nw32 * 32; c0 += W * 64) {
        const int cc = c0 + lane;
        const uint64_t m = __ballot(cc < n && a.colsol[cc] >= 0);
        if (lane == 0) { st_sc1(a.gasg + (c0 >> 5), (uint32_t)m); if ((c0 >> 5) + 1 < nw32) st_sc1(a.gasg + (c0 >> 5) + 1, (uint32_t)(m >> 32)); }
    }
    if (gtid == 0) st_sc1(&c->T, ~0ull);
    mc_barrier(c, G, gen);

    auto is_asg = [&](int col) -> bool { return (ld_sc1(a.gasg + (col >> 5)) >> (col & 31)) & 1u; };
    auto is_dense = [&](int i) -> bool { return (ld_sc1(a.gdense + (i >> 5)) >> (i & 31)) & 1u; };
    auto offer = [&](int col, unsigned long long lv, int row) -> unsigned long long {
        return atomicMin(a.label + col, lkey(lv, (uint32_t)row));
    };
    auto after_offer = [&](int col, unsigned long long lv, int row, unsigned long long old) {
        const unsigned long long key = lkey(lv, (uint32_t)row);
        if (key < old) {
            if (old == ~0ull) st_sc1(a.touched + atomicAdd(&c->ntouch, 1), col);
            if (lv_of(old) > lv) {                              // the label value itself dropped (not only the row of a tie)
                const unsigned long long ck = lkey(lv, (uint32_t)col);
                if (is_asg(col)) {
                    atomicOr(a.gbmin + (int64_t)(col >> 6) * MC_REC + 1, 1ull << (col & 63));
                    MC_WAIT_VM();                                   // the dirty bit is in place before the block minimum says so
                    atomicMin(a.gbmin + (int64_t)(col >> 6) * MC_REC, ck);
                } else atomicMin(&c->T, ck);
            }
        }
    };
    auto relax_to = [&](int col, unsigned long long lv, int row) { after_offer(col, lv, row, offer(col, lv, row)); };

    long long c_proc = 0, c_dense = 0, c_trivial = 0, c_hops = 0, c_macro = 0, c_verify = 0;
    int f = 0, par = 0;
    for (;;) {
        // ---- searches that end at the free row's own best column: workgroup 0, wave 0 ----
        if (g == 0 && w == 0) {
            while (f < numfree) {
                const int fr = a.freerows[f];
                const uint32_t col = a.cache_col[(int64_t)fr * KC + lane];
                const float val = a.cache_val[(int64_t)fr * KC + lane];
                const float tau = rdlane(val, KCU);
                const bool valid = lane < KCU && col != COLSENT;
                const uint32_t od = valid ? f2ord(val - ld_sc1(a.v + (valid ? col : 0))) : 0xFFFFFFFFu;
                const uint32_t omin = wave_min_u32(od);
                const bool un = valid && od == omin && !is_asg((int)col);
                const uint64_t mu = __ballot(un);
                if (!(omin != 0xFFFFFFFFu && mu && tau > ord2f(omin))) break;
                const int l = __ffsll((unsigned long long)mu) - 1;
                if (lane == l) {
                    st_sc1(a.rowsol + fr, (int32_t)col); st_sc1(a.colsol + col, (int32_t)fr); st_sc1(a.cassign + col, val);
                    atomicOr(a.gasg + (col >> 5), 1u << (col & 31));
                }
                MC_WAIT_VM();
                c_trivial++; c_hops++;
                f++;
            }
            if (lane == 0) st_sc1(&c->f, f);
        }
        mc_barrier(c, G, gen);
        f = uni(ld_sc1(&c->f));
        if (f >= numfree || ld_sc1(&c->err)) break;
        const int fr = a.freerows[f];
        const float *__restrict__ frow = a.cost + wrow_off(a.rowmap, fr, a.ld);
        const float ftau = a.cache_val[(int64_t)fr * KC + KCU];
        if (g == 0 && w == 0) {                                    // root: d[j] = c[fr][j] - v[j] for the cached columns
            const uint32_t col = a.cache_col[(int64_t)fr * KC + lane];
            const float val = a.cache_val[(int64_t)fr * KC + lane];
            if (lane < KCU && col != COLSENT) relax_to((int)col, (unsigned long long)f2ord(val - ld_sc1(a.v + col)) << 12, fr);
        }
        if (gtid == 0) { st_sc1(&c->active, W); st_sc1(&c->progress[par], 0); }
        mc_barrier(c, G, gen);

        for (;;) {
            // ================= asynchronous rounds, until a whole macro round settles nothing =================
            for (;;) {
                bool counted = true, progressed = false;
                long long idle = 0;
                for (;;) {
                    const unsigned long long Tlv = uni(lv_of(ld_sc1(&c->T)));
                    unsigned long long mk = ~0ull;
                    for (int b = gw + W * lane; b < nblk; b += W * 64) mk = umin64(mk, ld_sc1(a.gbmin + (int64_t)b * MC_REC));
                    uint64_t pkey[AP];
                    {
#pragma unroll
                        for (int q = 0; q < AP; q++) {
                            const uint32_t dk = (uint32_t)(mk >> 32);
                            const uint32_t m = wave_min_u32(dk);
                            const uint32_t m2 = wave_min_u32(dk == m ? (uint32_t)mk : 0xFFFFFFFFu);
                            pkey[q] = m == 0xFFFFFFFFu ? KEYMAX : (((uint64_t)m << 32) | m2);
                            if (mk == pkey[q]) mk = ~0ull;
                            if (lv_of(pkey[q]) >= Tlv) pkey[q] = KEYMAX;
                        }
                    }
                    if (pkey[0] == KEYMAX) {                       // nothing to settle in this wave's blocks right now
                        if (counted) { MC_WAIT_VM(); if (lane == 0) atomicSub(&c->active, 1); counted = false; }
                        if (uni(ld_sc1(&c->active)) <= 0) break;
                        __builtin_amdgcn_s_sleep(127);              // (hundreds of idle waves polling three words must not saturate their L2 channel)
                        if (++idle > (1ll << 21)) { if (lane == 0) atomicExch(&c->err, 3); break; }
                        continue;
                    }
                    if (!counted) { if (lane == 0) atomicAdd(&c->active, 1); counted = true; }
                    if (!progressed) { if (lane == 0) st_sc1(&c->progress[par], 1); progressed = true; }
                    bool pk[AP]; int pj[AP], oi[AP];
                    unsigned long long lab[AP];
                    float ca[AP], vp[AP], val[AP];
                    uint32_t col[AP];
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        pk[q] = pkey[q] != KEYMAX;
                        pj[q] = (int)lid_of(pkey[q]);
                        if (pk[q] && lane == 0) atomicAnd(a.gbmin + (int64_t)(pj[q] >> 6) * MC_REC + 1, ~(1ull << (pj[q] & 63)));
                    }
                    MC_WAIT_VM();                                   // the bits are cleared before the labels are read
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        lab[q] = ~0ull; ca[q] = 0.0f; vp[q] = 0.0f; oi[q] = 0;
                        if (pk[q]) { lab[q] = uni(ld_sc1(a.label + pj[q])); ca[q] = uni(ld_sc1(a.cassign + pj[q])); vp[q] = uni(ld_sc1(a.v + pj[q])); oi[q] = uni(ld_sc1(a.colsol + pj[q])); }
                    }
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        col[q] = COLSENT; val[q] = 0.0f;
                        if (pk[q]) { col[q] = a.cache_col[(int64_t)oi[q] * KC + lane]; val[q] = a.cache_val[(int64_t)oi[q] * KC + lane]; }
                    }
                    unsigned long long old[AP], co[AP];
                    bool off[AP], dn[AP];
                    const unsigned long long Tnow = uni(lv_of(ld_sc1(&c->T)));
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        off[q] = false; dn[q] = false; old[q] = 0; co[q] = 0;
                        const uint32_t dord = (uint32_t)(lab[q] >> 32), kq = (uint32_t)(lab[q] >> 20) & LKMAX;
                        if (pk[q] && lv_of(lab[q]) < Tnow) {
                            c_proc++;
                            if (is_dense(oi[q])) { dn[q] = true; continue; }
                            const float h = (ca[q] - vp[q]) - ord2f(dord);
                            if (lane < KCU && col[q] != COLSENT && (int)col[q] != pj[q]) {
                                const unsigned long long lv = edge_lv(f2ord((val[q] - ld_sc1(a.v + col[q])) - h), dord, kq);
                                if (lv <= Tnow) { off[q] = true; co[q] = lv; old[q] = offer((int)col[q], lv, oi[q]); }
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        const bool better = off[q] && (lkey(co[q], (uint32_t)oi[q]) < old[q]);
                        if (__ballot(better) && better) after_offer((int)col[q], co[q], oi[q], old[q]);
                    }
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        if (!dn[q]) continue;                      // the owner's cache could not certify: its whole cost row
                        const uint32_t dord = (uint32_t)(lab[q] >> 32), kq = (uint32_t)(lab[q] >> 20) & LKMAX;
                        const float h = (ca[q] - vp[q]) - ord2f(dord);
                        const float *__restrict__ row = a.cost + wrow_off(a.rowmap, oi[q], a.ld);
                        const int pjq = pj[q], oiq = oi[q];
                        const unsigned long long Tsw = uni(lv_of(ld_sc1(&c->T)));        // (once per sweep: a stale bound only prunes less)
                        wave_row_sweep(row, n, lane, [&](int cidx, float x) {
                            const unsigned long long lv = edge_lv(f2ord((x - ld_sc1(a.v + cidx)) - h), dord, kq);
                            if (cidx != pjq && lv <= Tsw && (lkey(lv, (uint32_t)oiq) < ld_sc1(a.label + cidx))) relax_to(cidx, lv, oiq);
                        });
                        c_dense++;
                    }
                    // the blocks this wave took from: their smallest dirty column now
#pragma unroll
                    for (int q = 0; q < AP; q++)
                        if (pk[q] && lane == 0) st_sc1(a.gbmin + (int64_t)(pj[q] >> 6) * MC_REC, ~0ull);
                    MC_WAIT_VM();                                   // (also: this wave's own dirty bits and block-minimum updates are in place)
#pragma unroll
                    for (int q = 0; q < AP; q++) {
                        if (!pk[q]) continue;
                        const int b = pj[q] >> 6, cc = b * 64 + lane;
                        const bool db = cc < n && ((uni(ld_sc1(a.gbmin + (int64_t)b * MC_REC + 1)) >> lane) & 1ull);
                        if (__ballot(db)) {
                            const unsigned long long lb = db ? ld_sc1(a.label + cc) : ~0ull;
                            const uint32_t dk = (uint32_t)(lb >> 32);
                            const uint32_t m = wave_min_u32(dk);
                            const uint32_t lo2 = (db && dk == m) ? (((uint32_t)lb & 0xFFF00000u) | (uint32_t)cc) : 0xFFFFFFFFu;
                            const uint32_t m2 = wave_min_u32(lo2);
                            if (lane == 0) atomicMin(a.gbmin + (int64_t)b * MC_REC, ((unsigned long long)m << 32) | m2);
                        }
                    }
                }
                c_macro++;
                mc_barrier(c, G, gen);
                const int prog = uni(ld_sc1(&c->progress[par]));
                if (gtid == 0) { st_sc1(&c->progress[par ^ 1], 0); st_sc1(&c->active, W); }
                par ^= 1;
                mc_barrier(c, G, gen);
                if (!prog || ld_sc1(&c->err)) break;
            }
            // ================= converged: do the caches certify what was skipped? =================
            const unsigned long long Tk = ld_sc1(&c->T);
            const uint32_t Dord = (uint32_t)(Tk >> 32);
            const float D = Tk == ~0ull ? INFINITY : ord2f(Dord);
            const int nt = uni(ld_sc1(&c->ntouch));
            for (int q = gtid; q < nt; q += GT) {
                const int k = ld_sc1(a.touched + q);
                const unsigned long long lbk = ld_sc1(a.label + k);
                const uint32_t dord = (uint32_t)(lbk >> 32);
                if (lv_of(lbk) < lv_of(Tk) && is_asg(k)) {             // (every settled column: see wide_aug)
                    const int i = ld_sc1(a.colsol + k);
                    if (!is_dense(i)) {
                        const float h = (ld_sc1(a.cassign + k) - ld_sc1(a.v + k)) - ord2f(dord);
                        const float bound = a.cache_val[(int64_t)i * KC + KCU] - h;
                        if (!(bound > D)) {
                            atomicOr(a.gdense + (i >> 5), 1u << (i & 31));
                            atomicOr(a.gbmin + (int64_t)(k >> 6) * MC_REC + 1, 1ull << (k & 63));
                            MC_WAIT_VM();
                            atomicMin(a.gbmin + (int64_t)(k >> 6) * MC_REC, lkey(lv_of(lbk), (uint32_t)k));
                            atomicAdd(&c->fail, 1);
                        }
                    }
                }
            }
            if (gtid == 0 && !ld_sc1(&c->rootdense) && !(ftau > D)) { st_sc1(&c->rootdense, 1); st_sc1(&c->doroot, 1); atomicAdd(&c->fail, 1); }
            c_verify++;
            mc_barrier(c, G, gen);
            const int fail = uni(ld_sc1(&c->fail)), doroot = uni(ld_sc1(&c->doroot));
            if (doroot)
                for (int cc = gtid; cc < n; cc += GT) {
                    const unsigned long long lv = (unsigned long long)f2ord(frow[cc] - ld_sc1(a.v + cc)) << 12;
                    if (lv <= lv_of(ld_sc1(&c->T)) && (lkey(lv, (uint32_t)fr) < ld_sc1(a.label + cc))) relax_to(cc, lv, fr);
                }
            mc_barrier(c, G, gen);
            if (gtid == 0) { if (fail) st_sc1(&c->anydense, 1); st_sc1(&c->fail, 0); st_sc1(&c->doroot, 0); }
            if (!fail || ld_sc1(&c->err)) break;
            mc_barrier(c, G, gen);                                  // (the resets above are in place before anyone counts failures again)
        }

        // ---- the search has ended at T: price update, path flip, reset ----
        const unsigned long long Tk = ld_sc1(&c->T);
        if (Tk == ~0ull || ld_sc1(&c->err)) { if (gtid == 0 && Tk == ~0ull) atomicExch(&c->err, 1); break; }
        const uint32_t Dord = (uint32_t)(Tk >> 32);
        const float D = ord2f(Dord);
        const int sink = (int)lid_of(Tk);
        const int nt = uni(ld_sc1(&c->ntouch));
        const int anydense = uni(ld_sc1(&c->anydense));
        int myscans = 0;
        for (int q = gtid; q < nt; q += GT) {
            const int k = ld_sc1(a.touched + q);
            const uint32_t dord = (uint32_t)(ld_sc1(a.label + k) >> 32);
            if (dord < Dord && is_asg(k)) {
                const float vk = ld_sc1(a.v + k);
                const float nv = (vk + ord2f(dord)) - D;
                if (nv < vk) st_sc1(a.v + k, nv);
                myscans++;
            }
        }
        if (myscans) atomicAdd(&c->scans, myscans);
        if (gtid == 0) {
            int j = sink;
            for (;;) {
                const int i = (int)lid_of(ld_sc1(a.label + j));
                const int jn = ld_sc1(a.rowsol + i);
                st_sc1(a.colsol + j, (int32_t)i); st_sc1(a.rowsol + i, (int32_t)j); st_sc1(a.cassign + j, a.cost[wrow_off(a.rowmap, i, a.ld) + j]);
                c_hops++;
                if (i == fr) break;
                j = jn;
            }
            atomicOr(a.gasg + (sink >> 5), 1u << (sink & 31));
        }
        mc_barrier(c, G, gen);
        for (int q = gtid; q < nt; q += GT) {
            const int k = ld_sc1(a.touched + q);
            st_sc1(a.label + k, ~0ull);
            atomicAnd(a.gbmin + (int64_t)(k >> 6) * MC_REC + 1, ~(1ull << (k & 63)));
            st_sc1(a.gbmin + (int64_t)(k >> 6) * MC_REC, ~0ull);
        }
        if (anydense) for (int q = gtid; q < nw32; q += GT) st_sc1(a.gdense + q, 0u);
        if (gtid == 0) {
            c->c_relax += ld_sc1(&c->scans);
            st_sc1(&c->scans, 0); st_sc1(&c->T, ~0ull); st_sc1(&c->ntouch, 0); st_sc1(&c->anydense, 0); st_sc1(&c->rootdense, 0); st_sc1(&c->f, f + 1);
        }
        f++;
        mc_barrier(c, G, gen);
    }

    // ---- counters of every wave; duals and total by workgroup 0 ----
    if (lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&c->c_proc), (unsigned long long)c_proc);
        atomicAdd(reinterpret_cast<unsigned long long *>(&c->c_dense), (unsigned long long)c_dense);
        if (g == 0 && w == 0) { c->c_trivial = c_trivial; c->c_hops += c_hops; c->c_macro = c_macro; c->c_verify = c_verify; }
    }
    mc_barrier(c, G, gen);
    if (g != 0) return;
    double tot = 0.0;
    for (int i = tid; i < n; i += WT) {
        const int j = ld_sc1(a.rowsol + i);
        if (j >= 0) {
            const float cij = ld_sc1(a.cassign + j);
            a.u[i] = cij - ld_sc1(a.v + j);
            tot += (double)cij;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
    if (lane == 0) s_tot[w] = tot;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < WNW; k++) t += s_tot[k];
        *reinterpret_cast<double *>(a.misc + 8) = t;
        long long *ctr = reinterpret_cast<long long *>(a.misc + 16);
        long long *wc = reinterpret_cast<long long *>(a.misc + 160);
        ctr[C_AUG_INIT] = numfree; ctr[C_AUG_RELAX] = c->c_relax; ctr[C_AUGS] = numfree; ctr[C_HOPS] = c->c_hops;
        wc[WC_DENSE_AUG] = ld_sc1(&c->c_dense); wc[WC_AUG_ROUNDS] = c->c_macro; wc[WC_AUG_PROCESSED] = ld_sc1(&c->c_proc);
        wc[WC_TRIVIAL] = c->c_trivial; wc[WC_VERIFY_PASSES] = c->c_verify;
        if (ld_sc1(&c->err)) *reinterpret_cast<int *>(a.misc + 4) = 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// ONE-EDGE SEARCHES ON THE WHOLE CHIP (CytoSPACE's repeated spot rows: reference linear_assignment_solvers.py:61-66 expands a spot into one
// identical row per slot; at config c3 35 000 of the 50 000 rows reach the searches and every one of their searches is a single edge).
// wide_aug's wave 0 disposes of such searches one after the other -- the free row's cache certifies its minimum, a column tied at that
// minimum is unassigned: the row takes the lowest such column, no price changes -- 0.22 us per row in runs, 7.7 ms at c3.  With the
// prices fixed that loop is a SERIAL DICTATORSHIP: row after row, in free-list order, takes the first column of a fixed list (its cached
// columns tied at its minimum, by column) that nobody before it took, and the loop ends at the first row that finds none.  Deferred
// acceptance computes the same assignment in parallel: every row proposes down its list, a column keeps the EARLIEST row that ever
// proposed (an atomic min of free-list positions -- claims only improve, so a column lost to an earlier row is lost for good), a row
// that lost its column moves on; at the fixed point the rows before the first one that ran out of columns hold exactly what the serial
// loop gives them (a later row never takes anything from an earlier one), and those are committed; wide_aug starts behind them.  The
// t-th row of a run of identical rows starts at its list's t-th column (the t rows before it want the same columns and come first).
// Scratch: the machine's arrays (a.scx; the row reduction is over): claims [n], lists as lane masks [n], {next candidate, lane held} [n].
// ------------------------------------------------------------------------------------------------------------------
// the k-th set bit of m at or after ... : lane of the (k + 1)-th set bit, -1 if there are not that many
__device__ __forceinline__ int nth_set(unsigned long long m, int k) {
    for (int t = 0; t < k && m; t++) m &= m - 1;
    return m ? __ffsll((long long)m) - 1 : -1;
}

// a wave per free row: its list (the cached columns tied at the row's minimum and unassigned, as a lane mask -- cache rows are sorted by
// column; 0: the cache does not certify the minimum, or nothing is free: the serial loop would stop here) and where it starts
__global__ __launch_bounds__(HEADB) void wide_claim_lists(const WideArgs *__restrict__ batch) {
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (HEADB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (HEADB / 64);
    if (blockIdx.x == 0 && threadIdx.x == 0) { ClaimCtl *c = cl_ctl(a); c->changed = 0; c->blocked = 0x7FFFFFFF; c->rounds = 0; }
    for (int i = blockIdx.x * HEADB + threadIdx.x; i < a.n; i += gridDim.x * HEADB) cl_claim(a)[i] = 0x7FFFFFFF;
    for (int p = gw; p < numfree; p += nw) {
        const int fr = uni(a.freerows[p]);
        const uint32_t col = a.cache_col[(int64_t)fr * KC + lane];
        const float val = a.cache_val[(int64_t)fr * KC + lane];
        const float tau = rdlane(val, KCU);
        const bool valid = lane < KCU && col != COLSENT;
        const float vv = valid ? a.v[col] : 0.0f;
        const int ow = valid ? a.colsol[col] : 0;
        const uint32_t od = valid ? f2ord(val - vv) : 0xFFFFFFFFu;
        const uint32_t omin = wave_min_u32(od);
        unsigned long long mu = __ballot(valid && od == omin && ow < 0);
        if (!(omin != 0xFFFFFFFFu && tau > ord2f(omin))) mu = 0;
        if (lane == 0) {
            // rank in the run of identical rows this row belongs to (consecutive in the free list, each the copy of the row before it)
            int t = 0;
            if (a.same_prev)
                while (t < 63 && p - t - 1 >= 0 && a.same_prev[fr - t] != 0 && a.freerows[p - t - 1] == fr - t - 1) t++;
            cl_mask(a)[p] = mu;
            cl_state(a)[p] = make_int2(t, -1);                   // x: candidates of the list passed over so far; y: the lane it holds (-1: none, -2: ran out)
            if (t == 0) {                                        // the head of a run: its length, for wide_aug's loop over the rows left
                int len = 1;
                if (a.same_prev)
                    while (p + len < numfree && a.freerows[p + len] == fr + len && a.same_prev[fr + len] != 0) len++;      // (the position first: fr + len < n only while the list says so)
                for (int k = 0; k < len; k++) cl_rem(a)[p + k] = len - k;
            }
        }
    }
}

// a round: a thread per free row -- still holding its column?  else down the list to the next column no earlier row has
__global__ __launch_bounds__(HEADB) void wide_claim_round(const WideArgs *__restrict__ batch) {
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    int *claim = cl_claim(a);
    int changed = 0;
    for (int p = blockIdx.x * HEADB + threadIdx.x; p < numfree; p += gridDim.x * HEADB) {
        int2 st = cl_state(a)[p];
        if (st.y == -2) continue;
        const int fr = a.freerows[p];
        const unsigned long long m = cl_mask(a)[p];
        if (st.y >= 0) {
            if (claim[a.cache_col[(int64_t)fr * KC + st.y]] == p) continue;      // (still the earliest row that asked)
            st.x++; st.y = -1; changed = 1;
        }
        for (;;) {
            const int l = nth_set(m, st.x);
            if (l < 0) { st.y = -2; changed = 1; break; }
            const int c = (int)a.cache_col[(int64_t)fr * KC + l];
            if (claim[c] > p && atomicMin(claim + c, p) > p) { st.y = l; changed = 1; break; }
            st.x++;
        }
        cl_state(a)[p] = st;
    }
    if (__ballot(changed) && (threadIdx.x & 63) == 0) atomicAdd(&cl_ctl(a)->changed, 1);
}

// after a few rounds: did anything move?  One thread looks at every problem of the batch and reports into pinned host memory (no
// synchronisation: the driver polls) -- hs[0] = groups of rounds checked so far, hs[1] = the first group after which NO problem had
// moved (0: none yet; rounds at the fixed point change nothing, so the groups the driver queued ahead are harmless).
__global__ void wide_claim_check(const WideArgs *__restrict__ batch, int nb, int group, int *hs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int any = 0;
    for (int b = 0; b < nb; b++) {
        const WideArgs a = load_wide_args(batch, b);
        ClaimCtl *c = cl_ctl(a);
        any |= c->changed;
        c->changed = 0; c->rounds += 1;
    }
    if (!any && __hip_atomic_load(hs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0)
        __hip_atomic_store(hs + 1, group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(hs, group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the fixed point: the first row that ran out of columns ...
__global__ __launch_bounds__(HEADB) void wide_claim_blocked(const WideArgs *__restrict__ batch) {
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    int first = 0x7FFFFFFF;
    for (int p = blockIdx.x * HEADB + threadIdx.x; p < numfree; p += gridDim.x * HEADB)
        if (cl_state(a)[p].y < 0) { first = p; break; }           // (positions ascend along a thread's stride: its first is its lowest)
    if (first != 0x7FFFFFFF) atomicMin(&cl_ctl(a)->blocked, first);
}
// ... and the rows before it take their columns: the serial loop's assignments (no price changes; one hop each)
__global__ __launch_bounds__(HEADB) void wide_claim_commit(const WideArgs *__restrict__ batch) {
    const WideArgs a = load_wide_args(batch, blockIdx.y);
    const int numfree = *reinterpret_cast<const int *>(a.misc + 128);
    const int pb = min(cl_ctl(a)->blocked, numfree);
    for (int p = blockIdx.x * HEADB + threadIdx.x; p < pb; p += gridDim.x * HEADB) {
        const int fr = a.freerows[p], l = cl_state(a)[p].y;
        const int c = (int)a.cache_col[(int64_t)fr * KC + l];
        a.rowsol[fr] = c; a.colsol[c] = fr; a.cassign[c] = a.cache_val[(int64_t)fr * KC + l];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        long long *ctr = reinterpret_cast<long long *>(a.misc + 16);
        long long *wc = reinterpret_cast<long long *>(a.misc + 160);
        ctr[C_HOPS] += pb; wc[WC_TRIVIAL] += pb;
        *reinterpret_cast<int *>(a.misc + 132) = pb;             // wide_aug starts behind them
    }
}

// A few ints of pinned, device-visible host memory per driver call (the machine's launches report into it).  The blocks come from
// a small process-wide pool and go back to it when the call ends: driver threads are short-lived (batch.hip spawns them per call, the
// Python side one per device and call), so a block owned by a thread would be leaked with every call.  The pool itself is never
// freed (a process that unloads the HIP runtime at exit must not call into it from a static destructor): it holds as many blocks
// as calls were ever in flight at once.
struct PinnedPool {
    struct Block { int *p; size_t cap; int dev; hipEvent_t ev; bool pending; };     // pending: launches that write into it may still be queued (ev: behind them)
    std::mutex m;
    std::vector<Block> idle;
    static PinnedPool &get() { static PinnedPool *pool = new PinnedPool(); return *pool; }
};
struct PinnedInts {
    int *p = nullptr; size_t cap = 0;
    int dev = 0; hipEvent_t ev = nullptr; hipStream_t stream = nullptr;
    PinnedInts() = default;
    PinnedInts(const PinnedInts &) = delete;
    PinnedInts &operator=(const PinnedInts &) = delete;
    ~PinnedInts() { release(); }
    // back to the pool, usable again once everything queued on `stream` so far has run (the kernels that report into the block)
    void release() {
        if (!p) return;
        const bool pending = ev && hipEventRecord(ev, stream) == hipSuccess;
        if (!pending) (void)hipStreamSynchronize(stream);
        PinnedPool &pool = PinnedPool::get();
        try { std::lock_guard<std::mutex> lk(pool.m); pool.idle.push_back({p, cap, dev, ev, pending}); }
        catch (...) { (void)hipStreamSynchronize(stream); (void)hipHostFree(p); if (ev) (void)hipEventDestroy(ev); }
        p = nullptr; cap = 0; ev = nullptr;
    }
    int ensure(size_t count, hipStream_t used_on) {
        stream = used_on;
        if (count <= cap) return CYTO_OK;
        release();
        CYTO_HIP(hipGetDevice(&dev));
        PinnedPool &pool = PinnedPool::get();
        {
            std::lock_guard<std::mutex> lk(pool.m);
            for (size_t k = 0; k < pool.idle.size(); k++) {
                PinnedPool::Block &b = pool.idle[k];
                if (b.dev != dev || b.cap < count) continue;
                if (b.pending && hipEventQuery(b.ev) != hipSuccess) { (void)hipGetLastError(); continue; }
                p = b.p; cap = b.cap; ev = b.ev;
                pool.idle[k] = pool.idle.back(); pool.idle.pop_back();
                return CYTO_OK;
            }
        }
        const size_t want = std::max<size_t>(64, count * 2);
        CYTO_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), want * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable));
        cap = want;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }     // (no event: release() drains the stream instead)
        return CYTO_OK;
    }
};
// The driver's wait for the chip: a few hundred polls of the pinned word back to back (a launch is a few microseconds), then the
// core is yielded between polls -- a process may run one such driver per device and sub-batch, more threads than cores.
struct SpinWait {
    unsigned polls = 0;
    void reset() { polls = 0; }
    void pause() {
        // (developer knob CYTO_SPIN_POLLS: polls before the first yield; 0 = never yield)
        static const unsigned k_spin = CYTO_KNOB("CYTO_SPIN_POLLS").set ? (unsigned)std::max(0, CYTO_KNOB("CYTO_SPIN_POLLS").value) : 512u;
        if (k_spin == 0 || ++polls < k_spin) __builtin_ia32_pause();
        else std::this_thread::yield();
    }
};

size_t wide_sc_ones_bytes(int n) { return (((size_t)n + 63) & ~(size_t)63) * (2 * 8); }
size_t wide_sc_ext_bytes(int n) {
    const size_t np = ((size_t)n + 63) & ~(size_t)63;
    return np * (2 * 8 + 2 * 4 + 2 * 16 + 2 * 4);
}

int wide_launch_claims(const WideArgs *d_args, int nb, int n, hipStream_t stream, int32_t *d_sync) {
    // the one-edge searches of every problem, on the whole chip (above); rounds in groups of four, then "did anything move?" -- reported
    // into pinned host memory: the driver keeps two groups queued ahead of the one under way and never waits for the stream
    if (n < 2 || !d_sync) return CYTO_OK;
    const int bxw = std::max(1, std::min((n + HEADB / 64 - 1) / (HEADB / 64), std::max(64, 4096 / std::max(1, nb))));
    const int bxt = std::max(1, std::min((n + HEADB - 1) / HEADB, std::max(16, 1024 / std::max(1, nb))));
    PinnedInts pin;
    int rc;
    if ((rc = pin.ensure(4, stream))) return rc;
    volatile int *hs = pin.p;
    hs[0] = 0; hs[1] = 0;
    hipLaunchKernelGGL(wide_claim_lists, dim3(bxw, nb), dim3(HEADB), 0, stream, d_args);
    constexpr int kAhead = 2, kMaxGroups = 1 << 16;
    int group = 0;
    SpinWait sw;
    auto t_progress = std::chrono::steady_clock::now();
    int seen = 0;
    bool fixed = false;
    for (;;) {
        if (hs[1] != 0) { fixed = true; break; }
        const int done = hs[0];
        if (done != seen) { seen = done; t_progress = std::chrono::steady_clock::now(); sw.reset(); }
        if (group - done >= kAhead) {
            sw.pause();
            if ((sw.polls & 0xFFFF) == 0) {
                const bool drained = hipStreamQuery(stream) != hipErrorNotReady;
                if ((drained && hs[0] == done && hs[1] == 0) ||                                   // (the queue has drained and nobody reported: a launch failed)
                    std::chrono::steady_clock::now() - t_progress > std::chrono::seconds(120)) {   // (no group finished for two minutes)
                    (void)hipStreamSynchronize(stream);
                    return CYTO_ERR_INTERNAL;
                }
            }
            continue;
        }
        if (group >= kMaxGroups) break;                                   // (a row moves down its list at most 63 times: cannot happen)
        group++;
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(wide_claim_round, dim3(bxt, nb), dim3(HEADB), 0, stream, d_args);
        hipLaunchKernelGGL(wide_claim_check, dim3(1), dim3(64), 0, stream, d_args, nb, group, pin.p);
    }
    if (!fixed) { (void)hipStreamSynchronize(stream); return CYTO_ERR_INTERNAL; }      // (never commit an assignment that is not the fixed point)
    hipLaunchKernelGGL(wide_claim_blocked, dim3(bxt, nb), dim3(HEADB), 0, stream, d_args);
    hipLaunchKernelGGL(wide_claim_commit, dim3(bxt, nb), dim3(HEADB), 0, stream, d_args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;                                                       // (the pinned block: usable again behind the check kernels still queued)
}

int wide_launch_rt(const WideArgs *d_args, int nb, int n, hipStream_t stream) {
    const int blocks = std::max(1, std::min((n + RTB / 64 - 1) / (RTB / 64), 2048 / std::max(1, std::min(nb, 8))));
    hipLaunchKernelGGL(wide_rt, dim3(blocks, nb), dim3(RTB), 0, stream, d_args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

int wide_launch_arr(const WideArgs *d_args, int nb, int n, hipStream_t stream, int wipe_every, bool resume, int32_t *d_sync,
                    int (*rebuild)(void *ctx, const int32_t *flags), void *ctx, const WideArgs *direct) {
    // The phase machine on the whole chip (one launch per round), in groups of launches: after a group the driver asks which problems
    // are not through (one small read) and rebuilds the row caches of those whose floors have gone stale; then wide_arr -- one
    // workgroup per problem -- for the chain rounds of the problems that did not scale (<= 64 active rows) and the free lists.
    // resume: wide_arr's chain rounds paused for fresh row caches -- it alone picks them up
    // (wipe: launches PER WORD BUFFER between two resets of its bid words -- the words' 12-bit tag counts launches since the reset)
    const int wipe = std::max(1, std::min(wipe_every > 0 ? wipe_every : 1024, 2047));
    int rc;
    if (n >= 2 && !resume) {
        // a wave per bid while the chip has room for them (a bid is a chain of L2 round trips: what counts is how many are in flight)
        // (at most 1 024 workgroups: late in a phase a few hundred rows bid and a launch is mostly the dispatch of workgroups that find
        //  nothing to do -- gpurun_out/r05i, the phase with 256 / 512 / 1 024 / 2 048 workgroups: 20 000^2 6.64 / 6.43 / 6.44 / 7.18 ms,
        //  50 000^2 19.1 / 18.3 / 18.3 / 19.3 ms, few-cell-type 20 000^2 27.7 / 24.4 / 24.2 / 25.1 ms, a 10 000-cell chunk 12.4 / 11.5 / 11.3 / 12.3 ms)
        int bid_total = 4096;
        if (CYTO_KNOB("CYTO_BID_TOTAL").set) bid_total = std::max(64, CYTO_KNOB("CYTO_BID_TOTAL").value);        // (developer knob, read once per process: tools/exp/bid_grid_batch_ab.sh)
        int bx = std::max(1, std::min((n + HEADB / 64 - 1) / (HEADB / 64), std::min(1024, std::max(64, bid_total / std::max(1, nb)))));
        if (CYTO_KNOB("CYTO_BID_GRID").set) bx = std::max(1, std::min(bx, CYTO_KNOB("CYTO_BID_GRID").value));        // (developer knob, read once per process: tools/exp/bid_grid_ab.sh)
        const int bxr = std::max(1, std::min((n + HEADB - 1) / HEADB, 2048 / std::max(1, std::min(nb, 16))));
        // (quads of a full-row bid's sweep in flight per lane: developer knob CYTO_BID_UNROLL, tools/exp/bid_unroll_ab.sh)
        const int unroll = CYTO_KNOB("CYTO_BID_UNROLL").set ? CYTO_KNOB("CYTO_BID_UNROLL").value : SC_UNROLL;
        // (one problem: the control block and the machine's arrays are kernel arguments -- one dependent load less per launch)
        char *sc_direct = nullptr, *scx_direct = nullptr;
        if (nb == 1 && direct) { sc_direct = direct->sc; scx_direct = direct->scx; }
        using RoundK = void (*)(const WideArgs *, int, int, char *, char *, int *, int);
        RoundK roundk = sc_direct ? (unroll == 8 ? (RoundK)wide_sc_round<8, true> : unroll == 4 ? (RoundK)wide_sc_round<4, true> : unroll == 3 ? (RoundK)wide_sc_round<3, true> : (RoundK)wide_sc_round<2, true>)
                                  : (unroll == 8 ? (RoundK)wide_sc_round<8, false> : unroll == 4 ? (RoundK)wide_sc_round<4, false> : unroll == 3 ? (RoundK)wide_sc_round<3, false> : (RoundK)wide_sc_round<2, false>);
        if ((rc = set_max_dynamic_lds(reinterpret_cast<const void *>(roundk)))) return rc;
        hipLaunchKernelGGL(wide_sc_init, dim3(bxr, nb), dim3(HEADB), 0, stream, d_args);
        // The driver never waits for the chip: the launches' first thread reports into pinned host memory which launch is under way,
        // how many machines are through and who wants fresh row caches; the driver keeps a bounded number of launches queued ahead of
        // the one under way (launches of a machine that is through return at once) and stops when every machine is through.
        PinnedInts t_hs;                                              // (from the pool; usable again once the launches queued below have run)
        if ((rc = t_hs.ensure(4 + (size_t)nb, stream))) return rc;
        volatile int *hs = t_hs.p;
        for (int k = 0; k < 4 + nb; k++) hs[k] = 0;
        std::vector<int32_t> seen((size_t)nb, 0), flags((size_t)nb, 0);
        // (developer knobs: launches queued ahead -- 8 / 16 / 48 / 128: 20 000^2 6.32 / 6.39 / 6.47 / 6.57 ms, gpurun_out/r05k; up to how many
        //  bids a launch gives every bid a wave of its own -- 512 / 1 024 / 2 048 / 4 096: 50 000^2 18.2 / 18.0 / 17.9 / 22.0 ms)
        const int ahead = CYTO_KNOB("CYTO_SC_AHEAD").set ? std::max(2, CYTO_KNOB("CYTO_SC_AHEAD").value) : 16;
        const int small_max = CYTO_KNOB("CYTO_SC_SMALL").set ? CYTO_KNOB("CYTO_SC_SMALL").value : SC_SMALL;
        int L = 0;
        // a NO-PROGRESS timeout: the clock restarts whenever the launch under way changes (a slow but advancing run -- counter passes of
        // a profiler, a GPU shared by several ranks -- is not an error); before an error return the stream is drained, because the
        // launches still queued read the buffers the caller releases and report into the pinned block
        SpinWait sw;
        int seen_launch = 0;
        auto t_progress = std::chrono::steady_clock::now();
        auto fail = [&]() -> int { (void)hipStreamSynchronize(stream); return CYTO_ERR_INTERNAL; };
        for (;;) {
            if (hs[1] >= nb) break;
            const int under_way = hs[0];
            if (under_way != seen_launch) { seen_launch = under_way; t_progress = std::chrono::steady_clock::now(); sw.reset(); }
            if (L - under_way >= ahead) {
                sw.pause();
                if ((sw.polls & 0xFFFF) == 0) {
                    if (hipStreamQuery(stream) != hipErrorNotReady && hs[1] < nb && L - hs[0] >= ahead) return fail();      // (the queue has drained and nobody reported: a launch failed)
                    if (std::chrono::steady_clock::now() - t_progress > std::chrono::seconds(120)) return fail();
                }
                continue;
            }
            bool want = false;
            for (int b = 0; b < nb; b++) { const int w_ = hs[4 + b]; flags[(size_t)b] = w_ != seen[(size_t)b] ? 1 : 0; seen[(size_t)b] = w_; want = want || flags[(size_t)b]; }
            if (want && rebuild && (rc = rebuild(ctx, flags.data()))) { (void)hipStreamSynchronize(stream); return rc; }
            for (int g = 0; g < 8; g++, L++) {
                if ((L >> 1) > 0 && (L >> 1) % wipe == 0) hipLaunchKernelGGL(wide_sc_wipe, dim3(bxr, nb), dim3(HEADB), 0, stream, d_args, L);
                hipLaunchKernelGGL(roundk, dim3(bx, nb), dim3(HEADB), SC_SHARED_BYTES, stream, d_args, L, n, sc_direct, scx_direct, t_hs.p, small_max);
            }
            if (L > (1 << 22)) return fail();                          // (every phase is bounded: cannot happen)
        }
        (void)d_sync;
        hipLaunchKernelGGL(wide_sc_finish, dim3(nb), dim3(64), 0, stream, d_args, L);
        CYTO_HIP(hipGetLastError());
    }
    const bool vlds = wide_arr_vlds(n), clds = wide_arr_clds(n);
    void (*k)(const WideArgs *) = vlds ? wide_arr<true, true> : clds ? wide_arr<false, true> : wide_arr<false, false>;
    if ((rc = set_max_dynamic_lds(reinterpret_cast<const void *>(k)))) return rc;
    hipLaunchKernelGGL(k, dim3(nb), dim3(WT), wide_arr_lds_bytes(n, vlds, clds), stream, d_args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

int wide_launch_aug(const WideArgs *d_args, int nb, int n, hipStream_t stream, int mc_groups, int par_groups) {
    if (mc_groups > 0) {                                           // one problem, several workgroups (blocks 0, 8, 16 ... take part)
        hipLaunchKernelGGL(wide_aug_mc, dim3(8 * mc_groups), dim3(WT), 0, stream, d_args);
        CYTO_HIP(hipGetLastError());
        return CYTO_OK;
    }
    const bool vlds = wide_aug_vlds(n), clds = wide_aug_clds(n);
    const size_t shm = wide_aug_lds_bytes(n, vlds, clds);
    if (shm > (size_t)LDS_DYNAMIC_MAX) return CYTO_ERR_UNSUPPORTED;
    if (par_groups > 1) {                                          // one problem, several SEARCHES at once (blocks 0, 8, 16 ... take part)
        void (*k)(const WideArgs *) = vlds ? wide_aug<true, true, true> : clds ? wide_aug<false, true, true> : wide_aug<false, false, true>;
        int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(k));
        if (rc) return rc;
        // (the searches' workgroups spread over all XCDs -- each with its own labels in its own L2 -- instead of packed on one: wide_aug 9.01 ->
        //  8.84 ms at 20 000^2, 20.6 -> 20.1 at 50 000^2, gpurun_out/r05l; developer knob CYTO_PAR_STRIDE = 8: the one-XCD placement)
        const int pstride = CYTO_KNOB("CYTO_PAR_STRIDE").set ? std::max(1, CYTO_KNOB("CYTO_PAR_STRIDE").value) : 1;
        hipLaunchKernelGGL(k, dim3(pstride * par_groups), dim3(WT), shm, stream, d_args);
        CYTO_HIP(hipGetLastError());
        return CYTO_OK;
    }
    void (*k)(const WideArgs *) = vlds ? wide_aug<true, true, false> : clds ? wide_aug<false, true, false> : wide_aug<false, false, false>;
    int rc = set_max_dynamic_lds(reinterpret_cast<const void *>(k));
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(nb), dim3(WT), shm, stream, d_args);
    CYTO_HIP(hipGetLastError());
    return CYTO_OK;
}

}  // namespace cyto
