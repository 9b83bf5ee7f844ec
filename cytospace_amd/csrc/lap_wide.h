// lap_wide.h -- host interface of the wide solver (lap_wide.hip) towards the float32 driver in lap_jv.hip.
#pragma once
#include "cyto_common.h"

namespace cyto {

// One problem of a batch.  Every pointer is device memory; the work arrays are the driver's (lap_jv.hip: F32Job).
//   v, u, cassign   [n] prices, row duals (written at the end), c[colsol[j]][j] per column
//   label           [n] 64-bit search labels (ordered distance << 32 | tight-hop count << 20 | predecessor row); all-ones between searches
//   bid             [n] 64-bit bids of a row-reduction round (ordered price << 32 | row); all-ones between rounds
//   rowsol, colsol  [n] (-1 = free / unassigned), matches [n] columns claimed per row by the column reduction
//   act0, act1      [n] active-row lists of the row-reduction rounds;  freerows [n];  touched [n] columns labelled in a search
//   slot_j, slot_p, slot_c  [n] per active slot: the bid's column (-1 = retired), price, raw cost of that entry
//   cache_col/val   [n][64] row caches (lap_jv.hip: build_row_caches)
//   misc            512 bytes: +4 status, +8 double total, +16 long long counters[] (lap_jv.hip indices), +160.. wide counters,
//                   +256 phase timers ([12]: launches of wide_arr, [13] scaled?, [14] phases begun), +384 what the phase machine of the row reduction leaves for wide_arr
//   gbmin, gdirty, gasg, gdense, ctl   the multi-workgroup augmentation's shared state (global memory; wide_aug_mc): per 64-column
//                   block the smallest dirty label, dirty / assigned / dense bitmaps, a 256-byte control block (zeroed by the host)
//   mc_groups       workgroups that search one problem together (0: the one-workgroup kernel)
//   same_prev       [n] 1 = the row equals the row before it (runs of identical rows: CytoSPACE repeats a spot's row per slot), or null
//   seg_sync        shared by the launch, or null: [0] workgroups that asked for fresh caches (zeroed by the driver before every launch
//                   of wide_arr / wide_aug), [1 + b] wide_arr: 1 = problem b's rounds paused; wide_aug: searches problem b still has to run
//   sc              2 KB, zeroed by the driver: the control block of the row-reduction phase machine (lap_wide.hip: ScCtl)
//   scx             the phase machine's own arrays (wide_sc_ext_bytes(n), 256-byte aligned; the first wide_sc_ones_bytes(n) all-ones, the rest zero):
//                   lap_wide.hip: ScMem
//   par_groups, par searches of one problem that run at once on as many workgroups (0 / 1: one at a time) and their state (lap_wide.hip: ParCtl)
//   arr_waste       wide_arr: full-row bids (with their cache refresh) of one launch after which the list rounds pause (aug_seg == 0)
//   aug_seg         when a launch of wide_aug returns to the driver for fresh row caches: -1 never, k > 0 after k searches, 0 when
//                   its full-row relaxations reach aug_waste or seg_quorum workgroups of the launch have asked (misc + 132 holds the
//                   number of searches done)
// (fields through an X-macro: the kernels read the block through a mirror struct whose pointers are typed as GLOBAL, so that
//  every access is a global_* instruction -- through pointers loaded from memory it would be a FLAT one, and flat accesses
//  also count on lgkmcnt: every LDS wait would wait for the outstanding global loads too)
#define WIDE_FIELDS(P, S)                                                                                                  \
    S(int, n) S(int64_t, ld) P(const float, cost) P(const int32_t, rowmap) P(float, v) P(float, u) P(float, cassign)          \
    P(unsigned long long, label) P(unsigned long long, bid) P(int32_t, rowsol) P(int32_t, colsol) P(int32_t, matches)          \
    P(int32_t, freerows) P(int32_t, act0) P(int32_t, act1) P(int32_t, touched) P(int32_t, slot_j) P(float, slot_p)            \
    P(float, slot_c) P(uint32_t, cache_col) P(float, cache_val) P(char, misc) S(long long, max_rounds)                     \
    P(unsigned long long, gbmin) P(uint32_t, gdirty) P(uint32_t, gasg) P(uint32_t, gdense) P(char, ctl) S(int, mc_groups)  \
    P(const int32_t, same_prev) P(int32_t, seg_sync) S(int, aug_seg) S(int, aug_waste) S(int, arr_waste) S(int, seg_quorum)   \
    P(char, sc) S(int, par_groups) P(char, par) P(char, scx)
#define WIDE_F_PTR(T, name) T *name;
#define WIDE_F_VAL(T, name) T name;
struct WideArgs { WIDE_FIELDS(WIDE_F_PTR, WIDE_F_VAL) };

// wide counters (long long each) at misc + 160
enum { WC_ROUNDS = 0, WC_BIDS, WC_RETIRED, WC_ACTIVE_LEFT, WC_FREE_ARR, WC_DENSE_ARR, WC_DENSE_AUG, WC_AUG_ROUNDS, WC_AUG_PROCESSED,
       WC_TRIVIAL, WC_VERIFY_PASSES, WC_AUG_LAUNCHES, WC_N };

constexpr size_t WIDE_SC_BYTES = 2048;
size_t wide_sc_ext_bytes(int n);
size_t wide_sc_ones_bytes(int n);                                                      // the leading part of scx that starts all-ones
size_t wide_aug_lds_bytes(int n);
// phases, each one launch for the whole batch (d_args: device array of nb WideArgs)
int wide_launch_rt(const WideArgs *d_args, int nb, int n, hipStream_t stream);        // Jacobi reduction transfer (v0 snapshot in cassign)
int wide_launch_arr(const WideArgs *d_args, int nb, int n, hipStream_t stream, int wipe_every, bool resume, int32_t *d_sync,
                    int (*rebuild)(void *ctx, const int32_t *flags), void *ctx, const WideArgs *direct);   // direct: host copy of the one problem's block (nb == 1), or null   // rebuild: fresh row caches for the flagged problems   // Jacobi rounds of augmenting row reduction + free list
int wide_launch_claims(const WideArgs *d_args, int nb, int n, hipStream_t stream, int32_t *d_sync);   // the one-edge searches of problems with repeated rows, on the whole chip (before the first wide_launch_aug; one search at a time per problem only)
int wide_launch_aug(const WideArgs *d_args, int nb, int n, hipStream_t stream, int mc_groups, int par_groups);   // shortest-path augmentation, duals, total
size_t wide_par_state_bytes(int n, int G);                                              // control block, per-search labels / lists, claim words, change logs
constexpr int WIDE_PAR_GMAX = 64;
int wide_mc_groups(int nb, int n);                                                     // how many workgroups search one problem together (0: one)
size_t wide_mc_state_bytes(int n);                                                     // gbmin + 3 bitmaps + control block

}  // namespace cyto
