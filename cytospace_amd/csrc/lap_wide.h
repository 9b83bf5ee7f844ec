// lap_wide.h -- host interface of the wide solver (lap_wide.hip) towards the float32 driver in lap_jv.hip.
#pragma once
#include "cyto_common.h"

namespace cyto {

// One problem of a batch.  Every pointer is device memory; the work arrays are the driver's (lap_jv.hip: F32Job).
//   v, u, cassign   [n] prices, row duals (written at the end), c[colsol[j]][j] per column
//   label           [n] 64-bit search labels (ordered distance << 32 | predecessor row); all-ones between searches
//   bid             [n] 64-bit bids of a row-reduction round (ordered price << 32 | row); all-ones between rounds
//   rowsol, colsol  [n] (-1 = free / unassigned), matches [n] columns claimed per row by the column reduction
//   act0, act1      [n] active-row lists of the row-reduction rounds;  freerows [n];  touched [n] columns labelled in a search
//   slot_j, slot_p, slot_c  [n] per active slot: the bid's column (-1 = retired), price, raw cost of that entry
//   cache_col/val   [n][64] row caches (lap_jv.hip: build_row_caches)
//   misc            256 bytes: +4 status, +8 double total, +16 long long counters[] (lap_jv.hip indices), +160.. wide counters
struct WideArgs {
    int n; int64_t ld; const float *cost; const int32_t *rowmap;
    float *v, *u, *cassign; unsigned long long *label, *bid;
    int32_t *rowsol, *colsol, *matches, *freerows, *act0, *act1, *touched, *slot_j;
    float *slot_p, *slot_c;
    uint32_t *cache_col; float *cache_val;
    char *misc;
    long long max_rounds;
};

// wide counters (long long each) at misc + 160
enum { WC_ROUNDS = 0, WC_BIDS, WC_RETIRED, WC_ACTIVE_LEFT, WC_FREE_ARR, WC_DENSE_ARR, WC_DENSE_AUG, WC_AUG_ROUNDS, WC_AUG_PROCESSED,
       WC_TRIVIAL, WC_VERIFY_PASSES, WC_N };

size_t wide_aug_lds_bytes(int n);
// phases, each one launch for the whole batch (d_args: device array of nb WideArgs)
int wide_launch_rt(const WideArgs *d_args, int nb, int n, hipStream_t stream);        // Jacobi reduction transfer (v0 snapshot in cassign)
int wide_launch_arr(const WideArgs *d_args, int nb, int n, hipStream_t stream);       // Jacobi rounds of augmenting row reduction + free list
int wide_launch_aug(const WideArgs *d_args, int nb, int n, hipStream_t stream);       // succ-clamped shortest-path augmentation, duals, total

}  // namespace cyto
