/* jv_oracle.h -- C interface of the CPU Jonker-Volgenant oracle (test infrastructure only;
 * see jv_oracle_impl.h for what it restates and for the tie-breaking contract). */
#ifndef JV_ORACLE_H
#define JV_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { JV_OK = 0, JV_ERR_BAD_ARG = 1, JV_ERR_NONFINITE = 2, JV_ERR_NOMEM = 3, JV_ERR_INTERNAL = 4 };

/* Row-scan counters: one "scan" = one pass over one full cost row (n elements).
 * bytes_JV = sizeof(T) * n * (scans_colred + scans_redtransfer + scans_arr
 *                             + scans_aug_init + scans_aug_relax)       */
typedef struct {
    int64_t scans_colred, scans_redtransfer, scans_arr, scans_aug_init, scans_aug_relax;
    int64_t augmentations, path_hops;
    int64_t free_after_colred, free_after_arr1, free_after_arr2;
    int64_t arr_budget_hit;
} jv_stats;

/* maximum number of augmenting-row-reduction steps (see jv_oracle_impl.h) */
#define JV_ARR_BUDGET(n) (1000 * (int64_t)(n) + 1000000)

/* cost: row-major n x n.  rowsol[i] = column of row i, colsol[j] = row of column j
 * (colsol is what CytoSPACE calls `y`, linear_assignment_solvers.py:38).  u, v duals. */
int jv_oracle_f32(int n, const float *cost, int32_t *rowsol, int32_t *colsol, float *u, float *v,
                  double *total_f64, float *total_T, jv_stats *st);
int jv_oracle_f64(int n, const double *cost, int32_t *rowsol, int32_t *colsol, double *u, double *v,
                  double *total_f64, double *total_T, jv_stats *st);

/* threads for the wide mode's rounds (the bids of a round are independent): jv_oracle.c */
void jv_oracle_set_threads(int n);

/* float64 with a warm start: the prices of the float32 wide solve of the narrowed matrix, every row free, then the classic
 * augmenting row reduction and augmentation in float64 (jv_oracle.c) */
int jv_oracle_warm_f64(int n, const double *cost, int32_t *rowsol, int32_t *colsol, double *u, double *v,
                       double *total_f64, double *total_T, jv_stats *st);

/* ---- wide mode (jv_oracle_impl.h, second half): Jacobi reduction transfer, Jacobi rounds of augmenting row reduction,
 * succ-clamped shortest-path augmentation.  Same optimum as the classic mode; what the HIP "wide" solver computes bit for bit.
 * max_rounds < 0: JV_WIDE_ROUNDS(n).  stop_phase: 0 = solve; 1 = return the state after reduction transfer, 2 = after the
 * row-reduction rounds (rowsol[i] = -1 for free rows, colsol[j] = -1 for unassigned columns; u[free] = 0). */
typedef struct {
    int64_t scans_redtransfer, scans_arr, scans_aug_init, scans_aug_relax;
    int64_t augmentations, path_hops;
    int64_t free_after_colred, free_after_arr;
    int64_t arr_rounds, arr_retired, arr_active_left;
    int64_t arr_scaled, arr_phases, gap_exp;
} jv_wide_stats;

#define JV_WIDE_ROUNDS(n) (4096 + (int64_t)(n) / 4)
/* the eps-scaled row reduction (jv_oracle_impl.h, WIDE MODE): eps = 0 rounds before the decision, the active-list length at which a
 * phase ends (and below which an instance never scales), phases at most, rounds per phase at most, eps_0 = 2^EMULT x the median gap's
 * binade, eps_k = eps_0 / 2^(ESTEP k) */
#define JV_WIDE_K0 8
#ifndef JV_WIDE_STOP_CAP
#define JV_WIDE_STOP_CAP 64
#endif
#define JV_WIDE_STOP(n) ((n) / 128 < 8 ? 8 : ((n) / 128 > JV_WIDE_STOP_CAP ? JV_WIDE_STOP_CAP : (n) / 128))
#define JV_WIDE_STOP_FINAL(n) (JV_WIDE_STOP(n) < 16 ? JV_WIDE_STOP(n) : 16)   /* the final eps = 0 phase goes on a little longer: what it leaves are searches */
#define JV_WIDE_NPH 16
#define JV_WIDE_PHCAP 1024
#define JV_WIDE_EMULT 3
#define JV_WIDE_ESTEP 1
#define JV_WIDE_KMAX 4095            /* tight hops a label counts before the distance itself is stepped */

int jv_oracle_wide_f32(int n, const float *cost, int32_t *rowsol, int32_t *colsol, float *u, float *v,
                       double *total_f64, float *total_T, jv_wide_stats *st, int64_t max_rounds, int stop_phase);
int jv_oracle_wide_f64(int n, const double *cost, int32_t *rowsol, int32_t *colsol, double *u, double *v,
                       double *total_f64, double *total_T, jv_wide_stats *st, int64_t max_rounds, int stop_phase);

#ifdef __cplusplus
}
#endif
#endif
