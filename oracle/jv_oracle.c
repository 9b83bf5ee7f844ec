/* jv_oracle.c -- CPU Jonker-Volgenant oracle, float and double instantiations.
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 * PARITY UNPINNED vs lapjv==1.3.14 (package absent); pinned vs scipy on certified-unique
 * instances.  Details: jv_oracle_impl.h. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "jv_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

/* the wide mode's bids of a round (independent of each other) run on this many threads: the binding sets it to the cores the
 * process may really use -- libgomp's default is every core of the HOST, and a container with 16 of 256 cores then spends its
 * time in 256 spinning threads */
/* debugging aid (tools/trace_aug_scans.py): when set, the classic augmentation writes 7 doubles per search
 * (scans, free row, levels, sink column, columns scanned AT the final distance, unassigned columns at the final distance, final distance) */
double *jv_oracle_trace_buf = NULL;
int jv_oracle_trace_rows = 0;
void jv_oracle_set_trace(double *buf, int rows) { jv_oracle_trace_buf = buf; jv_oracle_trace_rows = rows; }

void jv_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#define T float
#define SUFFIX f32
#define JV_T_IS_FLOAT 1
#include "jv_oracle_impl.h"
#undef T
#undef SUFFIX
#undef JV_T_IS_FLOAT

#define T double
#define SUFFIX f64
#include "jv_oracle_impl.h"
#undef T
#undef SUFFIX


/* ---- float64, warm-started (what the HIP float64 path computes by default; TEST INFRASTRUCTURE like the rest) ----
 * The precision of `lapjv(cost, force_doubles=True)` / `lap.lapjv` (linear_assignment_solvers.py:13-15, 36).  The classic float64
 * solve spends nearly all its time in the row-reduction price wars; the prices those wars converge to are known to float32
 * resolution beforehand -- from the float32 wide solve of the narrowed matrix.  So: narrow the costs to float32 (round to nearest),
 * solve that problem in wide mode, take ITS prices (converted exactly) as the start, every row free, and run the classic float64
 * augmenting row reduction (two sweeps) and augmentation from there.  Any prices with nothing assigned are a valid JV state: the
 * optimum reached is the float64 problem's. */
int jv_oracle_warm_f64(int n, const double *cost, int32_t *rowsol, int32_t *colsol, double *u, double *v,
                       double *total_f64, double *total_T, jv_stats *st) {
    if (n <= 0) return JV_ERR_BAD_ARG;
    const size_t N = (size_t)n;
    for (size_t k = 0; k < N * N; k++)
        if (!isfinite(cost[k])) return JV_ERR_NONFINITE;
    float *c32 = (float *)malloc(N * N * sizeof(float)), *u32 = (float *)malloc(N * sizeof(float)), *v32 = (float *)malloc(N * sizeof(float));
    double *v0 = (double *)malloc(N * sizeof(double));
    if (!c32 || !u32 || !v32 || !v0) { free(c32); free(u32); free(v32); free(v0); return JV_ERR_NOMEM; }
    int cold = 0;
    for (size_t k = 0; k < N * N; k++) { c32[k] = (float)cost[k]; if (!isfinite((double)c32[k])) cold = 1; }
    int rc = JV_OK;
    if (!cold) rc = jv_oracle_wide_f32(n, c32, rowsol, colsol, u32, v32, NULL, NULL, NULL, -1, 0);
    free(c32);
    if (rc == JV_OK && !cold) {
        for (size_t j = 0; j < N; j++) v0[j] = (double)v32[j];
        rc = jv_oracle_from_f64(n, cost, v0, rowsol, colsol, u, v, total_f64, total_T, st);
    } else if (rc == JV_OK || rc == JV_ERR_NONFINITE) {
        rc = jv_oracle_f64(n, cost, rowsol, colsol, u, v, total_f64, total_T, st);   /* costs beyond float32's range: the cold start */
    }
    free(u32); free(v32); free(v0);
    return rc;
}
