/*
 * cytohip.h -- C ABI of libcytohip.so, the MI355X (gfx950) implementation of CytoSPACE's
 * cell-to-spot linear-assignment hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an int status
 * (0 = CYTO_OK) and never throws.  Buffers are caller-owned and borrowed for the call only.
 * The reference is pure Python; the "FFI" a maintainer binds is ctypes.
 * Citations are into /root/reference/.
 */
#ifndef CYTOHIP_H
#define CYTOHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (errors of the reference it stands for are noted) ---- */
enum {
    CYTO_OK = 0,
    CYTO_ERR_BAD_ARG = 1,       /* non-square / null / negative sizes; reference: ValueError in lapjv */
    CYTO_ERR_NONFINITE = 2,     /* NaN/Inf in the cost matrix (zero-variance column in common.py:197) */
    CYTO_ERR_NOMEM = 3,         /* host or device allocation failed; reference: BrokenProcessPool/OOM */
    CYTO_ERR_INTERNAL = 4,      /* solver invariant violated */
    CYTO_ERR_HIP = 5,           /* a HIP runtime call failed; cyto_last_hip_error() has the text */
    CYTO_ERR_NO_DEVICE = 6,     /* no gfx950 device visible */
    CYTO_ERR_UNSUPPORTED = 7,   /* size outside what this build supports (n > 262144 per LAP) */
    CYTO_ERR_SHAPE = 8,         /* gene counts differ; reference: ValueError, common/common.py:191-192 */
    CYTO_ERR_PEER = 9           /* another rank of the communicator failed, aborted or did not arrive; reference: BrokenProcessPool */
};

const char *cyto_strerror(int status);
const char *cyto_last_hip_error(void);   /* thread-local text of the last failing HIP call */
const char *cyto_version(void);
/* ABI check for callers that mirror the structs below (ctypes, cgo ...): the library's own sizeof(cyto_lap_info),
 * sizeof(cyto_lap_opts), sizeof(cyto_assign_info), sizeof(cyto_chunk).  The structs carry no size member: a caller built
 * against another header version must compare these at load time (cytospace_amd/_lib.py does) instead of passing a struct
 * the library would overrun.  ABI history: INTEGRATION.md, "ABI versions". */
int cyto_abi_sizes(size_t *lap_info, size_t *lap_opts, size_t *assign_info, size_t *chunk);

/* ---- device + memory plumbing (so callers can keep the cost matrix resident in HBM) ---- */
int cyto_device_count(int *count);
int cyto_device_name(int device_id, char *buf, size_t buflen);
int cyto_malloc(void **dptr, size_t bytes, int device_id);
int cyto_free(void *dptr, int device_id);
int cyto_memcpy_h2d(void *dst, const void *src, size_t bytes, int device_id);
int cyto_memcpy_d2h(void *dst, const void *src, size_t bytes, int device_id);
int cyto_memcpy_d2d(void *dst, const void *src, size_t bytes, int device_id);
int cyto_device_synchronize(int device_id);
/* Work buffers of the solves are taken from a per-device cache of HBM blocks and go back to it afterwards (a steady
 * stream of chunk solves performs no hipMalloc / hipFree: hipFree would synchronise the whole device and serialise the
 * chunks that run side by side).  This returns every cached block to the runtime. */
int cyto_trim_device_cache(int device_id);

/* ---- A5: the LAP solve.  Replaces `_, y, _ = lapjv.lapjv(cost_scaled)`
 * (cytospace/linear_assignment_solvers/linear_assignment_solvers.py:34-40; solver imported
 * at :16-18).  Jonker-Volgenant: column reduction, reduction transfer, two augmenting row
 * reduction sweeps, shortest-augmenting-path augmentation.
 *
 *   n, ld         square problem size and leading dimension (elements) of the row-major cost
 *   cost          n x n costs; host pointer (cost_on_device = 0) or device pointer (= 1)
 *   rowsol[i]     column assigned to row i                      (host, n, out; may be NULL)
 *   colsol[j]     row assigned to column j -- CytoSPACE's `y`   (host, n, out; may be NULL)
 *   u, v          dual variables                                 (host, n, out; may be NULL)
 *   total         sum_i cost[i][rowsol[i]] accumulated in f64    (out; may be NULL)
 *   info          timing/work counters                           (out; may be NULL)
 *   stream        hipStream_t to launch on, NULL = the device's default stream
 * Results are bit-identical to oracle/jv_oracle.c for the same dtype (tests/test_lap_gpu.py).
 */
typedef struct {
    double ms_colred;        /* HIP-event time of the column-reduction kernels */
    double ms_cache;         /* HIP-event time of the row-cache build kernel (float32 path; else 0) */
    double ms_chain;         /* HIP-event time of the persistent chain kernels (RT+ARR, then augmentation) */
    double ms_total;         /* sum of the three (kernel time only: no H2D/D2H, no allocation) */
    int64_t scans_colred, scans_redtransfer, scans_arr, scans_aug_init, scans_aug_relax;
    int64_t augmentations, path_hops;
    int64_t free_after_colred, free_after_arr1, free_after_arr2;
    int64_t hbm_row_reads;   /* full cost rows the kernels actually fetched from HBM */
    int64_t dense_refreshes; /* RT/ARR steps whose row cache was exhausted (full re-scan) */
    double ms_arr;           /* float32 path: the RT+ARR kernel (jv_chain2) alone; wide solver: the row-reduction phase (wide_rt, the
                                phase machine's rounds on the whole chip, wide_arr) and the cache build for the searches behind it */
    double ms_aug;           /* float32 path: the augmentation kernel alone (wide solver: wide_aug -- plus, where the searches
                                paused for fresh caches, the rebuilds and the launches after the first) */
    int64_t aug_scans_skipped; /* augmentation scans elided as provable no-ops (duplicate rows) */
    int64_t row_groups;      /* number of runs of bitwise identical consecutive rows (== n: none) */
    int64_t aug_dense_scans; /* augmentation scans that had to read the full cost row (cache certificate failed) */
    int64_t aug_sparse_inits; /* augmentations whose initial row scan was served from the row cache */
    int64_t aug_handover;    /* -1, or the index of the search at which the cache-certified augmentation handed the
                                remaining free rows to the dense kernel (its certificates kept failing) */
    /* wide solver (cyto_lap_opts.mode = 2) */
    int64_t wide;                /* 1: solved by the wide solver */
    int64_t wide_rounds;         /* Jacobi rounds of augmenting row reduction that ran */
    int64_t wide_retired;        /* rows that left the rounds on a tie (handed to the augmentation) */
    int64_t wide_dense_arr;      /* bids whose row cache could not certify the top-2 (full row read) */
    int64_t wide_dense_aug;      /* augmentation: relaxations from the full cost row (cache certificate failed) */
    int64_t wide_aug_rounds;     /* augmentation: rounds of the speculative search (16 columns settled per round at most); schedule-dependent */
    int64_t wide_aug_settled;    /* augmentation: columns settled, re-settlements after a label improved included; schedule-dependent (results are not) */
    int64_t wide_trivial;        /* augmentation: searches that ended at the free row's own best column */
    int64_t wide_verify_passes;  /* augmentation: certificate passes (>= one per non-trivial search) */
    int64_t wide_list_rounds, wide_chain_rounds;   /* row-reduction rounds in the list / chain regime */
    double wide_ms_list, wide_ms_chain;            /* time inside wide_arr spent in them (the kernels' own 100 MHz clock) */
    double wide_ms_aug_rounds, wide_ms_aug_verify, wide_ms_aug_finish, wide_ms_aug_trivial;   /* wide_aug: search rounds, certificate
                                                      passes, price update + flip + reset, one-edge searches (one-workgroup kernel) */
    int64_t wide_arr_launches;   /* row reduction: launches of the round kernel (row-cache rebuilds in between) */
    int64_t wide_aug_launches;   /* augmentation: launches of the search kernel (row-cache rebuilds in between: cyto_lap_opts.wide_rebuild) */
    int64_t wide_scaled;         /* row reduction: 1 = the instance went through the eps-scaled phases */
    int64_t wide_phases;         /* row reduction: phases begun (scaled phases + the final eps = 0 phase) */
    int64_t wide_par_batches;    /* augmentation, several searches at once (cyto_lap_opts.wide_par): batches of searches run from one state */
    int64_t wide_par_discarded;  /* ... searches that met an earlier search of their batch and ran again in the next one */
    int64_t f64_warm;            /* float64: 1 = warm-started from the prices of the float32 wide solve of the narrowed matrix */
    double f64_warm_ms;          /* float64: kernel time of that float32 solve */
    /* float32, cyto_lap_opts.certify: the float64 certificate of the result */
    int64_t certified;           /* 1: the three fields below were computed */
    double gap_f64;              /* sum_i (u_i - min_j (c[i][j] - v[j])) with u_i = c[i][rowsol[i]] - v[rowsol[i]], every difference in
                                    float64: total - optimum <= gap_f64 (0: optimal for the float32 matrix in exact arithmetic) */
    double gap_max_f64;          /* the largest term of that sum */
    int64_t gap_rows;            /* rows with a positive term (their column loses to another by a fraction of a float32 ulp) */
    int64_t polished;            /* cyto_lap_opts.polish: 1 = the certificate left a gap and the float64 polish ran (rowsol / colsol / total and
                                    the duals -- narrowed to float32 -- are the float64 solve's; gap_f64 is the float32 result's, before it) */
    double polish_ms;            /* kernel time of the polish */
} cyto_lap_info;

int cyto_lap_f32(int n, const float *cost, int64_t ld, int cost_on_device,
                 int32_t *rowsol, int32_t *colsol, float *u, float *v, double *total,
                 cyto_lap_info *info, int device_id, void *stream);
int cyto_lap_f64(int n, const double *cost, int64_t ld, int cost_on_device,
                 int32_t *rowsol, int32_t *colsol, double *u, double *v, double *total,
                 cyto_lap_info *info, int device_id, void *stream);

/* The same solves with explicit kernel-selection options.  Within one `mode`, results never depend on them (every variant realises the same
 * search, bit for bit); they exist so that the kernels the solver picks for n > 26 624 / 32 768 and the hand-over paths can
 * be exercised on instances small enough for the CPU oracle (tests/test_lap_gpu.py).  NULL = the defaults. */
typedef struct {
    int32_t chain_variant;      /* 0: by size.  1: prices in L2 (u16 colsol in LDS), cached refresh -- what n > 26 624 uses.
                                   2: streaming dense refresh -- what n > 32 768 uses.  float64: != 0 selects the streaming chain (what n > 4 096
                                   uses), 1 with its row caches, 2 without (every scan reads its row).  3: as 2 (float64: as 1) with colsol in
                                   global memory too -- what n > 65 535 uses */
    int32_t augmentation;       /* 0: by size (cache-certified search above 5 120 columns).  1: dense register-resident search
                                   (n <= 26 624 only).  2: cache-certified search */
    int32_t no_handover;        /* 1: the cache-certified search never hands deep searches over to the dense kernel */
    int32_t inject_exceptions;  /* self-test of the rounding-exception list of the cache-certified search: treat the first k
                                   columns as exception columns (k > 64 overflows the list: certificates are abandoned) */
    int32_t group_state_global; /* 1: the duplicate-row group state (best offset / search stamp per row group) in global memory even
                                   where it fits LDS -- what large problems with thousands of row groups use */
    int32_t aux_state_global;   /* 1: the dense augmentation's per-column auxiliaries (cost of the assigned entry, owner's row group) in
                                   global memory even where they fit LDS -- what n > ~13 000 uses */
    int32_t mode;               /* 0: default.  1: the chain solver (classic Gauss-Seidel order: reduction transfer and augmenting row
                                   reduction row after row, Dijkstra one column per step -- oracle/jv_oracle_impl.h, first half).
                                   2: the wide solver (Jacobi reduction transfer, Jacobi rounds of row reduction, speculative
                                   shortest paths with (distance, tight-hop) labels -- same file, "WIDE MODE"; float32).  Both reach the same optimum;
                                   the duals and, where the optimum is not unique, the particular optimal assignment differ */
    int32_t wide_rounds;        /* wide solver: budget of row-reduction rounds.  0: 4096 + n / 4.  -1: none */
    int32_t wide_groups;        /* wide solver, one problem: workgroups that run a search together, asynchronously, with the search
                                   state in L2 (wide_aug_mc).  0 / -1: one workgroup, state in LDS (faster on everything but
                                   few-cell-type chunks).  k > 0: k (<= 32).  Results do not depend on it */
    int32_t wide_rebuild;       /* wide solver, augmentation: when the searches pause for the row caches to be rebuilt by the whole chip against
                                   the prices reached (a floor goes stale as searches lower the prices; deep-search instances otherwise
                                   fall back to full cost rows).  0: when the full-row relaxations since the last rebuild have cost what
                                   a rebuild costs.  -1: never.  k > 0: every k searches.  Results do not depend on it */
    int32_t wide_par;           /* wide solver, one problem: searches of consecutive free rows that run at once, a workgroup each, from one state and
                                   are committed in row order while their settled sets are disjoint (the rest runs again).  0: 16 for a problem
                                   of >= 2 048 rows without runs of identical rows, else one at a time.  -1: one at a time.  k > 1: k (<= 64;
                                   clamped to an eighth of the device's CUs -- the workgroups, one per CU, spread over the XCDs, meet at grid barriers and must all be resident).
                                   Results do not depend on it */
    int32_t wide_wipe;          /* wide solver, row reduction: the per-column bid words carry a 12-bit round tag relative to their last wipe;
                                   0: each of the two word buffers is wiped every 1 024 of its launches.  k > 0: every k (a self-test of the protocol at sizes the CPU
                                   oracle checks).  Results do not depend on it */
    int32_t cache_waves;        /* row-cache builder (float32): 0: by size (a wave per row, 8 or 20 waves per CU).  k > 0: k waves per CU (<= 32).
                                   -1: the workgroup-per-row builders of round 3.  Results do not depend on it */
    int32_t cache_unroll;       /* ... 16-byte quads in flight per lane of the wave builder: 0 (default: 8), 4 or 8 */
    int32_t cache_stream;       /* ... 0 / 1: the guess-free streaming selection for rows of >= 2 048 columns.  -1: a neighbouring row's floor
                                   as the guess (round 4's first form) */
    int32_t certify;            /* float32: 1 = one more pass over the matrix behind the solve proves in float64 how far from optimal the
                                   assignment can be (cyto_lap_info.gap_f64: total - optimum <= gap, rounding of the float32 duals and nothing
                                   else; ~2 ms at n = 50 000).  An instance whose optimum is unique by more than the gap has exactly the
                                   returned indices, whatever the solver's constants.  (took the first of the reserved words: same size) */
    int32_t polish;             /* float32, single problems: 1 = certify, and where the certificate cannot prove the result optimal (gap_f64 > 0)
                                   finish in float64 -- the matrix widened on the device (n^2 x 8 more bytes), the float32 prices as the start with
                                   every row free, the float64 augmenting row reduction and augmentation of the force_doubles path (~1.2 n steps):
                                   the optimum of the float32 matrix in float64 arithmetic, indices that do not depend on the float32 solver's
                                   constants.  Several times the cost of the solve: off by default.  (the second of the reserved words) */
    int32_t reserved[3];        /* must be zero */
} cyto_lap_opts;
int cyto_lap_f32_opts(int n, const float *cost, int64_t ld, int cost_on_device,
                      int32_t *rowsol, int32_t *colsol, float *u, float *v, double *total,
                      cyto_lap_info *info, int device_id, void *stream, const cyto_lap_opts *opts);
int cyto_lap_f64_opts(int n, const double *cost, int64_t ld, int cost_on_device,
                      int32_t *rowsol, int32_t *colsol, double *u, double *v, double *total,
                      cyto_lap_info *info, int device_id, void *stream, const cyto_lap_opts *opts);

/* Row indirection (first half).  calculate_cost repeats every spot row slots[s] times
 * (linear_assignment_solvers.py:63-66: `cost[location_repeat, :]`); here the cost holds every DISTINCT row once (nu x ld) and
 * rowmap[i] (host, n entries, non-decreasing like np.repeat's output, values in [0, nu)) names the stored row of LAP row i --
 * 10x less memory and streaming traffic at config c3.  Results are bit-identical to solving the materialised n x n matrix. */
int cyto_lap_f32_rowmap(int n, const float *cost_rows, int64_t ld, int nu, int cost_on_device, const int32_t *rowmap,
                        int32_t *rowsol, int32_t *colsol, float *u, float *v, double *total,
                        cyto_lap_info *info, int device_id, void *stream, const cyto_lap_opts *opts);

/* `lapjv(cost_scaled)` as the reference calls it (linear_assignment_solvers.py:38): a float64 HOST matrix solved in
 * float32.  The matrix is uploaded as it is and narrowed on the device (same rounding as numpy's astype(float32)). */
int cyto_lap_f32_from_f64(int n, const double *cost_host, int64_t ld, int32_t *rowsol, int32_t *colsol,
                          float *u, float *v, double *total, cyto_lap_info *info, int device_id, void *stream);

/* ---- A8 (one GPU): nb independent LAPs solved concurrently.  Replaces the per-chunk worker processes of
 * apply_linear_assignment (cytospace/cytospace.py:430-451) for the solver-only seam: the sequential
 * chain of one solve occupies one workgroup, so chunks run side by side (one launch per phase, a workgroup per problem).
 * n[b], cost[b], ld[b]: per problem; outputs are arrays of per-problem host pointers (each may be NULL);
 * total/info/status_out: arrays of length nb (may be NULL).  max_concurrent <= 0 -> min(nb, 256). */
int cyto_lap_batch_f32(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                       int32_t *const *rowsol, int32_t *const *colsol, float *const *u, float *const *v,
                       double *total, cyto_lap_info *info, int *status_out, int max_concurrent, int device_id);

int cyto_lap_batch_f32_opts(int nb, const int *n, const float *const *cost, const int64_t *ld, int cost_on_device,
                            int32_t *const *rowsol, int32_t *const *colsol, float *const *u, float *const *v,
                            double *total, cyto_lap_info *info, int *status_out, int max_concurrent, int device_id,
                            const cyto_lap_opts *opts);     /* opts->mode selects the solver for the whole batch (NULL: defaults) */

/* ---- A8 (several GPUs, one process each): the only collective on the path is the broadcast of the
 * shared standardised ST matrix (RCCL over xGMI).  id128: 128-byte ncclUniqueId made on one rank by
 * cyto_comm_unique_id and handed to the others by the launcher (bench/driver: torch.distributed, MPI,
 * a file ...).  The reference has no counterpart: it pickles the matrix to every worker (cytospace.py:446-451). */
int cyto_comm_unique_id(char *id128);
int cyto_comm_init(const char *id128, int rank, int nranks, int device_id, void **comm_out);
/* One PROCESS driving several devices, one host thread per rank -- what apply_linear_assignment's `number_of_processors` pool
 * (cytospace.py:430-451) becomes on a multi-GPU node without a launcher: comms_out[r] is the handle rank r's thread uses, on
 * device_ids[r].  Distinct devices: RCCL (ncclCommInitAll).  A device listed more than once ("logical" ranks: several workers
 * on one GPU): the ranks meet inside the process and the broadcast is a device-to-device / peer copy out of the root's buffer. */
int cyto_comm_init_local(int nranks, const int *device_ids, void **comms_out);
int cyto_comm_count(void *comm, int *nranks_out);      /* ranks the communicator spans (RCCL: ncclCommCount) */
int cyto_comm_kind(void *comm, int *kind_out);         /* 0: RCCL, 1: in-process (logical ranks) */
int cyto_comm_bcast_f32(void *comm, float *dev_buf, size_t count, int root, int device_id, void *stream);
/* *status := the largest status any rank brought (0: all fine).  Collective: how the ranks agree to enter or skip a data
 * collective together. */
int cyto_comm_agree(void *comm, int *status);
/* For a rank that cannot reach a collective its peers wait in (its host code failed first): in-process kind -- the waiting ranks
 * return CYTO_ERR_PEER; RCCL kind -- ncclCommAbort of this rank's communicator and, for communicators made by
 * cyto_comm_init_local (host threads of one process), of EVERY sibling's: ranks already inside a collective are released and
 * return CYTO_ERR_PEER, as does every later collective on any of the handles.  Afterwards the handles are good for
 * cyto_comm_destroy only.  cyto_comm_aborted: 1 once that has happened. */
int cyto_comm_abort(void *comm);
int cyto_comm_aborted(void *comm, int *aborted_out);
int cyto_comm_destroy(void *comm);

/* ---- element types of the expression matrices at this boundary.  Every `x_is_f64` parameter below and cyto_matrix.is_f64 carry one
 * of these (0 and 1 mean what the name says; 2 and 3 came in round 6): raw counts are small integers, and a uint16 / uint8 matrix is a
 * half / a quarter of the float32 upload -- the transform kernels widen in their loads, the numbers are the same bit for bit. */
enum { CYTO_DTYPE_F32 = 0, CYTO_DTYPE_F64 = 1, CYTO_DTYPE_U16 = 2, CYTO_DTYPE_U8 = 3 };

/* ---- A1: normalize_data (cytospace/common/common.py:142-147): nan_to_num, per-column counts per
 * million over the gene axis, log2(x + 1), nan_to_num.  x: G x C host matrix (float64 if x_is_f64
 * else float32; CYTO_DTYPE_*), out: G x C host float64.  Computed on the device in float64. */
int cyto_normalize_data(int G, int C, const void *x, int64_t ldx, int x_is_f64, double *out, int64_t ldo,
                        int device_id);

/* ---- A1 + first half of A2: per-column normalise (skipped if already_normalized) and standardise:
 *   z[g][c] = (y[g][c] - mean_c) / (std_c * sqrt(G))   (population std, like numpy's .std(0))
 * written as float32 into the DEVICE buffer z_dev (Gpad x ldz, zero padded by this call;
 * Gpad % 32 == 0 and ldz % 128 == 0 are what cyto_cost_pearson needs).  x may be a host or a
 * device pointer.  With these z the contraction of cyto_cost_pearson IS the Pearson correlation
 * of matrix_correlation_pearson (cytospace/common/common.py:190-199). */
int cyto_standardize(int G, int C, const void *x, int64_t ldx, int x_is_f64, int x_on_device,
                     int already_normalized, float *z_dev, int64_t ldz, int Gpad, int device_id, void *stream);

/* ---- second half of A2 + A3: cost = -corr on the fp32 matrix cores, each spot row written to its
 * slots[s] consecutive LAP rows in spot order (calculate_cost, lapjv/Pearson branch:
 * cytospace/linear_assignment_solvers/linear_assignment_solvers.py:42-69).
 * zst: Gpad x ldzst, zsc: Gpad x ldzsc (device); slots: host int64[S]; cost_dev: device,
 * (sum slots) x ldc float32.  gemm_ms (optional) = HIP-event time of the GEMM kernel. */
int cyto_cost_pearson(int Gpad, int S, int C, const float *zst, int64_t ldzst, const float *zsc, int64_t ldzsc,
                      const int64_t *slots, float *cost_dev, int64_t ldc, double *gemm_ms, int device_id,
                      void *stream);

/* ---- A7: the fused per-chunk path.  Replaces solve_linear_assignment_problem
 * (cytospace/cytospace.py:304-351) for solver_method "lapjv" + "Pearson_correlation": cost build and
 * JV solve on the device; mapped_spot[c] = index (into st's columns) of the spot cell c is mapped to.
 * sc: G x C, st: G x S host float64 row-major; slots: int64[S] with sum == C (square LAP).
 * ValueError-equivalents: CYTO_ERR_BAD_ARG (not square), CYTO_ERR_NONFINITE (zero-variance column). */
typedef struct {
    double ms_standardize;   /* normalise + standardise kernels incl. the H2D copies of sc and st */
    double ms_gemm;          /* HIP-event time of the MFMA cost GEMM */
    double gemm_flops;       /* 2 * Gpad * S * C */
    cyto_lap_info lap;
} cyto_assign_info;

int cyto_assign_pearson(int G, int C, int S, const double *sc, const double *st, const int64_t *slots,
                        int already_normalized, int64_t *mapped_spot, double *total_cost,
                        cyto_assign_info *info, int device_id);

/* ---- the other distance metrics of calculate_cost
 * (cytospace/linear_assignment_solvers/linear_assignment_solvers.py:53-59) through the same contraction.
 * Spearman_correlation: matrix_correlation_spearman (cytospace/common/common.py:202-215) = per-column average-tie
 * ranks (pandas rank() defaults), then the Pearson formula; Euclidean: scipy cdist(..., 'euclidean') transposed.
 * cyto_transform writes the float32 GEMM operand of one matrix: standardised values, standardised ranks, or the
 * plain values; cyto_cost_metric / cyto_assign_metric are cyto_cost_pearson / cyto_assign_pearson with a metric. */
enum { CYTO_METRIC_PEARSON = 0, CYTO_METRIC_SPEARMAN = 1, CYTO_METRIC_EUCLIDEAN = 2 };
enum { CYTO_TRANSFORM_STANDARDIZE = 0, CYTO_TRANSFORM_RANK = 1, CYTO_TRANSFORM_RAW = 2 };
int cyto_transform(int transform, int G, int C, const void *x, int64_t ldx, int x_is_f64, int x_on_device,
                   int already_normalized, float *z_dev, int64_t ldz, int Gpad, int device_id, void *stream);
int cyto_cost_metric(int metric, int Gpad, int S, int C, const float *zst, int64_t ldzst, const float *zsc, int64_t ldzsc,
                     const int64_t *slots, float *cost_dev, int64_t ldc, double *gemm_ms, int device_id, void *stream);
int cyto_assign_metric(int metric, int G, int C, int S, const double *sc, const double *st, const int64_t *slots,
                       int already_normalized, int64_t *mapped_spot, double *total_cost,
                       cyto_assign_info *info, int device_id);


/* ---- A8, the multi-chunk seam: apply_linear_assignment (cytospace/cytospace.py:354-469) normalises the two
 * matrices once and submits one solve_linear_assignment_problem per chunk, each with a column subset of the
 * scRNA matrix and either a column subset of the ST matrix (--single-cell, :434-435) or the full ST matrix with
 * per-chunk slot counts (--sampling-sub-spots, :438-439).  Here the matrices are uploaded and transformed once
 * into a device-resident context; a chunk gathers its columns out of it (spots with slots == 0 are not
 * contracted).  mapped_spot[c] = position in the chunk's spot list, as the reference's per-chunk result.
 * cyto_ctx_assign_chunk may be called concurrently from several host threads on one context. */
typedef struct cyto_expr_ctx cyto_expr_ctx;
int cyto_ctx_create(int metric, int G, int C, int S, const double *sc, const double *st, int already_normalized,
                    int device_id, cyto_expr_ctx **out);
int cyto_ctx_assign_chunk(cyto_expr_ctx *ctx, const int64_t *idx_sc, int n_sc, const int64_t *idx_st, int n_st,
                          const int64_t *slots, int64_t *mapped_spot, double *total_cost, cyto_assign_info *info);
void cyto_ctx_destroy(cyto_expr_ctx *ctx);

/* One process per GPU (cytospace.py:430-451 forks one worker per chunk and pickles the whole ST matrix to each): only rank
 * `root` holds the ST matrix; it transforms it once and the float32 operand reaches the other ranks with ONE broadcast over
 * xGMI (comm from cyto_comm_init; RCCL) -- the only collective of the path.  sc / C are THIS rank's cells only (raw counts
 * with already_normalized = 0: nothing normalised ever crosses PCIe twice).  st may be NULL on ranks != root.
 * bcast_ms (optional): HIP-event time of the broadcast. */
int cyto_ctx_create_shared(int metric, int G, int C, int S, const void *sc, const void *st, int x_is_f64, int already_normalized,
                           void *comm, int root, int rank, int device_id, cyto_expr_ctx **out, double *bcast_ms);

/* The general form: each matrix is a dense genes x columns array on the host or ALREADY ON THE DEVICE (e.g. expanded from sparse
 * counts by cyto_csc_to_dense_f32), float32 or float64, with its own leading dimension.  comm == NULL: no broadcast. */
typedef struct {
    const void *data;
    int64_t ld;              /* elements per row (>= number of columns) */
    int32_t is_f64;          /* the element type, CYTO_DTYPE_*: 0 float32, 1 float64 (what the name says), 2 uint16, 3 uint8 */
    int32_t on_device;       /* 1: device pointer */
} cyto_matrix;
int cyto_ctx_create_ex(int metric, int G, const cyto_matrix *sc, int C, const cyto_matrix *st, int S, int already_normalized,
                       void *comm, int root, int rank, int device_id, cyto_expr_ctx **out, double *bcast_ms);

/* SURVEY 8(f) rank 2, device-side staging: sparse counts as the reference reads them from a 10x MatrixMarket file
 * (scipy.io.mmread, cytospace/common/common.py:49; CSC: colptr[C + 1], rowidx[nnz], vals[nnz], host arrays) are uploaded as
 * non-zeros and expanded into the dense float32 G x ld DEVICE matrix the transforms read (zero filled here).  The reference
 * densifies on the host (common.py:57).  CYTO_ERR_BAD_ARG: malformed colptr or a row index outside [0, G). */
int cyto_csc_to_dense_f32(int G, int C, int64_t nnz, const int64_t *colptr, const int32_t *rowidx, const float *vals,
                          float *dense_dev, int64_t ld, int device_id, void *stream);

/* Every chunk of a rank in one call: per chunk the column gathers and the cost GEMM, then ALL the chunks' LAPs together (a
 * workgroup per chunk in every chain phase: cyto_lap_batch_f32).  Fields as cyto_ctx_assign_chunk's arguments; status, total_cost
 * and info are outputs.  max_concurrent bounds the chunks whose cost matrices exist at once (<= 0: min(nchunks, 64)). */
typedef struct {
    const int64_t *idx_sc; int32_t n_sc;
    const int64_t *idx_st; int32_t n_st;          /* NULL / ignored: all S spots */
    const int64_t *slots;
    int64_t *mapped_spot;                          /* out, n_sc entries */
    double total_cost;                             /* out */
    int32_t status;                                /* out */
    cyto_assign_info info;                         /* out */
} cyto_chunk;
int cyto_ctx_assign_chunks(cyto_expr_ctx *ctx, int nchunks, cyto_chunk *chunks, int max_concurrent);

/* The same two entry points for float32 host matrices (x_is_f64 = 0): half the host-to-device traffic, identical
 * results for data that is exactly representable in float32 (raw counts); the reference's own arrays are float64. */
int cyto_assign_metric_typed(int metric, int G, int C, int S, const void *sc, const void *st, int x_is_f64,
                             const int64_t *slots, int already_normalized, int64_t *mapped_spot, double *total_cost,
                             cyto_assign_info *info, int device_id);
int cyto_ctx_create_typed(int metric, int G, int C, int S, const void *sc, const void *st, int x_is_f64,
                          int already_normalized, int device_id, cyto_expr_ctx **out);
/* ... and with an element type PER MATRIX (host matrices, ld == columns): scRNA counts as uint8 beside ST counts as uint16, say --
 * what cytospace_amd.cytospace.assign_pearson passes for integer count matrices (config c3: the upload bounds the call). */
int cyto_assign_metric_ex(int metric, int G, const cyto_matrix *sc, int C, const cyto_matrix *st, int S, const int64_t *slots,
                          int already_normalized, int64_t *mapped_spot, double *total_cost, cyto_assign_info *info, int device_id);

#ifdef __cplusplus
}
#endif
#endif /* CYTOHIP_H */
