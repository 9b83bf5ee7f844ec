/*
 * jv_oracle_impl.h -- body of the CPU Jonker-Volgenant oracle, included twice
 * (T = float, T = double) by jv_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cytospace_amd/ may call this.
 *
 * What it restates
 * ----------------
 * The reference reaches its LAP solver through ONE call,
 *     `_, y, _ = solver(cost_scaled)`
 * (/root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:34-40,
 *  solver imported at :16-18 as `from lapjv import lapjv`, PyPI lapjv==1.3.14 per
 *  /root/reference/README.md:64-66).  That package is a third-party dependency whose
 *  source is NOT under /root/reference and which is not installed in this image, so
 *  the arithmetic below restates the PUBLISHED algorithm it implements:
 *     R. Jonker, A. Volgenant, "A shortest augmenting path algorithm for dense and
 *     sparse linear assignment problems", Computing 38 (1987) 325-340
 * with the four phases BASELINE.json's north_star names: COLUMN REDUCTION,
 * REDUCTION TRANSFER, AUGMENTING ROW REDUCTION (two sweeps), AUGMENTATION
 * (Dijkstra-like shortest augmenting path, price update, path flip).
 *
 * PARITY UNPINNED against lapjv itself (no wheel, no source, no reference test
 * holds a golden vector for this call).  It IS pinned against an independent exact
 * solver (scipy.optimize.linear_sum_assignment) on uniqueness-certified instances:
 * tests/test_oracle_cpu.py, tests/golden/gv8_lap.npz.
 *
 * Tie-breaking contract (what "bit-exact" means for the HIP kernels)
 * -----------------------------------------------------------------
 * Every per-element operation is a subtract or a compare, evaluated in T with no
 * FMA contraction, in the operand order written below.  Every arg-min resolves ties
 * to the LOWEST index.  The augmentation scans, among the not-yet-scanned columns
 * with minimal distance d, an unassigned column first (terminating the search),
 * otherwise the lowest-index one.  (lapjv keeps a `collist` permutation and its AVX2
 * lanes; that order is not reproducible from the publication and only matters when
 * several columns tie EXACTLY; the optimal permutation of an instance with a unique
 * optimum does not depend on it.)
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* lexicographic (value, index) two-smallest scan of h[j] = c[j] - v[j], j in [0,n)
 * == the scalar loop of JV's augmenting row reduction, ties to the lowest index. */
static void FN(top2_)(int n, const T *restrict c, const T *restrict v,
                      T *umin_o, int *j1_o, T *usub_o, int *j2_o) {
    T umin = c[0] - v[0];
    int j1 = 0;
    T usub = (T)INFINITY;
    int j2 = -1;
    for (int j = 1; j < n; j++) {
        T h = c[j] - v[j];
        if (h < usub) {
            if (h >= umin) { usub = h; j2 = j; }
            else { usub = umin; umin = h; j2 = j1; j1 = j; }
        }
    }
    *umin_o = umin; *j1_o = j1; *usub_o = usub; *j2_o = j2;
}

/* v_init != NULL: the WARM start -- prices given (any prices are a valid JV state as long as nothing is assigned), every row free:
 * no column reduction, no reduction transfer; augmenting row reduction and augmentation as ever.  (jv_oracle_warm_f64: the prices
 * of the float32 wide solve of the narrowed matrix.) */
static int FN(jv_oracle_from_)(int n, const T *restrict cost, const T *restrict v_init, int32_t *restrict rowsol,
                   int32_t *restrict colsol, T *restrict u, T *restrict v,
                   double *total_f64, T *total_T, jv_stats *st) {
    jv_stats s;
    memset(&s, 0, sizeof s);
    if (n <= 0) return JV_ERR_BAD_ARG;
    const size_t N = (size_t)n;
    for (size_t k = 0; k < N * N; k++)
        if (!isfinite((double)cost[k])) return JV_ERR_NONFINITE;

    int32_t *freerows = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *matches = (int32_t *)calloc(N, sizeof(int32_t));
    int32_t *pred = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *imin = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *lvl = (int32_t *)malloc(N * sizeof(int32_t));
    uint8_t *scanned = (uint8_t *)malloc(N);
    T *d = (T *)malloc(N * sizeof(T));
    if (!freerows || !matches || !pred || !imin || !lvl || !scanned || !d) return JV_ERR_NOMEM;

    /* ---- COLUMN REDUCTION: v[j] = min_i c[i][j] (lowest i on ties); columns are
     * claimed from the last to the first, the first claim of a row wins. ---- */
    if (v_init) {
        for (int j = 0; j < n; j++) { v[j] = v_init[j]; colsol[j] = -1; }
        for (int i = 0; i < n; i++) rowsol[i] = -1;
    } else {
    for (int j = 0; j < n; j++) { v[j] = cost[j]; imin[j] = 0; }
    for (int i = 1; i < n; i++) {
        const T *restrict ci = cost + (size_t)i * N;
        for (int j = 0; j < n; j++)
            if (ci[j] < v[j]) { v[j] = ci[j]; imin[j] = i; }
    }
    for (int i = 0; i < n; i++) rowsol[i] = -1;
    for (int j = n - 1; j >= 0; j--) {
        int i = imin[j];
        if (++matches[i] == 1) { rowsol[i] = j; colsol[j] = i; }
        else colsol[j] = -1;
    }
    }
    s.scans_colred = n;

    /* ---- REDUCTION TRANSFER (rows in ascending order; later rows see the prices
     * lowered by earlier ones). n == 1 has no other column to transfer from. ---- */
    int numfree = 0;
    for (int i = 0; i < n; i++) {
        if (matches[i] == 0) freerows[numfree++] = i;
        else if (matches[i] == 1 && n > 1) {
            const int j1 = rowsol[i];
            const T *restrict ci = cost + (size_t)i * N;
            T mn = (T)INFINITY;
            for (int j = 0; j < n; j++) {
                T h = ci[j] - v[j];
                if (j != j1 && h < mn) mn = h;
            }
            v[j1] = v[j1] - mn;
            s.scans_redtransfer++;
        }
    }
    s.free_after_colred = numfree;

    /* ---- AUGMENTING ROW REDUCTION, two sweeps ----
     * Step budget (JV_ARR_BUDGET): ARR is an initialisation heuristic that can be cut short at
     * any point without affecting optimality.  With near-tied rows (e.g. duplicated spot rows
     * whose ties were broken by a 1e-16 perturbation, solved in float64) it degenerates into a
     * price war of ~1e14 one-ulp steps; after the budget the rows still waiting are handed to
     * the augmentation phase in list order.  The HIP kernels apply the identical rule. */
    const int64_t arr_budget = JV_ARR_BUDGET(n);
    for (int sweep = 0; sweep < 2; sweep++) {
        int k = 0;
        const int prevnumfree = numfree;
        numfree = 0;
        while (k < prevnumfree) {
            if (s.scans_arr >= arr_budget) {
                while (k < prevnumfree) freerows[numfree++] = freerows[k++];
                s.arr_budget_hit = 1;
                break;
            }
            const int i = freerows[k++];
            T umin, usub; int j1, j2;
            FN(top2_)(n, cost + (size_t)i * N, v, &umin, &j1, &usub, &j2);
            s.scans_arr++;
            int i0 = colsol[j1];
            const T vj1_new = v[j1] - (usub - umin);
            const int lowers = vj1_new < v[j1];
            if (lowers) v[j1] = vj1_new;
            else if (i0 >= 0) { j1 = j2; i0 = colsol[j2]; }
            rowsol[i] = j1;
            colsol[j1] = i;
            if (i0 >= 0) {
                if (lowers) freerows[--k] = i0;
                else freerows[numfree++] = i0;
            }
        }
        if (sweep == 0) s.free_after_arr1 = numfree;
    }
    s.free_after_arr2 = numfree;

    /* ---- AUGMENTATION ---- */
    for (int f = 0; f < numfree; f++) {
        const int freerow = freerows[f];
        const T *restrict cf = cost + (size_t)freerow * N;
        for (int j = 0; j < n; j++) { d[j] = cf[j] - v[j]; pred[j] = freerow; scanned[j] = 0; }
        s.scans_aug_init++;
        int level = 0, have = 0, endofpath = -1;
        T curmin = 0;
        for (;;) {
            /* pick: lexicographic min of (d, assigned?, j) over unscanned columns */
            T dmin = (T)INFINITY;
            for (int j = 0; j < n; j++) {
                T dj = scanned[j] ? (T)INFINITY : d[j];
                dmin = dj < dmin ? dj : dmin;
            }
            int jpick = -1, jfirst = -1;
            for (int j = 0; j < n; j++) {
                if (!scanned[j] && d[j] == dmin) {
                    if (jfirst < 0) jfirst = j;
                    if (colsol[j] < 0) { jpick = j; break; }
                }
            }
            if (jpick < 0) jpick = jfirst;
            if (jpick < 0) { free(freerows); free(matches); free(pred); free(imin); free(lvl); free(scanned); free(d); return JV_ERR_INTERNAL; }
            if (!have || dmin != curmin) { level++; curmin = dmin; have = 1; }
            if (colsol[jpick] < 0) { endofpath = jpick; break; }
            scanned[jpick] = 1;
            lvl[jpick] = level;
            const int i = colsol[jpick];
            const T *restrict ci = cost + (size_t)i * N;
            const T h = (ci[jpick] - v[jpick]) - curmin;
            for (int j = 0; j < n; j++) {
                const T v2 = (ci[j] - v[j]) - h;
                const int upd = (v2 < d[j]) & !scanned[j];
                d[j] = upd ? v2 : d[j];
                pred[j] = upd ? i : pred[j];
            }
            s.scans_aug_relax++;
        }
        if (jv_oracle_trace_buf && f < jv_oracle_trace_rows) {
            double *t = jv_oracle_trace_buf + 7 * (size_t)f;
            int at = 0, un = 0, sc = 0;
            for (int j = 0; j < n; j++) { sc += scanned[j]; at += scanned[j] && d[j] == curmin; un += !scanned[j] && colsol[j] < 0 && d[j] == curmin; }
            t[0] = sc; t[1] = freerow; t[2] = level; t[3] = endofpath; t[4] = at; t[5] = un; t[6] = (double)curmin;
        }
        /* price update: columns scanned at an earlier level than the final one */
        for (int j = 0; j < n; j++)
            if (scanned[j] && lvl[j] < level) v[j] = (v[j] + d[j]) - curmin;
        /* flip the alternating path */
        int i;
        do {
            i = pred[endofpath];
            colsol[endofpath] = i;
            const int j1 = endofpath;
            endofpath = rowsol[i];
            rowsol[i] = j1;
            s.path_hops++;
        } while (i != freerow);
        s.augmentations++;
    }

    /* ---- duals and cost ---- */
    double tot = 0.0;
    T totT = 0;
    for (int i = 0; i < n; i++) {
        const int j = rowsol[i];
        const T cij = cost[(size_t)i * N + j];
        u[i] = cij - v[j];
        totT = totT + cij;
        tot += (double)cij;
    }
    if (total_f64) *total_f64 = tot;
    if (total_T) *total_T = totT;
    if (st) *st = s;
    free(freerows); free(matches); free(pred); free(imin); free(lvl); free(scanned); free(d);
    return JV_OK;
}


int FN(jv_oracle_)(int n, const T *restrict cost, int32_t *restrict rowsol, int32_t *restrict colsol, T *restrict u, T *restrict v,
                   double *total_f64, T *total_T, jv_stats *st) {
    return FN(jv_oracle_from_)(n, cost, NULL, rowsol, colsol, u, v, total_f64, total_T, st);
}


/* =====================================================================================================
 * WIDE MODE -- the same four JV phases in a form whose every phase has width (what the HIP "wide" solver
 * computes; TEST INFRASTRUCTURE like the rest of this file).
 *
 * The classic order above is a Gauss-Seidel chain: reduction transfer and augmenting row reduction visit
 * one row after the other and every step sees the prices the previous one left; the Dijkstra search settles
 * one column per step.  Nothing in the METHOD needs that order -- any prices v with "every assigned row sits
 * on a minimum of its reduced costs c[i][.] - v[.]" are a valid state for the augmentation phase, and the
 * optimum reached is the same one (unique optimum => identical indices).  Wide mode fixes an order-free
 * definition of each phase, so that a massively parallel schedule and this serial restatement produce the
 * same bits:
 *
 *  COLUMN REDUCTION     as above.
 *  REDUCTION TRANSFER   Jacobi: every row that owns exactly one column computes its margin against the
 *                       post-column-reduction prices v0 (all rows read the same snapshot), then all margins
 *                       are subtracted.  (Other prices can only have dropped too, so the margin is still a
 *                       lower bound of the true second-best: the owned column stays a row minimum.)
 *  AUGMENTING ROW REDUCTION  Jacobi rounds of an eps = 0 auction.  In a round EVERY active free row takes the
 *                       lexicographic top-2 (value, column) of its reduced costs against the round's price
 *                       snapshot and bids p = v[j1] - (u2 - u1) for its best column j1; if that does not
 *                       lower the price (a tie) it may only claim an UNASSIGNED column at its current price
 *                       (j1, else j2 when u2 == u1), otherwise it retires to the augmentation phase.  Per
 *                       column the lowest (price, row) wins: price, owner and the displaced owner (active in
 *                       the next round) change together; losers stay active.  Stops when no row is active
 *                       or after JV_WIDE_ROUNDS(n) rounds.  A round is a pure function of the state: the
 *                       order in which rows are visited cannot matter.
 *  AUGMENTATION         free rows in ascending order, each by a shortest-path search whose labels are the
 *                       UNIQUE fixed point of a monotone system, so that any label-correcting schedule
 *                       (speculative, parallel) and Dijkstra's order below agree bit for bit.  A label is a pair
 *                       (distance d, tight-hop count k), ordered lexicographically:
 *                         root f:               (fl(c[f][j] - v[j]), 0)
 *                         column jp with label (dp, kp), owner i, h = fl(fl(c[i][jp] - v[jp]) - dp),
 *                         raw = fl(fl(c[i][j] - v[j]) - h):
 *                                               raw > dp  ->  (raw, 0)
 *                                               else      ->  (dp, kp + 1)      [a tight edge: the distance does not grow]
 *                       (kp = JV_WIDE_KMAX: (succ(dp), 0) instead, succ = next representable value).  Every label is
 *                       strictly larger than its predecessor's, the edge functions are monotone: ONE fixed point,
 *                       equal labels cannot influence each other, predecessor chains are acyclic even on duplicated
 *                       rows, where tight cycles are the rule -- and the DISTANCES carry no bias: the rounding anomaly
 *                       raw < dp is clamped to dp, nothing is added (a first version clamped to succ(dp): one ulp per
 *                       tight hop, and searches run along thousands of tight edges the earlier price updates left --
 *                       at n = 70 000 the sum was enough to miss the optimum by 2e-8).  pred[j] = the LOWEST row among
 *                       those attaining the label.  The search ends at the unassigned column with the smallest
 *                       (label, column); columns with a distance < that distance get the classic price update,
 *                       clamped so that a price never rises: v[k] = min(v[k], fl(fl(v[k] + d[k]) - dist)).
 * ===================================================================================================== */
static inline T FN(succ_)(T x) {
#if defined(JV_T_IS_FLOAT)
    return nextafterf(x + 0.0f, INFINITY);
#else
    return nextafter(x + 0.0, INFINITY);
#endif
}

/* the next representable value BELOW x (x finite; +0 and -0 are one value) */
static inline T FN(pred_)(T x) {
#if defined(JV_T_IS_FLOAT)
    return nextafterf(x + 0.0f, -INFINITY);
#else
    return nextafter(x + 0.0, -INFINITY);
#endif
}
static inline int FN(expfield_)(T g) {
#if defined(JV_T_IS_FLOAT)
    uint32_t b; memcpy(&b, &g, 4); return (int)((b >> 23) & 0xFFu);
#else
    uint64_t b; memcpy(&b, &g, 8); return (int)((b >> 52) & 0x7FFu);
#endif
}
static inline T FN(pow2_)(int expfield) {       /* the power of two whose exponent field is expfield (1 <= expfield < all-ones) */
#if defined(JV_T_IS_FLOAT)
    const uint32_t b = (uint32_t)expfield << 23; T x; memcpy(&x, &b, 4); return x;
#else
    const uint64_t b = (uint64_t)expfield << 52; T x; memcpy(&x, &b, 8); return x;
#endif
}
#if defined(JV_T_IS_FLOAT)
#define JV_T_EXPMAX 255
#define JV_T_ULP1 ((T)1.1920928955078125e-07)        /* 2^-23 */
#else
#undef JV_T_EXPMAX
#undef JV_T_ULP1
#define JV_T_EXPMAX 2047
#define JV_T_ULP1 ((T)2.220446049250313e-16)         /* 2^-52 */
#endif

int FN(jv_oracle_wide_)(int n, const T *restrict cost, int32_t *restrict rowsol, int32_t *restrict colsol,
                        T *restrict u, T *restrict v, double *total_f64, T *total_T, jv_wide_stats *st,
                        int64_t max_rounds, int stop_phase) {
    jv_wide_stats s;
    memset(&s, 0, sizeof s);
    if (n <= 0) return JV_ERR_BAD_ARG;
    const size_t N = (size_t)n;
    for (size_t k = 0; k < N * N; k++)
        if (!isfinite((double)cost[k])) return JV_ERR_NONFINITE;
    if (max_rounds < 0) max_rounds = JV_WIDE_ROUNDS(n);

    int32_t *freerows = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *matches = (int32_t *)calloc(N, sizeof(int32_t));
    int32_t *pred = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *imin = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *bidrow = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *touched = (int32_t *)malloc(N * sizeof(int32_t));
    int32_t *kk = (int32_t *)malloc(N * sizeof(int32_t));
    uint8_t *scanned = (uint8_t *)malloc(N);
    uint8_t *active = (uint8_t *)calloc(N, 1);
    T *d = (T *)malloc(N * sizeof(T));
    T *bidp = (T *)malloc(N * sizeof(T));
    T *margin = (T *)malloc(N * sizeof(T));
    int32_t *alist = (int32_t *)malloc(N * sizeof(int32_t)), *abj = (int32_t *)malloc(N * sizeof(int32_t));
    T *abp = (T *)malloc(N * sizeof(T));
    int rc = JV_OK;
    if (!freerows || !matches || !pred || !imin || !bidrow || !touched || !kk || !scanned || !active || !d || !bidp || !margin || !alist || !abj || !abp) { rc = JV_ERR_NOMEM; goto done; }

    /* ---- COLUMN REDUCTION (identical to the classic mode) ---- */
    for (int j = 0; j < n; j++) { v[j] = cost[j]; imin[j] = 0; }
    for (int i = 1; i < n; i++) {
        const T *restrict ci = cost + (size_t)i * N;
        for (int j = 0; j < n; j++)
            if (ci[j] < v[j]) { v[j] = ci[j]; imin[j] = i; }
    }
    for (int i = 0; i < n; i++) rowsol[i] = -1;
    for (int j = n - 1; j >= 0; j--) {
        int i = imin[j];
        if (++matches[i] == 1) { rowsol[i] = j; colsol[j] = i; }
        else colsol[j] = -1;
    }

    /* ---- the scale of the instance, taken at the post-column-reduction prices: the median binary exponent of the rows' gaps
     * u2 - u1 (lexicographic top-2 of c[i][.] - v0[.]; bin = the exponent FIELD of the gap, 0 for a zero or subnormal gap: integer
     * counts, no visiting order), and the largest |v0[j]| ---- */
    int gap_exp = 0;
    T vmax = 0;
    if (n > 1) {
        int64_t *hist = (int64_t *)calloc(JV_T_EXPMAX + 1, sizeof(int64_t));
        if (!hist) { rc = JV_ERR_NOMEM; goto done; }
        _Pragma("omp parallel for schedule(dynamic, 16)")
        for (int i = 0; i < n; i++) {
            T umin, usub; int j1, j2;
            FN(top2_)(n, cost + (size_t)i * N, v, &umin, &j1, &usub, &j2);
            abj[i] = FN(expfield_)(usub - umin);
        }
        for (int i = 0; i < n; i++) hist[abj[i]]++;
        int64_t cum = 0;
        for (int e = 0; e <= JV_T_EXPMAX; e++) { cum += hist[e]; if (cum * 2 >= n) { gap_exp = e; break; } }
        free(hist);
        for (int j = 0; j < n; j++) { const T a = v[j] < 0 ? -v[j] : v[j]; if (a > vmax) vmax = a; }
    }
    s.gap_exp = gap_exp;

    /* ---- REDUCTION TRANSFER, Jacobi ---- */
    for (int i = 0; i < n; i++) {
        if (matches[i] == 1 && n > 1) {
            const int j1 = rowsol[i];
            const T *restrict ci = cost + (size_t)i * N;
            T mn = (T)INFINITY;
            for (int j = 0; j < n; j++) {
                T h = ci[j] - v[j];
                if (j != j1 && h < mn) mn = h;
            }
            margin[i] = mn;
            s.scans_redtransfer++;
        }
    }
    for (int i = 0; i < n; i++)
        if (matches[i] == 1 && n > 1) v[rowsol[i]] = v[rowsol[i]] - margin[i];
    int nact = 0;
    for (int i = 0; i < n; i++) if (rowsol[i] < 0) { active[i] = 1; nact++; }
    s.free_after_colred = nact;
    if (stop_phase == 1) goto finish;

    /* ---- AUGMENTING ROW REDUCTION: Jacobi rounds, eps-scaled when the instance asks for it (see the header of this mode) ---- */
    for (int j = 0; j < n; j++) bidrow[j] = -1;
    {
        int64_t total = 0, rip = 0;
        int scaled = 0;
        /* a round (eps == 0: the claim / retire rules; eps > 0: every bid lowers its column's price by gap + eps, at least one ulp) */
#define JV_WIDE_ROUND(EPS)                                                                                                 \
        do {                                                                                                               \
            const T eps_ = (EPS);                                                                                          \
            int na_ = 0;                                                                                                   \
            for (int i = 0; i < n; i++) if (active[i]) alist[na_++] = i;                                                   \
            _Pragma("omp parallel for schedule(dynamic, 16)")                                                              \
            for (int t = 0; t < na_; t++) {                                                                                \
                const int i = alist[t];                                                                                    \
                T umin, usub; int j1, j2;                                                                                  \
                FN(top2_)(n, cost + (size_t)i * N, v, &umin, &j1, &usub, &j2);                                             \
                int jt = -1; T pt = 0;                                                                                     \
                if (eps_ > 0) {                                                                                            \
                    T p = v[j1] - ((usub - umin) + eps_);                                                                  \
                    if (!(p < v[j1])) p = FN(pred_)(v[j1]);                                                                \
                    jt = j1; pt = p;                                                                                       \
                } else {                                                                                                   \
                    const T p = v[j1] - (usub - umin);                                                                     \
                    if (p < v[j1]) { jt = j1; pt = p; }                                                                    \
                    else if (colsol[j1] < 0) { jt = j1; pt = v[j1]; }                                                      \
                    else if (j2 >= 0 && usub == umin && colsol[j2] < 0) { jt = j2; pt = v[j2]; }                           \
                }                                                                                                          \
                abj[t] = jt; abp[t] = pt;                                                                                  \
            }                                                                                                              \
            int ntouched = 0;                                                                                              \
            for (int t = 0; t < na_; t++) {                                                                                \
                const int i = alist[t], jt = abj[t]; const T pt = abp[t];                                                  \
                if (jt < 0) { active[i] = 0; s.arr_retired++; continue; }                                                  \
                if (bidrow[jt] < 0) { touched[ntouched++] = jt; bidrow[jt] = i; bidp[jt] = pt; }                           \
                else if (pt < bidp[jt]) { bidrow[jt] = i; bidp[jt] = pt; }   /* rows ascend: an equal price keeps the lower row */ \
            }                                                                                                              \
            for (int t = 0; t < ntouched; t++) {                                                                           \
                const int j = touched[t], w = bidrow[j], i0 = colsol[j];                                                   \
                v[j] = bidp[j]; colsol[j] = w; rowsol[w] = j; active[w] = 0;                                               \
                if (i0 >= 0) { rowsol[i0] = -1; active[i0] = 1; }                                                          \
                bidrow[j] = -1;                                                                                            \
            }                                                                                                              \
            s.scans_arr += na_;                                                                                            \
            nact = 0;                                                                                                      \
            for (int i = 0; i < n; i++) nact += active[i];                                                                 \
            total++; rip++;                                                                                                \
        } while (0)
#define JV_WIDE_RESET()                                                                                                    \
        do {                                                                                                               \
            for (int i = 0; i < n; i++) { rowsol[i] = -1; active[i] = 1; }                                                 \
            for (int j = 0; j < n; j++) colsol[j] = -1;                                                                    \
            nact = n; rip = 0;                                                                                             \
        } while (0)

        /* the phase machine.  LEGACY: the eps = 0 rounds from the column reduction's state; after JV_WIDE_K0 of them an instance whose
         * list of active rows is still long (no ties to retire on, no end in sight: the price wars of generic costs) switches to the
         * scaled phases, any other one goes on until nobody is active.  EPS (phase k): every row unassigned, prices kept, rounds with
         * eps_k until the list is short.  FINAL: the same with eps = 0 and the claim / retire rules. */
        enum { M_LEGACY, M_EPS, M_FINAL } mode = M_LEGACY;
        int k = 0;
        T eps = 0;
        const int e0 = gap_exp > 0 ? (gap_exp + JV_WIDE_EMULT > JV_T_EXPMAX - 1 ? JV_T_EXPMAX - 1 : gap_exp + JV_WIDE_EMULT) : 0;
        const T epsmin = vmax * JV_T_ULP1;                               /* below the resolution of the prices: no such phase */
        for (;;) {
            /* the budget of rounds ends the LEGACY rounds and the scaled phases -- never the FINAL phase: the assignments a
             * scaled phase leaves satisfy eps-complementary slackness only, the augmentation needs the eps = 0 phase's */
            const int over = total >= max_rounds;
            int next_phase = 0, last = 0;
            if (mode == M_LEGACY) {
                if (over || nact == 0) break;
                if (rip == JV_WIDE_K0 && nact > JV_WIDE_STOP(n) && e0 > 0) { next_phase = 1; k = -1; }
            } else if (mode == M_EPS) {
                if (over) { next_phase = 1; last = 1; }
                else if (rip >= 1 && (nact <= JV_WIDE_STOP(n) || rip >= JV_WIDE_PHCAP)) next_phase = 1;
            } else if (rip >= 1 && (nact <= JV_WIDE_STOP_FINAL(n) || rip >= JV_WIDE_PHCAP)) break;
            if (next_phase) {
                k++;
                const int ek = e0 - JV_WIDE_ESTEP * k;
                eps = (!last && k < JV_WIDE_NPH && ek >= 1) ? FN(pow2_)(ek) : (T)0;
                if (eps < epsmin) eps = 0;
                mode = eps > 0 ? M_EPS : M_FINAL;
                JV_WIDE_RESET();
                s.arr_phases++;
                scaled = 1;
                continue;
            }
            JV_WIDE_ROUND(mode == M_EPS ? eps : (T)0);
        }
        s.arr_rounds = total;
        s.arr_scaled = scaled;
#undef JV_WIDE_ROUND
#undef JV_WIDE_RESET
    }
    s.arr_active_left = nact;
    {
        int numfree = 0;
        for (int i = 0; i < n; i++) if (rowsol[i] < 0) freerows[numfree++] = i;
        s.free_after_arr = numfree;
        if (stop_phase == 2) goto finish;

        /* ---- AUGMENTATION: shortest paths with (distance, tight-hop count) labels (see the header of this mode) ---- */
        for (int f = 0; f < numfree; f++) {
            const int freerow = freerows[f];
            const T *restrict cf = cost + (size_t)freerow * N;
            for (int j = 0; j < n; j++) { d[j] = cf[j] - v[j]; kk[j] = 0; pred[j] = freerow; scanned[j] = 0; }
            s.scans_aug_init++;
            int endofpath = -1;
            T dist = 0;
            for (;;) {
                /* pick: lexicographic min of (d, k) over the unscanned columns; among those an unassigned column first, else the lowest */
                T dmin = (T)INFINITY;
                for (int j = 0; j < n; j++) {
                    T dj = scanned[j] ? (T)INFINITY : d[j];
                    dmin = dj < dmin ? dj : dmin;
                }
                int kmin = INT32_MAX;
                for (int j = 0; j < n; j++)
                    if (!scanned[j] && d[j] == dmin && kk[j] < kmin) kmin = kk[j];
                int jpick = -1, jfirst = -1;
                for (int j = 0; j < n; j++) {
                    if (!scanned[j] && d[j] == dmin && kk[j] == kmin) {
                        if (jfirst < 0) jfirst = j;
                        if (colsol[j] < 0) { jpick = j; break; }
                    }
                }
                if (jpick < 0) jpick = jfirst;
                if (jpick < 0) { rc = JV_ERR_INTERNAL; goto done; }
                if (colsol[jpick] < 0) { endofpath = jpick; dist = dmin; break; }
                scanned[jpick] = 1;
                const int i = colsol[jpick];
                const T *restrict ci = cost + (size_t)i * N;
                const T h = (ci[jpick] - v[jpick]) - dmin;
                /* a candidate that does not exceed the label it comes from (a tight edge; rounding can even put it below) takes
                 * that label and one more tight hop; JV_WIDE_KMAX tight hops in a row: the next representable distance instead */
                const T dtight = kmin < JV_WIDE_KMAX ? dmin : FN(succ_)(dmin);
                const int ktight = kmin < JV_WIDE_KMAX ? kmin + 1 : 0;
                for (int j = 0; j < n; j++) {
                    const T raw = (ci[j] - v[j]) - h;
                    const int strict = raw > dmin;
                    const T nd = strict ? raw : dtight;
                    const int nk = strict ? 0 : ktight;
                    const int less = (nd < d[j]) | ((nd == d[j]) & (nk < kk[j]));
                    const int same = (nd == d[j]) & (nk == kk[j]);
                    const int upd = (less | (same & (i < pred[j]))) & !scanned[j];
                    d[j] = upd ? nd : d[j];
                    kk[j] = upd ? nk : kk[j];
                    pred[j] = upd ? i : pred[j];
                }
            }
            for (int j = 0; j < n; j++) s.scans_aug_relax += scanned[j] && d[j] < dist;     /* (columns settled below the final distance) */
            for (int j = 0; j < n; j++)
                if (scanned[j] && d[j] < dist) { const T nv = (v[j] + d[j]) - dist; if (nv < v[j]) v[j] = nv; }
            int i;
            do {
                i = pred[endofpath];
                colsol[endofpath] = i;
                const int j1 = endofpath;
                endofpath = rowsol[i];
                rowsol[i] = j1;
                s.path_hops++;
            } while (i != freerow);
            s.augmentations++;
        }
    }

finish:
    {
        double tot = 0.0;
        T totT = 0;
        for (int i = 0; i < n; i++) {
            const int j = rowsol[i];
            if (j < 0) { u[i] = 0; continue; }                 /* (only with stop_phase != 0) */
            const T cij = cost[(size_t)i * N + j];
            u[i] = cij - v[j];
            totT = totT + cij;
            tot += (double)cij;
        }
        if (total_f64) *total_f64 = tot;
        if (total_T) *total_T = totT;
    }
    if (st) *st = s;
done:
    free(freerows); free(matches); free(pred); free(imin); free(bidrow); free(touched); free(kk); free(scanned); free(active);
    free(d); free(bidp); free(margin); free(alist); free(abj); free(abp);
    return rc;
}

#undef FN
#undef CAT
#undef CAT_
