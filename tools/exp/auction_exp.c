/* auction_exp.c -- CPU exploration harness (NOT product, NOT oracle): how many Jacobi rounds / bids / search settlements do
 * variants of the row-reduction phase need?  usage: auction_exp cost.bin n mode [args]
 *   mode 0: current wide restatement (RT Jacobi + eps=0 rounds, budget R)            args: R
 *   mode 1: eps-scaling Jacobi auction, then eps=0 phase (rebid all), then searches   args: eps0 factor epsmin R0
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int n;
static float *cost, *v;
static int *rowsol, *colsol;

static void top2(const float *c, float *umin_o, int *j1_o, float *usub_o, int *j2_o) {
    float umin = c[0] - v[0], usub = INFINITY; int j1 = 0, j2 = -1;
    for (int j = 1; j < n; j++) {
        float h = c[j] - v[j];
        if (h < usub) { if (h >= umin) { usub = h; j2 = j; } else { usub = umin; umin = h; j2 = j1; j1 = j; } }
    }
    *umin_o = umin; *j1_o = j1; *usub_o = usub; *j2_o = j2;
}

typedef struct { long rounds, bids, wide_rounds, chain_rounds, wide_bids, chain_bids, retired; } phase_stats;

/* Jacobi rounds with increment eps (eps == 0: the retire rules of the wide restatement).  active[] in/out. */
static int stop_na = 0;
static float *Fl = NULL; static long dense_bids = 0; static int cache_k = 48;
static float *V0 = NULL;            /* NEED=1: the post-column-reduction prices (the epoch of the first cache build) */
static long need_hist[8];           /* per phase: bids whose second-best value lies below the k-th smallest ORIGINAL reduced cost, k <= 48,128,512,1024,2048,4096,8192,inf */
static int cmpf_(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return x < y ? -1 : x > y; }
static float kth_reduced(const float *c, int k) {   /* k-th smallest (0-based) of c[j]-v[j] */
    float *t = malloc(sizeof(float) * n); for (int j = 0; j < n; j++) t[j] = c[j] - v[j];
    /* quickselect via partial: simple nth by sorting a sample is wrong; do full qsort only for small n, else select */
    int lo = 0, hi = n - 1;
    while (lo < hi) { float piv = t[(lo + hi) / 2]; int i = lo, j = hi; while (i <= j) { while (t[i] < piv) i++; while (t[j] > piv) j--; if (i <= j) { float x = t[i]; t[i] = t[j]; t[j] = x; i++; j--; } } if (k <= j) hi = j; else if (k >= i) lo = i; else break; }
    float r = t[k]; free(t); return r;
}
static void rebuild_all(void) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; i++) Fl[i] = kth_reduced(cost + (size_t)i * n, cache_k);
}
static void rounds(float eps, long max_rounds, uint8_t *active, phase_stats *ps, int verbose) {
    int *list = malloc(sizeof(int) * n), *bj = malloc(sizeof(int) * n); float *bp = malloc(sizeof(float) * n);
    int *bidrow = malloc(sizeof(int) * n); float *bidp = malloc(sizeof(float) * n); int *touched = malloc(sizeof(int) * n);
    for (int j = 0; j < n; j++) bidrow[j] = -1;
    memset(ps, 0, sizeof *ps);
    long hist[8] = {0};
    for (;;) {
        int na = 0;
        for (int i = 0; i < n; i++) if (active[i]) list[na++] = i;
        if (na == 0 || ps->rounds >= max_rounds || (na <= stop_na && ps->rounds > 0)) break;
#pragma omp parallel for schedule(dynamic, 8)
        for (int t = 0; t < na; t++) {
            int i = list[t]; float umin, usub; int j1, j2;
            top2(cost + (size_t)i * n, &umin, &j1, &usub, &j2);
            int jt = -1; float pt = 0;
            if (eps > 0) {
                float p = v[j1] - ((usub - umin) + eps);
                if (!(p < v[j1])) p = nextafterf(v[j1], -INFINITY);
                jt = j1; pt = p;
            } else {
                float p = v[j1] - (usub - umin);
                if (p < v[j1]) { jt = j1; pt = p; }
                else if (colsol[j1] < 0) { jt = j1; pt = v[j1]; }
                else if (j2 >= 0 && usub == umin && colsol[j2] < 0) { jt = j2; pt = v[j2]; }
            }
            if (V0) {
                const float *c = cost + (size_t)i * n; int need = 0;
                for (int j = 0; j < n; j++) need += (c[j] - V0[j]) < usub;
                int b = need <= 48 ? 0 : need <= 128 ? 1 : need <= 512 ? 2 : need <= 1024 ? 3 : need <= 2048 ? 4 : need <= 4096 ? 5 : need <= 8192 ? 6 : 7;
#pragma omp atomic
                need_hist[b]++;
            }
            if (Fl && !(usub < Fl[i])) {
#pragma omp atomic
                dense_bids++;
                Fl[i] = kth_reduced(cost + (size_t)i * n, cache_k); }
            bj[t] = jt; bp[t] = pt;
        }
        int nt = 0;
        for (int t = 0; t < na; t++) {
            int i = list[t], jt = bj[t]; float pt = bp[t];
            if (jt < 0) { active[i] = 0; ps->retired++; continue; }
            if (bidrow[jt] < 0) { touched[nt++] = jt; bidrow[jt] = i; bidp[jt] = pt; }
            else if (pt < bidp[jt]) { bidrow[jt] = i; bidp[jt] = pt; }
        }
        for (int t = 0; t < nt; t++) {
            int j = touched[t], w = bidrow[j], i0 = colsol[j];
            v[j] = bidp[j]; colsol[j] = w; rowsol[w] = j; active[w] = 0;
            if (i0 >= 0) { rowsol[i0] = -1; active[i0] = 1; }
            bidrow[j] = -1;
        }
        ps->rounds++; ps->bids += na;
        if (na > 64) { ps->wide_rounds++; ps->wide_bids += na; } else { ps->chain_rounds++; ps->chain_bids += na; }
        int b = na <= 1 ? 0 : na <= 4 ? 1 : na <= 16 ? 2 : na <= 64 ? 3 : na <= 256 ? 4 : na <= 1024 ? 5 : na <= 4096 ? 6 : 7;
        hist[b]++;
    }
    if (verbose && V0) {
        printf("      band needed (columns whose reduced cost at v0 < the bid's second-best now): <=48:%ld <=128:%ld <=512:%ld <=1024:%ld <=2048:%ld <=4096:%ld <=8192:%ld more:%ld\n",
               need_hist[0], need_hist[1], need_hist[2], need_hist[3], need_hist[4], need_hist[5], need_hist[6], need_hist[7]);
        memset(need_hist, 0, sizeof need_hist);
    }
    if (verbose) {
        int left = 0; for (int i = 0; i < n; i++) left += active[i];
        printf("   eps=%.3e rounds=%ld bids=%ld  wide(>64): %ld rounds %ld bids | chain: %ld rounds %ld bids | retired %ld left %d\n",
               eps, ps->rounds, ps->bids, ps->wide_rounds, ps->wide_bids, ps->chain_rounds, ps->chain_bids, ps->retired, left);
        printf("      rounds by na: 1:%ld 2-4:%ld 5-16:%ld 17-64:%ld 65-256:%ld -1024:%ld -4096:%ld >4096:%ld\n",
               hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    }
    free(list); free(bj); free(bp); free(bidrow); free(bidp); free(touched);
}

static int raise_mode = 0;   /* 1: before the searches, 2: also at every phase boundary */
static void raise_unassigned(const char *tag) {
    float *u = malloc(sizeof(float) * n);
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; i++) { float umin, usub; int j1, j2; top2(cost + (size_t)i * n, &umin, &j1, &usub, &j2); u[i] = umin; }
    int nr = 0; double sum = 0; float mx = 0;
    for (int j = 0; j < n; j++) if (colsol[j] < 0) {
        float m = INFINITY; for (int i = 0; i < n; i++) { float x = cost[(size_t)i * n + j] - u[i]; if (x < m) m = x; }
        if (m > v[j]) { nr++; sum += m - v[j]; if (m - v[j] > mx) mx = m - v[j]; v[j] = m; }
    }
    printf("      raise(%s): %d columns raised, mean %.3e max %.3e\n", tag, nr, nr ? sum / nr : 0.0, mx);
    free(u);
}
/* searches: plain Dijkstra (distances only), counts */
static void searches(void) {
    if (raise_mode >= 1) raise_unassigned("end");
    float *d = malloc(sizeof(float) * n); int *pred = malloc(sizeof(int) * n); uint8_t *sc = malloc(n);
    int numfree = 0; long settled = 0, maxset = 0;
    int *fr = malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) if (rowsol[i] < 0) fr[numfree++] = i;
    double t0 = omp_get_wtime();
    for (int f = 0; f < numfree; f++) {
        int freerow = fr[f];
        const float *cf = cost + (size_t)freerow * n;
        for (int j = 0; j < n; j++) { d[j] = cf[j] - v[j]; pred[j] = freerow; sc[j] = 0; }
        int end = -1; float dist = 0; long ns = 0;
        for (;;) {
            float dmin = INFINITY; int jp = -1;
            for (int j = 0; j < n; j++) if (!sc[j] && (d[j] < dmin || (d[j] == dmin && jp >= 0 && colsol[jp] >= 0 && colsol[j] < 0))) { dmin = d[j]; jp = j; }
            if (colsol[jp] < 0) { end = jp; dist = dmin; break; }
            sc[jp] = 1; ns++;
            int i = colsol[jp]; const float *ci = cost + (size_t)i * n;
            float h = (ci[jp] - v[jp]) - dmin;
            for (int j = 0; j < n; j++) { float v2 = (ci[j] - v[j]) - h; if (v2 < dmin) v2 = dmin; if (!sc[j] && v2 < d[j]) { d[j] = v2; pred[j] = i; } }
        }
        long below = 0;
        for (int j = 0; j < n; j++) if (sc[j] && d[j] < dist) { float nv = (v[j] + d[j]) - dist; if (nv < v[j]) v[j] = nv; below++; }
        settled += ns; if (ns > maxset) maxset = ns;
        int i, e = end; long hops = 0;
        do { i = pred[e]; colsol[e] = i; int j1 = e; e = rowsol[i]; rowsol[i] = j1; hops++; } while (i != freerow);
        if (numfree <= 200 && ns > 500) printf("      search %d: row %d settled %ld hops %ld dist %.3e\n", f, freerow, ns, hops, dist);
    }
    printf("   searches: free=%d settled=%ld (max %ld) time %.1fs\n", numfree, settled, maxset, omp_get_wtime() - t0);
    double tot = 0; for (int i = 0; i < n; i++) tot += cost[(size_t)i * n + rowsol[i]];
    printf("   total = %.9f\n", tot);
    free(d); free(pred); free(sc); free(fr);
}

/* K searches from one snapshot; commit the conflict-free prefix (settled sets + sinks disjoint); count batches */
static int spec_K = 0;
static void searches_spec(void) {
    int K = spec_K;
    float *d = malloc(sizeof(float) * n); int *pred = malloc(sizeof(int) * n); uint8_t *sc = malloc(n);
    int numfree = 0; int *fr = malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) if (rowsol[i] < 0) fr[numfree++] = i;
    int *stamp = calloc(n, sizeof(int));
    /* per speculative search: settled list, d values, pred of path */
    int **setl = malloc(sizeof(int *) * K); float **setd = malloc(sizeof(float *) * K); int **pth = malloc(sizeof(int *) * K);
    for (int q = 0; q < K; q++) { setl[q] = malloc(sizeof(int) * n); setd[q] = malloc(sizeof(float) * n); pth[q] = malloc(sizeof(int) * 2 * n); }
    int *nset = malloc(sizeof(int) * K), *npth = malloc(sizeof(int) * K), *sink = malloc(sizeof(int) * K); float *dist = malloc(sizeof(float) * K);
    long batches = 0, wasted = 0, crit = 0, total_set = 0; int f = 0, batch_id = 0;
    const int keep = getenv("SPEC_KEEP") != NULL;
    /* per free row: a retained result (valid while disjoint from everything committed since it was computed) */
    int *have = calloc(numfree, sizeof(int)); int **kset = calloc(numfree, sizeof(int *)); float **ksd = calloc(numfree, sizeof(float *)); int **kpth = calloc(numfree, sizeof(int *));
    int *knset = calloc(numfree, sizeof(int)), *knpth = calloc(numfree, sizeof(int)), *ksink = calloc(numfree, sizeof(int)); float *kdist = calloc(numfree, sizeof(float));
    while (keep && f < numfree) {
        batch_id++; batches++;
        int kk = numfree - f < K ? numfree - f : K;
        long maxset = 0;
        for (int q = 0; q < kk; q++) {
            int fi = f + q;
            if (have[fi]) continue;
            int freerow = fr[fi]; const float *cf = cost + (size_t)freerow * n;
            for (int j = 0; j < n; j++) { d[j] = cf[j] - v[j]; pred[j] = freerow; sc[j] = 0; }
            int end = -1; float ds = 0; int ns = 0; int *sl = malloc(sizeof(int) * n);
            for (;;) {
                float dmin = INFINITY; int jp = -1;
                for (int j = 0; j < n; j++) if (!sc[j] && (d[j] < dmin || (d[j] == dmin && jp >= 0 && colsol[jp] >= 0 && colsol[j] < 0))) { dmin = d[j]; jp = j; }
                if (colsol[jp] < 0) { end = jp; ds = dmin; break; }
                sc[jp] = 1; sl[ns++] = jp;
                int i = colsol[jp]; const float *ci = cost + (size_t)i * n; float h = (ci[jp] - v[jp]) - dmin;
                for (int j = 0; j < n; j++) { float v2 = (ci[j] - v[j]) - h; if (v2 < dmin) v2 = dmin; if (!sc[j] && v2 < d[j]) { d[j] = v2; pred[j] = i; } }
            }
            kset[fi] = realloc(sl, sizeof(int) * (ns + 1)); ksd[fi] = malloc(sizeof(float) * (ns + 1));
            for (int t = 0; t < ns; t++) ksd[fi][t] = d[kset[fi][t]];
            knset[fi] = ns; ksink[fi] = end; kdist[fi] = ds;
            int *pp = malloc(sizeof(int) * 2 * (ns + 2)); int np = 0, e = end, i; do { i = pred[e]; pp[np++] = e; pp[np++] = i; e = rowsol[i]; } while (i != freerow);
            kpth[fi] = pp; knpth[fi] = np; have[fi] = 1;
            if (ns > maxset) maxset = ns;
        }
        crit += maxset;
        int committed = 0;
        for (int q = 0; q < kk; q++) {
            int fi = f + q;
            int conflict = stamp[ksink[fi]] == batch_id;
            for (int t = 0; t < knset[fi] && !conflict; t++) conflict = stamp[kset[fi][t]] == batch_id;
            if (conflict) break;
            for (int t = 0; t < knset[fi]; t++) { int j = kset[fi][t]; stamp[j] = batch_id; if (ksd[fi][t] < kdist[fi]) { float nv = (v[j] + ksd[fi][t]) - kdist[fi]; if (nv < v[j]) v[j] = nv; } }
            stamp[ksink[fi]] = batch_id;
            for (int t = 0; t < knpth[fi]; t += 2) { int e = kpth[fi][t], i = kpth[fi][t + 1]; colsol[e] = i; rowsol[i] = e; }
            committed++; total_set += knset[fi];
        }
        /* results computed but not committed: kept if disjoint from this batch's commits (checked for ALL retained results) */
        for (int fi = f + committed; fi < numfree; fi++) if (have[fi]) {
            int conflict = stamp[ksink[fi]] == batch_id;
            for (int t = 0; t < knset[fi] && !conflict; t++) conflict = stamp[kset[fi][t]] == batch_id;
            if (conflict) { wasted += knset[fi]; have[fi] = 0; free(kset[fi]); free(ksd[fi]); free(kpth[fi]); }
        }
        f += committed;
    }
    if (keep) {
        printf("   speculative KEEP K=%d: free=%d batches=%ld critical-path settled=%ld (sequential %ld) wasted=%ld\n", K, numfree, batches, crit, total_set, wasted);
        double tot = 0; for (int i = 0; i < n; i++) tot += cost[(size_t)i * n + rowsol[i]];
        printf("   total = %.9f\n", tot);
        return;
    }
    while (f < numfree) {
        batch_id++; batches++;
        int kk = numfree - f < K ? numfree - f : K;
        long maxset = 0;
        for (int q = 0; q < kk; q++) {
            int freerow = fr[f + q]; const float *cf = cost + (size_t)freerow * n;
            for (int j = 0; j < n; j++) { d[j] = cf[j] - v[j]; pred[j] = freerow; sc[j] = 0; }
            int end = -1; float ds = 0; int ns = 0;
            for (;;) {
                float dmin = INFINITY; int jp = -1;
                for (int j = 0; j < n; j++) if (!sc[j] && (d[j] < dmin || (d[j] == dmin && jp >= 0 && colsol[jp] >= 0 && colsol[j] < 0))) { dmin = d[j]; jp = j; }
                if (colsol[jp] < 0) { end = jp; ds = dmin; break; }
                sc[jp] = 1; setl[q][ns++] = jp;
                int i = colsol[jp]; const float *ci = cost + (size_t)i * n; float h = (ci[jp] - v[jp]) - dmin;
                for (int j = 0; j < n; j++) { float v2 = (ci[j] - v[j]) - h; if (v2 < dmin) v2 = dmin; if (!sc[j] && v2 < d[j]) { d[j] = v2; pred[j] = i; } }
            }
            for (int t = 0; t < ns; t++) setd[q][t] = d[setl[q][t]];
            nset[q] = ns; sink[q] = end; dist[q] = ds;
            int np = 0, e = end, i; do { i = pred[e]; pth[q][np++] = e; pth[q][np++] = i; e = rowsol[i]; } while (i != freerow);
            npth[q] = np;
            if (ns > maxset) maxset = ns;
        }
        crit += maxset;
        int committed = 0;
        for (int q = 0; q < kk; q++) {
            int conflict = stamp[sink[q]] == batch_id;
            for (int t = 0; t < nset[q] && !conflict; t++) conflict = stamp[setl[q][t]] == batch_id;
            if (conflict) break;
            for (int t = 0; t < nset[q]; t++) { int j = setl[q][t]; stamp[j] = batch_id; if (setd[q][t] < dist[q]) { float nv = (v[j] + setd[q][t]) - dist[q]; if (nv < v[j]) v[j] = nv; } }
            stamp[sink[q]] = batch_id;
            for (int t = 0; t < npth[q]; t += 2) { int e = pth[q][t], i = pth[q][t + 1]; colsol[e] = i; rowsol[i] = e; }
            committed++; total_set += nset[q];
        }
        for (int q = committed; q < kk; q++) wasted += nset[q];
        f += committed;
    }
    printf("   speculative K=%d: free=%d batches=%ld critical-path settled=%ld (sequential %ld) wasted=%ld\n", K, numfree, batches, crit, total_set, wasted);
    double tot = 0; for (int i = 0; i < n; i++) tot += cost[(size_t)i * n + rowsol[i]];
    printf("   total = %.9f\n", tot);
}

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    n = atoi(argv[2]); int mode = atoi(argv[3]);
    cost = malloc(sizeof(float) * (size_t)n * n); v = malloc(sizeof(float) * n); rowsol = malloc(sizeof(int) * n); colsol = malloc(sizeof(int) * n);
    FILE *fp = fopen(argv[1], "rb"); if (!fp || fread(cost, sizeof(float), (size_t)n * n, fp) != (size_t)n * n) { fprintf(stderr, "read\n"); return 1; } fclose(fp);
    uint8_t *active = calloc(n, 1); int *imin = malloc(sizeof(int) * n); int *matches = calloc(n, sizeof(int));
    for (int j = 0; j < n; j++) { v[j] = cost[j]; imin[j] = 0; }
    for (int i = 1; i < n; i++) { const float *ci = cost + (size_t)i * n; for (int j = 0; j < n; j++) if (ci[j] < v[j]) { v[j] = ci[j]; imin[j] = i; } }
    {   /* gap statistics at v0 */
        float *gap = malloc(sizeof(float) * n); float cmin = INFINITY, cmax = -INFINITY;
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = 0; i < n; i++) { float umin, usub; int j1, j2; top2(cost + (size_t)i * n, &umin, &j1, &usub, &j2); gap[i] = usub - umin; }
        for (size_t k = 0; k < (size_t)n * n; k++) { if (cost[k] < cmin) cmin = cost[k]; if (cost[k] > cmax) cmax = cost[k]; }
        int cmpf(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return x < y ? -1 : x > y; }
        qsort(gap, n, sizeof(float), cmpf);
        float vmin = INFINITY, vmax = -INFINITY; for (int j = 0; j < n; j++) { if (v[j] < vmin) vmin = v[j]; if (v[j] > vmax) vmax = v[j]; }
        printf("   cost range [%g, %g]; colmin range [%g, %g]; gap quantiles: min %g 10%% %g 50%% %g 90%% %g 99%% %g max %g\n", cmin, cmax, vmin, vmax, gap[0], gap[n / 10], gap[n / 2], gap[n * 9 / 10], gap[n * 99 / 100], gap[n - 1]);
        free(gap);
    }
    if (getenv("NEED")) { V0 = malloc(sizeof(float) * n); memcpy(V0, v, sizeof(float) * n); }
    phase_stats ps; double t0 = omp_get_wtime();
    if (mode == 0) {
        long R = argc > 4 ? atol(argv[4]) : 4096 + n / 4;
        for (int i = 0; i < n; i++) rowsol[i] = -1;
        for (int j = n - 1; j >= 0; j--) { int i = imin[j]; if (++matches[i] == 1) { rowsol[i] = j; colsol[j] = i; } else colsol[j] = -1; }
        float *margin = malloc(sizeof(float) * n);
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = 0; i < n; i++) if (matches[i] == 1) { int j1 = rowsol[i]; const float *ci = cost + (size_t)i * n; float mn = INFINITY; for (int j = 0; j < n; j++) { float h = ci[j] - v[j]; if (j != j1 && h < mn) mn = h; } margin[i] = mn; }
        for (int i = 0; i < n; i++) if (matches[i] == 1) v[rowsol[i]] -= margin[i];
        for (int i = 0; i < n; i++) active[i] = rowsol[i] < 0;
        rounds(0.0f, R, active, &ps, 1);
    } else if (mode == 1) {
        float eps0 = argc > 4 ? atof(argv[4]) : 1e-3f, factor = argc > 5 ? atof(argv[5]) : 5.0f, epsmin = argc > 6 ? atof(argv[6]) : 1e-8f;
        long R0 = argc > 7 ? atol(argv[7]) : 4096 + n / 4;
        int keep = argc > 8 ? atoi(argv[8]) : 0; stop_na = argc > 9 ? atoi(argv[9]) : 0; int stop_last = argc > 10 ? atoi(argv[10]) : 0;     /* 1: eps = 0 phase keeps rows that sit exactly on their minimum */
        long trounds = 0, tbids = 0, twr = 0, tcr = 0;
        for (float eps = eps0; eps >= epsmin; eps /= factor) {
            for (int i = 0; i < n; i++) { rowsol[i] = -1; active[i] = 1; } for (int j = 0; j < n; j++) colsol[j] = -1;
            rounds(eps, 1L << 40, active, &ps, 1);
            trounds += ps.rounds; tbids += ps.bids; twr += ps.wide_rounds; tcr += ps.chain_rounds;
        }
        if (!keep) { for (int i = 0; i < n; i++) { rowsol[i] = -1; active[i] = 1; } for (int j = 0; j < n; j++) colsol[j] = -1; }
        else {
            int viol = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+:viol)
            for (int i = 0; i < n; i++) { float umin, usub; int j1, j2; top2(cost + (size_t)i * n, &umin, &j1, &usub, &j2); int j = rowsol[i]; float mine = cost[(size_t)i * n + j] - v[j]; active[i] = !(mine <= umin); viol += active[i]; }
            for (int i = 0; i < n; i++) if (active[i]) { colsol[rowsol[i]] = -1; rowsol[i] = -1; }
            printf("   eps=0 phase keeps assignments: %d violators\n", viol);
        }
        stop_na = stop_last;
        rounds(0.0f, R0, active, &ps, 1);
        trounds += ps.rounds; tbids += ps.bids; twr += ps.wide_rounds; tcr += ps.chain_rounds;
        printf("   TOTAL rounds=%ld (wide %ld chain %ld) bids=%ld\n", trounds, twr, tcr, tbids);
    }
    if (mode == 2) {
        int K0 = argc > 4 ? atoi(argv[4]) : 8; int mult_log2 = argc > 5 ? atoi(argv[5]) : 5; int flog2 = argc > 6 ? atoi(argv[6]) : 2; int nph = argc > 7 ? atoi(argv[7]) : 10;
        int stop_ph = argc > 8 ? atoi(argv[8]) : 64, stop_last = argc > 9 ? atoi(argv[9]) : 64; long capr = argc > 10 ? atol(argv[10]) : 4096;
        int rb_policy = argc > 11 ? atoi(argv[11]) : -1; raise_mode = argc > 12 ? atoi(argv[12]) : 0; if (rb_policy >= 0) { Fl = malloc(sizeof(float) * n); rebuild_all(); }
        /* gap histogram at v0 */
        long hist[256] = {0};
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = 0; i < n; i++) { float umin, usub; int j1, j2; top2(cost + (size_t)i * n, &umin, &j1, &usub, &j2); float g = usub - umin; uint32_t b; memcpy(&b, &g, 4);
            int e = (b >> 23) & 0xFF;
#pragma omp atomic
            hist[e]++; }
        long cum = 0; int me = 0; for (int e = 0; e < 256; e++) { cum += hist[e]; if (cum * 2 >= n) { me = e; break; } }
        float vmaxabs = 0; for (int j = 0; j < n; j++) if (fabsf(v[j]) > vmaxabs) vmaxabs = fabsf(v[j]);
        for (int i = 0; i < n; i++) rowsol[i] = -1;
        for (int j = n - 1; j >= 0; j--) { int i = imin[j]; if (++matches[i] == 1) { rowsol[i] = j; colsol[j] = i; } else colsol[j] = -1; }
        float *margin = malloc(sizeof(float) * n);
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = 0; i < n; i++) if (matches[i] == 1) { int j1 = rowsol[i]; const float *ci = cost + (size_t)i * n; float mn = INFINITY; for (int j = 0; j < n; j++) { float h = ci[j] - v[j]; if (j != j1 && h < mn) mn = h; } margin[i] = mn; }
        for (int i = 0; i < n; i++) if (matches[i] == 1) v[rowsol[i]] -= margin[i];
        for (int i = 0; i < n; i++) active[i] = rowsol[i] < 0;
        stop_na = 0;
        rounds(0.0f, K0, active, &ps, 1);
        long trounds = ps.rounds, tbids = ps.bids;
        int na = 0; for (int i = 0; i < n; i++) na += active[i];
        printf("   after K0=%d rounds: active %d; median gap exponent %d (2^%d)\n", K0, na, me, me - 127);
        if (na > stop_ph && me > 0) {
            int e0 = me + mult_log2; if (e0 > 254) e0 = 254;
            for (int k = 0; k < nph; k++) {
                int ek = e0 - k * flog2; if (ek < 1) break;
                uint32_t b = (uint32_t)ek << 23; float eps; memcpy(&eps, &b, 4);
                if (eps < vmaxabs * 1.1920929e-07f) break;
                if (raise_mode >= 2 && k > 0) { raise_unassigned("phase"); if (Fl) rebuild_all(); }
                for (int i = 0; i < n; i++) { rowsol[i] = -1; active[i] = 1; } for (int j = 0; j < n; j++) colsol[j] = -1;
                stop_na = (getenv("LATE") && k >= nph - atoi(getenv("LATE"))) ? atoi(getenv("LATE_STOP")) : stop_ph;
                if (Fl && (rb_policy == 1 || (rb_policy > 1 && k < rb_policy))) rebuild_all();
                long d0 = dense_bids;
                rounds(eps, capr, active, &ps, 1);
                if (Fl) printf("      dense bids in phase: %ld\n", dense_bids - d0);
                trounds += ps.rounds; tbids += ps.bids;
            }
            if (raise_mode >= 2) { raise_unassigned("phase"); if (Fl) rebuild_all(); }
            if (getenv("FINAL_KEEP")) {
                /* the last phase keeps every assignment that satisfies complementary slackness exactly (the row sits on its minimum); the
                 * others and the rows the last scaled phase left unassigned bid */
                int viol = 0, unas = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+:viol)
                for (int i = 0; i < n; i++) {
                    if (rowsol[i] < 0) { active[i] = 1; continue; }
                    float umin, usub; int j1, j2; top2(cost + (size_t)i * n, &umin, &j1, &usub, &j2);
                    int j = rowsol[i]; float mine = cost[(size_t)i * n + j] - v[j]; active[i] = !(mine <= umin); viol += active[i];
                }
                for (int i = 0; i < n; i++) { if (rowsol[i] < 0) unas++; else if (active[i]) { colsol[rowsol[i]] = -1; rowsol[i] = -1; } }
                printf("   last phase keeps assignments: %d violators, %d unassigned\n", viol, unas);
            } else {
            for (int i = 0; i < n; i++) { rowsol[i] = -1; active[i] = 1; } for (int j = 0; j < n; j++) colsol[j] = -1;
            }
            stop_na = stop_last;
            { long d0 = dense_bids; rounds(0.0f, capr, active, &ps, 1); if (Fl) printf("      dense bids in last phase: %ld (total %ld)\n", dense_bids - d0, dense_bids); }
            trounds += ps.rounds; tbids += ps.bids;
        } else {
            rounds(0.0f, 4096 + n / 4 - K0, active, &ps, 1);
            trounds += ps.rounds; tbids += ps.bids;
        }
        printf("   TOTAL rounds=%ld bids=%ld\n", trounds, tbids);
    }
    printf("   row reduction time %.1fs\n", omp_get_wtime() - t0);
    if (getenv("SPEC_K")) { spec_K = atoi(getenv("SPEC_K")); searches_spec(); } else searches();
    return 0;
}
