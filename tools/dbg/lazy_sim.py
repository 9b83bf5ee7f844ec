"""Prototype: how many augmentation scans could be served from per-row caches (floor certificates)?
numpy JV (same phases as the oracle, not bit-faithful in tie-breaking) + instrumentation."""
import sys, numpy as np
def run(c, K=63):
    n = c.shape[0]
    c = c.astype(np.float32)
    v = c.min(0); imin = c.argmin(0)
    rowsol = -np.ones(n, int); colsol = -np.ones(n, int); matches = np.zeros(n, int)
    for j in range(n - 1, -1, -1):
        i = imin[j]; matches[i] += 1
        if matches[i] == 1: rowsol[i] = j; colsol[j] = i
    free = [i for i in range(n) if matches[i] == 0]
    for i in range(n):
        if matches[i] == 1:
            j1 = rowsol[i]; h = c[i] - v; h[j1] = np.inf; v[j1] -= h.min()
    for sweep in range(2):
        k = 0; prev = free; free = []
        prev = list(prev)
        while k < len(prev):
            i = prev[k]; k += 1
            h = c[i] - v
            j1 = int(h.argmin()); umin = h[j1]; h2 = h.copy(); h2[j1] = np.inf; j2 = int(h2.argmin()); usub = h2[j2]
            i0 = colsol[j1]
            vnew = v[j1] - (usub - umin)
            lowers = vnew < v[j1]
            if lowers: v[j1] = vnew
            elif i0 >= 0: j1 = j2; i0 = colsol[j2]
            rowsol[i] = j1; colsol[j1] = i
            if i0 >= 0:
                if lowers: k -= 1; prev[k] = i0
                else: free.append(i0)
    # caches at augmentation start: floor = (K+1)-th smallest reduced cost of each row
    red = c - v[None, :]
    part = np.partition(red, K, axis=1)
    floor = part[:, K].copy()
    cache = [set(np.argpartition(red[i], K)[:K].tolist()) for i in range(n)]
    stats = dict(scans=0, cert=0, init=0, init_cert=0, dense_after=0)
    for f in free:
        d = c[f] - v; pred = np.full(n, f); scanned = np.zeros(n, bool)
        un = colsol < 0
        stats["init"] += 1
        # sparse init possible if some unassigned column in the cache has d <= floor
        cf = np.fromiter(cache[f], int)
        cu = cf[un[cf]]
        dense_mode = True
        if len(cu) and d[cu].min() <= floor[f]:
            stats["init_cert"] += 1; dense_mode = False
        known = np.zeros(n, bool)
        if not dense_mode: known[cf] = True
        else: known[:] = True
        lvl = np.zeros(n, int); level = 0; have = False; curmin = 0.0
        while True:
            dm = np.where(scanned, np.inf, d)
            dmin = dm.min()
            cand = np.flatnonzero((dm == dmin))
            cu2 = cand[un[cand]]
            jp = int(cu2[0]) if len(cu2) else int(cand[0])
            if (not have) or dmin != curmin: level += 1; curmin = dmin; have = True
            if colsol[jp] < 0: end = jp; break
            scanned[jp] = True; lvl[jp] = level
            i = colsol[jp]
            h = (c[i, jp] - v[jp]) - curmin
            stats["scans"] += 1
            # certificate: floor_i - h >= T_ub, T_ub = min d over KNOWN unassigned unscanned columns
            ku = known & un & ~scanned
            T_ub = d[ku].min() if ku.any() else np.inf
            if floor[i] - h >= T_ub:
                stats["cert"] += 1
                ci = np.fromiter(cache[i], int); known[ci] = True
            else:
                if not dense_mode: stats["dense_after"] += 1
                known[:] = True
            v2 = (c[i] - v) - h
            upd = (v2 < d) & ~scanned
            d = np.where(upd, v2, d); pred = np.where(upd, i, pred)
        m = scanned & (lvl < level)
        v[m] = (v[m] + d[m]) - curmin
        ep = end
        while True:
            i = pred[ep]; colsol[ep] = i; j1 = ep; ep = rowsol[i]; rowsol[i] = j1
            if i == f: break
    tot = c[np.arange(n), rowsol].sum()
    return stats, tot
if __name__ == "__main__":
    n = int(sys.argv[1]); dup = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(n)
    if dup > 1:
        base = -(rng.random((n // dup, n)) ** 3).astype(np.float32); c = np.repeat(base, dup, axis=0)
    else:
        c = rng.random((n, n)).astype(np.float32)
    st, tot = run(c)
    from scipy.optimize import linear_sum_assignment
    r, cc = linear_sum_assignment(c.astype(np.float64))
    print(n, dup, st, "cert frac %.3f" % (st["cert"] / max(1, st["scans"])), "init frac %.3f" % (st["init_cert"] / max(1, st["init"])),
          "total ok", abs(tot - c[r, cc].sum()) < 1e-2)
