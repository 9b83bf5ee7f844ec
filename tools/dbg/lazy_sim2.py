"""Prototype of the sparse (pruned, cache-certified) augmentation: exactness vs the dense search + statistics."""
import sys, numpy as np
def jv_pre(c):
    n = c.shape[0]
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
        k = 0; prev = list(free); free = []
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
    return v, rowsol, colsol, free

def dense_search(c, v, colsol, f):
    n = len(v)
    d = c[f] - v; pred = np.full(n, f); scanned = np.zeros(n, bool); lvl = np.zeros(n, int)
    level = 0; have = False; curmin = np.float32(0); picks = []
    while True:
        dm = np.where(scanned, np.inf, d); dmin = dm.min()
        cand = np.flatnonzero(dm == dmin); cu = cand[colsol[cand] < 0]
        jp = int(cu[0]) if len(cu) else int(cand[0])
        if (not have) or dmin != curmin: level += 1; curmin = dmin; have = True
        if colsol[jp] < 0: end = jp; break
        scanned[jp] = True; lvl[jp] = level; picks.append(jp)
        i = colsol[jp]; h = (c[i, jp] - v[jp]) - curmin
        v2 = (c[i] - v) - h; upd = (v2 < d) & ~scanned
        d = np.where(upd, v2, d); pred = np.where(upd, i, pred)
    return picks, end, curmin, level, {j: (d[j], pred[j], lvl[j]) for j in picks}, pred[end]

def sparse_search(c, v, colsol, f, ccol, floor, st):
    """entries: col -> [d, pred, scanned, lvl]"""
    n = len(v); un = colsol < 0
    ent = {}
    T = np.float32(np.inf)
    def relax(cols, vals, i):
        nonlocal T
        for j, x in zip(cols, vals):
            if x > T: continue
            e = ent.get(j)
            if e is None:
                ent[j] = [x, i, False, 0]; st["inserts"] += 1
            elif not e[2] and x < e[0]: e[0] = x; e[1] = i
            if un[j] and x < T: T = x
    # init: sparse if certified
    cf = ccol[f]; df = c[f, cf] - v[cf]
    cu = un[cf]
    T0 = df[cu].min() if cu.any() else np.float32(np.inf)
    if floor[f] > T0:
        T = T0; st["init_sparse"] += 1
        relax(cf, df, f)
    else:
        dfull = c[f] - v; T = dfull[un].min(); st["init_dense"] += 1
        idx = np.flatnonzero(dfull <= T); relax(idx, dfull[idx], f)
    level = 0; have = False; curmin = np.float32(0); picks = []
    while True:
        best = None
        for j, e in ent.items():
            if e[2]: continue
            key = (e[0], 0 if un[j] else 1, j)
            if best is None or key < best: best = key
        dmin, _, jp = best
        if (not have) or dmin != curmin: level += 1; curmin = dmin; have = True
        if un[jp]: end = jp; break
        e = ent[jp]; e[2] = True; e[3] = level; picks.append(jp)
        i = colsol[jp]; h = (c[i, jp] - v[jp]) - curmin
        st["scans"] += 1
        if np.float32(floor[i] - h) > T:
            st["cert"] += 1
            ci = ccol[i]; relax(ci, (c[i, ci] - v[ci]) - h, i)
        else:
            v2 = (c[i] - v) - h
            idx = np.flatnonzero(v2 <= T); st["dense_cand"] += len(idx); relax(idx, v2[idx], i)
    st["m_hist"].append(len(ent))
    return picks, end, curmin, level, {j: (ent[j][0], ent[j][1], ent[j][3]) for j in picks}, ent[end][1]

def run(c, K=63, check=True):
    c = c.astype(np.float32); n = c.shape[0]
    v, rowsol, colsol, free = jv_pre(c)
    red = c - v[None, :]
    ccol = np.argpartition(red, K, axis=1)[:, :K]
    floor = np.partition(red, K, axis=1)[:, K].copy()
    st = dict(scans=0, cert=0, init_sparse=0, init_dense=0, inserts=0, dense_cand=0, m_hist=[])
    for f in free:
        res = sparse_search(c, v, colsol, f, ccol, floor, st)
        if check:
            ref = dense_search(c, v, colsol, f)
            assert res[0] == ref[0] and res[1] == ref[1] and res[2] == ref[2] and res[3] == ref[3] and res[5] == ref[5], "mismatch"
            for j in res[0]: assert res[4][j] == ref[4][j], ("entry mismatch", j, res[4][j], ref[4][j])
        picks, end, curmin, level, info, pend = res
        for j in picks:
            dj, pj, lj = info[j]
            if lj < level: v[j] = (v[j] + dj) - curmin
        pred = {j: info[j][1] for j in picks}; pred[end] = pend
        ep = end
        while True:
            i = pred[ep]; colsol[ep] = i; j1 = ep; ep = rowsol[i]; rowsol[i] = j1
            if i == f: break
    m = np.array(st.pop("m_hist")) if st["m_hist"] else np.array([0])
    return st, m, len(free)
if __name__ == "__main__":
    n = int(sys.argv[1]); dup = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(n)
    if dup > 1:
        base = -(rng.random((n // dup, n)) ** 3).astype(np.float32); c = np.repeat(base, dup, axis=0)
    else:
        c = rng.random((n, n)).astype(np.float32)
    st, m, nf = run(c)
    print(f"n={n} dup={dup} free={nf}", st, f"cert={st['cert']/max(1,st['scans']):.3f} m: mean {m.mean():.0f} p90 {np.percentile(m,90):.0f} max {m.max()}")
