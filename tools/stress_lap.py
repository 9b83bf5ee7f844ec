"""Randomised parity stress of the default float32 dispatch against the CPU oracle (structures that reach the rare
paths: duplicated rows, heavy integer ties, few cell types, constant columns, tiny and huge magnitudes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.lap import lap_solve
from oracle.jv import jv_oracle

def make(kind, n, rng):
    if kind == "uniform":
        return rng.random((n, n)).astype(np.float32)
    if kind == "dup":
        k = int(rng.integers(2, 9)); base = -(rng.random((n // k + 1, n)) ** 3).astype(np.float32)
        return np.repeat(base, k, axis=0)[:n]
    if kind == "ints":
        return rng.integers(0, int(rng.integers(3, 50)), (n, n)).astype(np.float32)
    if kind == "types":
        t = int(rng.integers(3, 12)); p = rng.normal(size=(t, 32)).astype(np.float32)
        a = p[rng.integers(0, t, n)] + 0.05 * rng.normal(size=(n, 32)).astype(np.float32)
        b = p[rng.integers(0, t, n)] + 0.05 * rng.normal(size=(n, 32)).astype(np.float32)
        return -(a @ b.T).astype(np.float32)
    if kind == "constcols":
        c = rng.random((n, n)).astype(np.float32); c[:, rng.integers(0, n, n // 10)] = 0.5; return c
    if kind == "scale":
        return (rng.random((n, n)) * 10.0 ** rng.integers(-20, 20)).astype(np.float32) - np.float32(10.0 ** rng.integers(-3, 3))
    raise ValueError(kind)

if __name__ == "__main__":
    f64 = "--f64" in sys.argv                 # float64 through the streaming chain with row caches (chain_variant 1) at any size
    sys.argv = [a for a in sys.argv if a != "--f64"]
    dt = np.float64 if f64 else np.float32
    opts = dict(chain_variant=1) if f64 else None
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 36
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (5200, 8200)
    kinds = ["uniform", "dup", "ints", "types", "constcols", "scale"]
    bad = 0
    t0 = time.time()
    for s in range(seed0, seed0 + count):
        rng = np.random.default_rng(1000 + s)
        kind = kinds[s % len(kinds)]; n = int(rng.integers(lo, hi))
        c = make(kind, n, rng)
        if f64: c = c.astype(np.float64) + (1e-16 * rng.random(c.shape) if s % 2 else 0.0)
        g = lap_solve(c, dt, return_info=True, opts=opts); o = jv_oracle(c, dt)
        ok = all(np.array_equal(g[k], o[k]) for k in ("rowsol", "colsol", "u", "v")) and g["info"].row_scans == o["stats"].row_scans
        i = g["info"]
        print(f"{s:3d} {kind:9s} n={n}: {'ok ' if ok else 'MISMATCH'} rows read {i.hbm_row_reads} of {i.row_scans} scans, aug scans {i.scans_aug_relax} dense {i.aug_dense_scans} sparse {i.aug_sparse_inits}/{i.augmentations} "
              f"handover {i.aug_handover} arr refresh {i.dense_refreshes} {i.ms_total:.0f} ms", flush=True)
        bad += (not ok)
    print(f"{count} instances, {bad} mismatches, {time.time()-t0:.0f}s")
