"""Randomised parity stress against the CPU oracle (structures that reach the rare paths: duplicated rows, heavy integer ties,
few cell types, constant columns, tiny and huge magnitudes).  Default: the float32 dispatch = the wide solver against the oracle's
WIDE mode (--groups G: its searches on G workgroups; --rounds R: round budget; --rebuild K: row caches rebuilt every K searches, -1 never); --chain: the chain solver against the classic mode;
--f64: float64 through the streaming chain.  usage: stress_lap.py [seed0 count lo hi] [--chain | --f64] [--groups G] [--par K] [--wipe K] [--rounds R] [--rebuild K]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.lap import lap_solve
from oracle.jv import jv_oracle, jv_oracle_wide

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
    def flag(name, default):
        if name in sys.argv:
            k = sys.argv.index(name); v = int(sys.argv[k + 1]); del sys.argv[k:k + 2]; return v
        return default
    groups, rounds, rebuild, par, wipe = flag("--groups", 0), flag("--rounds", 0), flag("--rebuild", 0), flag("--par", 0), flag("--wipe", 0)
    f64 = "--f64" in sys.argv                 # float64 through the streaming chain with row caches (chain_variant 1) at any size
    chain = "--chain" in sys.argv
    sys.argv = [a for a in sys.argv if a not in ("--f64", "--chain")]
    dt = np.float64 if f64 else np.float32
    opts = dict(chain_variant=1) if f64 else (dict(mode=1) if chain else dict(mode=2, wide_groups=groups, wide_rounds=rounds, wide_rebuild=rebuild, wide_par=par, wide_wipe=wipe))
    wide = not (f64 or chain)
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
        g = lap_solve(c, dt, return_info=True, opts=opts)
        i = g["info"]
        if wide:
            o = jv_oracle_wide(c, dt, max_rounds=(-1 if rounds == 0 else max(rounds, 0)))
            st = o["stats"]
            ok = all(np.array_equal(g[k], o[k]) for k in ("rowsol", "colsol", "u", "v")) and i.scans_arr == st.scans_arr and \
                i.scans_aug_relax == st.scans_aug_relax and i.wide_rounds == st.arr_rounds and i.wide_retired == st.arr_retired and i.path_hops == st.path_hops and i.wide_scaled == st.arr_scaled and i.wide_phases == st.arr_phases
            print(f"{s:3d} {kind:9s} n={n}: {'ok ' if ok else 'MISMATCH'} scaled {i.wide_scaled}/{i.wide_phases} par {i.wide_par_batches}/{i.wide_par_discarded} rounds {i.wide_rounds} bids {i.scans_arr} free {i.free_after_arr2} settled {i.wide_aug_settled} "
                  f"for {i.scans_aug_relax} dense ({i.wide_dense_arr},{i.wide_dense_aug}) launches {i.wide_aug_launches} one-edge {i.wide_trivial} {i.ms_total:.0f} ms", flush=True)
        else:
            o = jv_oracle(c, dt)
            ok = all(np.array_equal(g[k], o[k]) for k in ("rowsol", "colsol", "u", "v")) and g["info"].row_scans == o["stats"].row_scans
            print(f"{s:3d} {kind:9s} n={n}: {'ok ' if ok else 'MISMATCH'} rows read {i.hbm_row_reads} of {i.row_scans} scans, aug scans {i.scans_aug_relax} dense {i.aug_dense_scans} sparse {i.aug_sparse_inits}/{i.augmentations} "
                  f"handover {i.aug_handover} arr refresh {i.dense_refreshes} {i.ms_total:.0f} ms", flush=True)
        bad += (not ok)
    print(f"{count} instances, {bad} mismatches, {time.time()-t0:.0f}s")
