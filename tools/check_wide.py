"""Wide solver (cyto_lap_opts.mode = 2) against the wide-mode oracle on seeded instances: rowsol/colsol/u/v bit for bit,
the semantic counters, and timing.  usage: check_wide.py [--big]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cytospace_amd.lap import lap_solve, lap_solve_rows  # noqa: E402
from oracle.jv import jv_oracle_wide, jv_oracle  # noqa: E402
from tools import instances as I  # noqa: E402

WIDE = dict(mode=2, wide_groups=int(sys.argv[sys.argv.index("--groups") + 1]) if "--groups" in sys.argv else 0,
            wide_par=int(sys.argv[sys.argv.index("--par") + 1]) if "--par" in sys.argv else 0)


def one(c, label, rounds=0, rowmap=None, uniq=None):
    t = time.time()
    o = jv_oracle_wide(c, np.float32, max_rounds=(-1 if rounds == 0 else (0 if rounds < 0 else rounds)))
    to = time.time() - t
    opts = dict(WIDE, wide_rounds=rounds)
    t = time.time()
    try:
        g = lap_solve(c, np.float32, return_info=True, opts=opts) if rowmap is None else \
            lap_solve_rows(uniq, rowmap, return_info=True, opts=opts)
    except Exception as e:  # noqa: BLE001
        print(f"{label:12s} n={len(c):6d} rounds={rounds}: EXCEPTION {e!r}", flush=True)
        return False
    tg = time.time() - t
    inf = g["info"]
    same = [np.array_equal(g[k], o[k]) for k in ("rowsol", "colsol", "u", "v")]
    st = o["stats"]
    cnt = (inf.scans_arr == st.scans_arr, inf.free_after_arr2 == st.free_after_arr, inf.scans_aug_relax == st.scans_aug_relax,
           inf.path_hops == st.path_hops, inf.wide_rounds == st.arr_rounds, inf.wide_retired == st.arr_retired,
           inf.scans_redtransfer == st.scans_redtransfer, inf.wide_scaled == st.arr_scaled, inf.wide_phases == st.arr_phases)
    ok = all(same) and all(cnt) and abs(g["total"] - o["total"]) <= 1e-6 * max(1.0, abs(o["total"]))
    print(f"{label:12s} n={len(c):6d} rounds={rounds:5d} {'OK ' if ok else 'BAD'} same(r,c,u,v)={same} counters={cnt} "
          f"free={inf.free_after_arr2} scaled={inf.wide_scaled}/{inf.wide_phases} arr_rounds={inf.wide_rounds} relax={inf.scans_aug_relax} settled={inf.wide_aug_settled} "
          f"aug_rounds={inf.wide_aug_rounds} par={inf.wide_par_batches}/{inf.wide_par_discarded} dense(arr,aug)=({inf.wide_dense_arr},{inf.wide_dense_aug}) trivial={inf.wide_trivial} "
          f"ms: arr={inf.ms_arr:.2f} aug={inf.ms_aug:.2f} cache={inf.ms_cache:.2f} colred={inf.ms_colred:.2f} | oracle {to:.2f}s gpu-call {tg:.2f}s",
          flush=True)
    if not ok and not all(same):
        bad = np.flatnonzero(g["colsol"] != o["colsol"])
        print("   first colsol diffs", bad[:8], "perm ok", sorted(g["rowsol"]) == list(range(len(c))),
              "total gpu/oracle", g["total"], o["total"], flush=True)
        vb = np.flatnonzero(g["v"] != o["v"])
        print("   v diffs", len(vb), vb[:8], flush=True)
    return ok


def main():
    big = "--big" in sys.argv
    rng = np.random.default_rng(5)
    ok = True
    for n in (1, 2, 3, 5, 17, 64, 65, 200, 513, 1000, 2300):
        ok &= one(rng.random((n, n)).astype(np.float32), "uniform")
    for n in (7, 64, 300, 1000):
        ok &= one(rng.integers(0, 10, (n, n)).astype(np.float32), "ties")
        ok &= one(np.repeat(rng.random(((n + 3) // 4, n)), 4, axis=0)[:n].astype(np.float32), "dups")
    for r in (-1, 1, 2, 7, 50):
        ok &= one(rng.random((700, 700)).astype(np.float32), "budget", rounds=r)
    for r in (1, 3, 8, 9, 40):                       # the first eight rounds run on the whole chip from n = 4096 on
        ok &= one(rng.random((4200, 4200)).astype(np.float32), "head", rounds=r)
    c, loc = I.c3_shaped_cost(5000, 10, 3)
    ok &= one(c, "c3-head")
    c, loc = I.c3_shaped_cost(3000, 10, 3)
    ok &= one(c, "c3-shaped")
    uniq, loc = I.c3_shaped_unique(3000, 10, 3)
    ok &= one(uniq[loc], "c3-rowmap", rowmap=loc, uniq=uniq)
    c, loc = I.c4_chunk_cost(2000, 4)
    ok &= one(c, "c4-chunk")
    if big:
        for n in (5000, 9000):
            ok &= one(I.uniform_cost(n), "uniform")
        c, loc = I.c4_chunk_cost(5000, 4)
        ok &= one(c, "c4-chunk")
    print("ALL OK" if ok else "FAILURES", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
