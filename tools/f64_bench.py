"""float64 solves (the force_doubles / lapjv_compat precision): the default, warm-started from the float32 wide solve of the narrowed
matrix, beside the cold classic chain (cyto_lap_opts.mode = 1).  usage: f64_bench.py [n ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cytospace_amd.lap import lap_solve  # noqa: E402
from oracle.jv import jv_oracle  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [6000, 10000, 17000]:
    c = np.random.default_rng(n).random((n, n))
    for label, opts in (("warm", None), ("warm", None), ("cold", dict(mode=1))):
        t = time.perf_counter()
        g = lap_solve(c, np.float64, return_info=True, opts=opts)
        wall = time.perf_counter() - t
        i = g["info"]
        print(f"n={n} {label}: kernels {i.ms_total:.1f} ms (float32 wide solve inside: {i.f64_warm_ms:.1f}), wall incl. H2D {wall * 1e3:.0f} ms, "
              f"rt+arr scans {i.scans_redtransfer + i.scans_arr}, free after arr {i.free_after_arr2}, aug scans {i.scans_aug_init + i.scans_aug_relax}, "
              f"rows read {i.hbm_row_reads}", flush=True)
        if label == "cold":
            cold = g
        else:
            warm = g
    print(f"   same indices warm / cold: {np.array_equal(warm['colsol'], cold['colsol'])}", flush=True)
    if n <= 6000:
        t = time.perf_counter(); o = jv_oracle(c, np.float64, warm=True); dt = time.perf_counter() - t
        print(f"   warm oracle {dt:.1f} s; identical: {all(np.array_equal(warm[k], o[k]) for k in ('rowsol', 'colsol', 'u', 'v'))}", flush=True)
