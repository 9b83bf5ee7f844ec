import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from cytospace_amd.lap import lap_solve
from oracle.jv import jv_oracle
for n in (6000, 10000, 17000):
    c = np.random.default_rng(n).random((n, n))
    g = lap_solve(c, np.float64, return_info=True)
    i = g["info"]
    print(f"n={n}: chain {i.ms_chain:.0f} ms, colred {i.ms_colred:.1f} ms, rt+arr scans {i.scans_redtransfer + i.scans_arr}, aug scans {i.scans_aug_init + i.scans_aug_relax}, rows read {i.hbm_row_reads}", flush=True)
    if n == 17000 or n == 6000:
        t = time.perf_counter(); o = jv_oracle(c, np.float64); dt = time.perf_counter() - t
        print(f"   oracle {dt:.1f} s; identical: {np.array_equal(g['colsol'], o['colsol']) and np.array_equal(g['v'], o['v']) and np.array_equal(g['u'], o['u'])}", flush=True)
