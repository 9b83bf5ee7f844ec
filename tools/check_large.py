"""One-off parity check of the large-n code paths (too slow for the test suite)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.lap import lap_solve
from oracle.jv import jv_oracle
cases = ((27000, np.float32), (33000, np.float32), (17000, np.float64))
if len(sys.argv) > 1:
    cases = tuple((int(a), np.float32) for a in sys.argv[1:])
for n, dt in cases:
    c = np.random.default_rng(n).random((n, n)).astype(np.float32)
    t = time.time(); g = lap_solve(c, dt, return_info=True); tg = time.time() - t
    t = time.time(); o = jv_oracle(c, dt); to = time.time() - t
    ok = (np.array_equal(g["rowsol"], o["rowsol"]) and np.array_equal(g["colsol"], o["colsol"]) and
          np.array_equal(g["u"], o["u"]) and np.array_equal(g["v"], o["v"]) and g["info"].row_scans == o["stats"].row_scans)
    i = g["info"]
    print(f"n={n} {dt.__name__}: parity={ok} gpu={tg:.1f}s (kernels {i.ms_total/1e3:.2f}s: colred {i.ms_colred:.0f} cache {i.ms_cache:.0f} "
          f"arr {i.ms_arr:.0f} aug {i.ms_aug:.0f} ms; arr scans {i.scans_arr} dense {i.dense_refreshes} aug scans {i.scans_aug_relax}) "
          f"oracle={to:.1f}s", flush=True)
