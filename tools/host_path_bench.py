"""Wall time of the solver-only seam as the reference would call it: lapjv_hip(cost) with a float64 numpy array on the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.lap import lapjv_hip, lap_solve
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
c64 = np.random.default_rng(n).random((n, n))
lapjv_hip(c64[:2000, :2000].copy())          # warm-up (library load, first HIP call)
t = time.perf_counter(); r = lapjv_hip(c64); t1 = time.perf_counter() - t
t = time.perf_counter(); c32 = np.ascontiguousarray(c64, dtype=np.float32); t2 = time.perf_counter() - t
t = time.perf_counter(); g = lap_solve(c32, np.float32, return_info=True); t3 = time.perf_counter() - t
print(f"n={n}: lapjv_hip(float64 host array, narrowed on the device) {t1:.2f}s; lap_solve(float32 host array) {t3:.2f}s "
      f"(kernels {g['info'].ms_total/1e3:.2f}s => H2D + alloc + D2H {t3 - g['info'].ms_total/1e3:.2f}s); "
      f"for comparison numpy's float64->float32 pass on the host alone: {t2:.2f}s")
