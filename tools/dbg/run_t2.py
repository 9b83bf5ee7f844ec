import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cytospace_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcytohip_t.so")
import numpy as np
from cytospace_amd.lap import lap_solve
n, dup = 10000, 5
rng = np.random.default_rng(n)
base = -(rng.random((n // dup, n)) ** 3).astype(np.float32)
c = np.repeat(base, dup, axis=0)
r = lap_solve(c, np.float32, return_info=True)
print(n, r["info"].ms_aug, flush=True)
