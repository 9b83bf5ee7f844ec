import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cytospace_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcytohip_t.so")
import numpy as np
from cytospace_amd.lap import lap_solve
for n in (1000, 20000):
    c = np.random.default_rng(n).random((n, n)).astype(np.float32)
    r = lap_solve(c, np.float32, return_info=True)
    print(n, r["info"].ms_arr, flush=True)
