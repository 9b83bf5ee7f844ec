"""Cycle breakdown of one cached ARR step of jv_chain2 (s_memtime stamps on wave 0; build with -DCYTO_ARR_PROF).
Usage (on the GPU box): python tools/prof_arr_step.py [n ...]   -> builds tools/libcytohip_arrprof.so, runs, prints."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cytospace_amd import build as B, _lib
lib = os.path.join(ROOT, "tools", "libcytohip_arrprof.so")
if not os.path.exists(lib) or "--rebuild" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DCYTO_ARR_PROF", "-o", lib] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"])
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = lib
import numpy as np
from cytospace_amd.lap import lap_solve
from tools import instances
L = _lib.lib()
NAMES = ["wait for the row's cache (L2; prefetched during the previous step)", "LDS gathers of v and colsol (64 lanes)",
         "first reduction + lane of the minimum + prefetch issue", "second reduction + read-lanes", "price / colsol update (LDS stores)"]
for n in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [20000]:
    buf = _lib.DeviceBuffer.from_numpy(instances.uniform_cost(n))
    out = (ctypes.c_longlong * 16)()
    lap_solve(None, np.float32, device_ptr=buf.ptr, n=n, ld=n)
    r = lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
    L.cyto_arr_prof_read(out)
    i = r["info"]
    steps = out[8]
    tot = sum(out[k] for k in range(5))
    print(f"n={n}: jv_chain2 {i.ms_arr:.1f} ms, {i.scans_arr} ARR + {i.scans_redtransfer} RT scans, {steps} cached ARR steps stamped "
          f"({i.ms_arr * 1e3 / max(1, i.scans_arr + i.scans_redtransfer):.3f} us per scan with the stamps in; s_memtime ticks at 2.4 GHz)")
    for k in range(5):
        print(f"   {NAMES[k]:72s} {out[k] / max(1, steps):8.1f} ticks per step  {100.0 * out[k] / max(1, tot):5.1f} %")
    print(f"   sum {tot / max(1, steps):.1f} ticks per step = {tot / max(1, steps) / 2400:.3f} us")
    buf.free()
