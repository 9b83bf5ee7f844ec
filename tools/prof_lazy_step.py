"""Cycle breakdown of one cache-certified augmentation step (jv_aug_lazy; s_memtime stamps of wave 0; build with -DLZ_PROF).
Usage (on the GPU box): python tools/prof_lazy_step.py [u20000 ...]   -> builds tools/libcytohip_lzprof.so, runs, prints."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cytospace_amd import build as B, _lib
lib = os.path.join(ROOT, "tools", "libcytohip_lzprof.so")
if not os.path.exists(lib) or "--rebuild" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DLZ_PROF", "-o", lib] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"])
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = lib
import numpy as np
from cytospace_amd.lap import lap_solve
from tools import instances
L = _lib.lib()
NAMES = ["-", "pick (LDS block minima, two wave reductions)", "owner look-up, loads issued, wait for them (vmcnt 0)", "h, skip test, new minimum of the picked block, logs",
         "certificate + cached relaxation (LDS gathers, atomics)", "exception list"]
for w in [a for a in sys.argv[1:] if not a.startswith("--")] or ["u20000"]:
    cost = instances.c4_chunk_cost(int(w[3:]))[0] if w.startswith("c4s") else instances.uniform_cost(int(w[1:]))
    n = len(cost)
    buf = _lib.DeviceBuffer.from_numpy(cost)
    out = (ctypes.c_longlong * 24)()
    lap_solve(None, np.float32, device_ptr=buf.ptr, n=n, ld=n)
    r = lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
    L.cyto_lz_prof_read(out)
    i = r["info"]
    print(f"{w}: {i.augmentations} searches ({out[19]} sparse inits), {i.scans_aug_relax} scans ({out[18]} full-row), jv_aug_lazy {i.ms_aug:.1f} ms = {i.ms_aug * 1e3 / max(1, i.scans_aug_relax):.3f} us/scan (stamps in)")
    for k in range(1, 6):
        print(f"   {NAMES[k]:62s} {out[k] / max(1, out[6 + k]):8.1f} ticks per pass x {out[6 + k]:8d} passes = {out[k] / 2.4e6:8.2f} ms")
    print(f"   distance atomics per cached step: {out[13] / max(1, out[6 + 4]):.1f}")
    buf.free()
