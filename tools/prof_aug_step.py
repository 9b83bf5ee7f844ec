"""Cycle breakdown of one dense augmentation step (s_memtime stamps of wave 0; build with -DCYTO_AUG_PROF).
Usage (on the GPU box): python tools/prof_aug_step.py [c4s10000 ...]   -> builds tools/libcytohip_prof.so, runs, prints."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cytospace_amd import build as B, _lib
lib = os.path.join(ROOT, "tools", "libcytohip_prof.so")
if not os.path.exists(lib) or "--rebuild" in sys.argv:
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DCYTO_AUG_PROF", "-o", lib] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"]
    subprocess.check_call(cmd)
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = lib
import numpy as np
from cytospace_amd.lap import lap_solve
from tools import instances
L = _lib.lib()
NAMES = ["pick: key reduce + record read", "decode record + publish group", "row load issue + log", "retire column",
         "wait for the row (vmcnt 0)", "relax", "skip path", "path + price update (per search)",
         "pick: thread min + wave min", "pick: column search (divergent)", "pick: wave min of columns", "pick: look-ups", "pick: post record + barrier", "-"]
for w in [a for a in sys.argv[1:] if not a.startswith("--")] or ["c4s10000"]:
    cost = instances.c4_chunk_cost(int(w[3:]))[0] if w.startswith("c4s") else instances.uniform_cost(int(w[1:]))
    n = len(cost)
    buf = _lib.DeviceBuffer.from_numpy(cost)
    out = (ctypes.c_longlong * 16)()
    lap_solve(None, np.float32, device_ptr=buf.ptr, n=n, ld=n, opts=dict(augmentation=1))
    L.cyto_aug_prof_read(out)
    r = lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n, opts=dict(augmentation=1))
    L.cyto_aug_prof_read(out)
    i = r["info"]
    steps = i.scans_aug_relax
    real = steps - i.aug_scans_skipped
    print(f"{w}: {i.augmentations} searches, {steps} scans ({i.aug_scans_skipped} skipped), aug kernel {i.ms_aug:.1f} ms = {i.ms_aug*1e3/steps:.3f} us/scan")
    tot = sum(out[k] for k in range(14))
    for k in range(13):
        per = out[k] / (i.augmentations if k == 7 else (i.aug_scans_skipped or 1) if k == 6 else steps)
        print(f"   {NAMES[k]:42s} {out[k]/1e6:10.1f} Mcycles (100 MHz s_memtime ticks: x clock ratio)  {100.0*out[k]/tot:5.1f} %   {per:8.1f} ticks per {'search' if k == 7 else 'scan'}")
    print(f"   total ticks {tot/1e6:.1f} M over {i.ms_aug:.1f} ms -> {tot/(i.ms_aug*1e3):.1f} ticks/us")
    buf.free()
