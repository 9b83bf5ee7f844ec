"""Per-search scan counts of the dense augmentation kernel (chain solver, mode 1) against the classic oracle's.
Usage (on the GPU box): python tools/trace_aug_scans.py [seed]   -> builds tools/libcytohip_trace.so (-DCYTO_AUG_TRACE), runs, prints
the searches whose scan count differs."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cytospace_amd import build as B, _lib
lib = os.path.join(ROOT, "tools", "libcytohip_trace.so")
if not os.path.exists(lib) or "--rebuild" in sys.argv:
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DCYTO_AUG_TRACE", "-o", lib] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-L/opt/rocm/lib", "-lrccl", "-lpthread"]
    subprocess.check_call(cmd)
if "--build-only" in sys.argv:
    sys.exit(0)
_lib.LIB_PATH = lib
import numpy as np
from cytospace_amd.lap import lap_solve
from tools.stress_lap import make
from oracle.jv import jv_oracle_trace
seed = int([a for a in sys.argv[1:] if not a.startswith("--")][0]) if len([a for a in sys.argv[1:] if not a.startswith("--")]) else 2407
kinds = ["uniform", "dup", "ints", "types", "constcols", "scale"]
rng = np.random.default_rng(1000 + seed)
n = int(rng.integers(3000, 6000))
c = make(kinds[seed % 6], n, rng)
o, otr = jv_oracle_trace(c)
g = lap_solve(c, np.float32, return_info=True, opts=dict(mode=1))
L = _lib.lib()
na = g["info"].augmentations
buf = (ctypes.c_longlong * (2 * na))()
L.cyto_aug_trace_read(buf, na)
gt = np.array(buf[:]).reshape(na, 2)
gper = np.diff(np.concatenate([[0], gt[:, 0]]))
oper = otr[:na, 0]
print(f"n {n}: searches {na} / oracle {o['stats'].augmentations}; scans {gt[-1, 0]} / {o['stats'].scans_aug_relax}; elided {gt[-1, 1]}")
for f in np.nonzero(gper != oper)[0]:
    print(f"  search {f}: free row {otr[f, 1]}, kernel {gper[f]} scans, oracle {oper[f]}; levels {otr[f, 2]}, sink {otr[f, 3]}, "
          f"scanned at the final distance {otr[f, 4]}, unassigned columns at the final distance {otr[f, 5]}, final distance {otr[f, 6]!r}")
