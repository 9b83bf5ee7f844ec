"""One instance of tools/stress_lap.py against the classic oracle through the chain solver (mode 1): what differs.
usage: one_seed.py <seed> [lo hi]   (run from a checkout's root: imports ./cytospace_amd)"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from cytospace_amd.lap import lap_solve
sys.path.insert(0, "/root/repo")
from oracle.jv import jv_oracle
from tools.stress_lap import make
s = int(sys.argv[1]); lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3000, 6000)
kinds = ["uniform", "dup", "ints", "types", "constcols", "scale"]
rng = np.random.default_rng(1000 + s); kind = kinds[s % len(kinds)]; n = int(rng.integers(lo, hi))
c = make(kind, n, rng)
o = jv_oracle(c, np.float32)
for rep in range(3):
    g = lap_solve(c, np.float32, return_info=True, opts=dict(mode=1))
    d = {k: int((g[k] != o[k]).sum()) for k in ("rowsol", "colsol", "u", "v")}
    print(os.getcwd(), kind, n, "differing entries", d, "total", g["total"], o["total"], "scans", g["info"].row_scans, o["stats"].row_scans, flush=True)
    i, st = g["info"], o["stats"]
    print("   gpu ", {k: getattr(i, k) for k in ("scans_colred", "scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax", "augmentations", "path_hops")})
    print("   cpu ", {k: getattr(st, k) for k in ("scans_colred", "scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax", "augmentations", "path_hops") if hasattr(st, k)}, flush=True)
