"""Augmentation kernel comparison (cyto_lap_opts.augmentation = 1 dense / 2 cache-certified / 0 default) on a few instance
families; prints kernel times and per-scan cost.  Results are bit-identical by construction (checked here)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd import _lib
from cytospace_amd.lap import lap_solve
from tools import instances

def run(tag, cost):
    n = len(cost)
    buf = _lib.DeviceBuffer.from_numpy(cost)
    ref = None
    for name, opts in (("default", None), ("dense", dict(augmentation=1)), ("lazy", dict(augmentation=2, no_handover=1))):
        if name == "dense" and n > 26624:
            continue
        lap_solve(None, np.float32, device_ptr=buf.ptr, n=n, ld=n, opts=opts)
        r = lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n, opts=opts)
        i = r["info"]
        same = ref is None or all(np.array_equal(r[k], ref[k]) for k in ("colsol", "rowsol", "u", "v"))
        ref = ref or r
        print(f"{tag:>14} n={n:6d} {name:8s} arr {i.ms_arr:8.1f} ms  aug {i.ms_aug:9.1f} ms  scans {i.scans_aug_relax:8d} searches {i.augmentations:6d} "
              f"skipped {i.aug_scans_skipped:7d} full-row {i.aug_dense_scans:7d} handover {i.aug_handover:5d}  "
              f"us/scan {i.ms_aug * 1e3 / max(1, i.scans_aug_relax + i.augmentations):6.3f}  identical={same}", flush=True)
    buf.free()

which = sys.argv[1:] or ["c4s5000", "c4s10000", "u10000", "u20000", "c3s20000", "c4s20000"]
for w in which:
    if w.startswith("c4s"):
        run(w, instances.c4_chunk_cost(int(w[3:]))[0])
    elif w.startswith("c3s"):
        run(w, instances.c3_shaped_cost(int(w[3:]))[0])
    else:
        run(w, instances.uniform_cost(int(w[1:])))
