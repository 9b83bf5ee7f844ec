"""Scratch timing of the HIP LAP at several sizes (developer tool, not the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.lap import lap_solve
from cytospace_amd import _lib

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dup = 1
    for a in sys.argv[1:]:
        if a.startswith("--dup="):
            dup = int(a.split("=")[1])
    sizes = [int(x) for x in args] or [1000, 4000, 8000]
    for n in sizes:
        if dup > 1:   # Visium-like: correlation-shaped cost, every spot row repeated `dup` times
            rng = np.random.default_rng(n)
            base = -(rng.random((n // dup, n)) ** 3).astype(np.float32)
            c = np.repeat(base, dup, axis=0)
        else:
            c = np.random.default_rng(n).random((n, n)).astype(np.float32)
        buf = _lib.DeviceBuffer.from_numpy(c)
        for rep in range(2):
            t = time.perf_counter()
            r = lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
            dt = time.perf_counter() - t
        i = r["info"]
        R = i.row_scans
        chain_scans = R - n
        print(f"n={n} wall={dt*1e3:.1f}ms colred={i.ms_colred:.3f}ms chain={i.ms_chain:.1f}ms scans={R} "
              f"us/scan={i.ms_chain*1e3/max(1,chain_scans):.3f} algGB/s={4*n*R/ (i.ms_total*1e-3)/1e9:.1f} "
              f"colredGB/s={4*n*n/(i.ms_colred*1e-3)/1e9:.0f} assign/s={n/(i.ms_total*1e-3):.0f} "
              f"arr={i.scans_arr} augrelax={i.scans_aug_relax} cache_ms={i.ms_cache:.3f} arr_ms={i.ms_arr:.1f} aug_ms={i.ms_aug:.1f} us/arr={i.ms_arr*1e3/max(1,i.scans_arr+i.scans_redtransfer):.3f} us/aug={i.ms_aug*1e3/max(1,i.scans_aug_relax+i.scans_aug_init):.3f} dense_refresh={i.dense_refreshes} groups={i.row_groups} aug_skipped={i.aug_scans_skipped} hbm_rows={i.hbm_row_reads} aug_dense={i.aug_dense_scans} augs={i.augmentations} sparse_init={i.aug_sparse_inits}", flush=True)
        buf.free()

main()
