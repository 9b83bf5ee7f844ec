"""Row-cache build kernels side by side (developer tool): `cyto_lap_info.ms_cache` (HIP events around the build that follows the
column reduction) for the wave-per-row builder at several (waves per CU, quads in flight) settings and for the workgroup-per-row
builders (cyto_lap_opts.cache_waves = -1), on a uniform n x n matrix resident in HBM.  The solve's results must not depend on the builder:
rowsol/colsol/u/v and the counters of every setting are compared with the first one's.

    python tools/cache_build_bench.py 20000 50000 [--typed]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.lap import lap_solve
from cytospace_amd import _lib

SETTINGS = [("0", "4"), ("8", "4"), ("16", "4"), ("32", "4"), ("8", "8"), ("16", "8"), ("20", "8")]


def main():
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [20000]
    typed = "--typed" in sys.argv
    for n in sizes:
        if typed:
            from tools.instances import typed_unique_cost
            c, _ = typed_unique_cost(n, n, 20)
        else:
            c = np.random.default_rng(n).random((n, n)).astype(np.float32)
        buf = _lib.DeviceBuffer.from_numpy(c)
        ref = None
        for waves, unroll in SETTINGS:
            o = dict(cache_waves=int(waves) if int(waves) > 0 else -1, cache_unroll=int(unroll))
            best = 1e9
            for rep in range(4):
                r = lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n, opts=o)
                best = min(best, r["info"].ms_cache)
            i = r["info"]
            key = (r["rowsol"].tobytes(), r["colsol"].tobytes(), r["u"].tobytes(), r["v"].tobytes(), i.row_scans)
            if ref is None:
                ref = key
            same = key == ref
            print(f"n={n} waves/CU={waves:>2} U={unroll} cache_ms={best:.3f} ({4.0 * n * n / (best * 1e-3) / 1e12:.2f} TB/s) "
                  f"total_ms={i.ms_total:.2f} hbm_rows={i.hbm_row_reads} dense_refresh={i.dense_refreshes} same_result={same}", flush=True)
        buf.free()


main()
