"""Where config c3's wall time goes beyond its kernels (developer tool): the fused call with float32 / uint16 counts, the resident-counts
path piece by piece (context = transforms; the chunk = gathers + GEMM + LAP), and the LAP alone on the finished cost matrix (wall against
its own kernel-time sum: what the host side of one solve costs)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cytospace_amd import _lib, common  # noqa: E402
from cytospace_amd.cytospace import ExpressionContext, assign_pearson  # noqa: E402
from cytospace_amd.lap import lap_solve_rows  # noqa: E402
from tools import instances  # noqa: E402

G, C, S = 20000, 50000, 5000
sc, st, slots = instances.synth_expression(G, C, S, seed=1)
sc16, st16 = sc.astype(np.uint16), st.astype(np.uint16)
L = _lib.lib()


def wall(f, reps=3):
    out = []
    for _ in range(reps):
        _lib.check(L.cyto_device_synchronize(0))
        t = time.perf_counter()
        r = f()
        _lib.check(L.cyto_device_synchronize(0))
        out.append((time.perf_counter() - t, r))
    return min(out, key=lambda x: x[0])


for name, a, b in (("float32", sc, st), ("uint16", sc16, st16)):
    assign_pearson(a, b, slots, already_normalized=False)
    w, (m, tot, info) = wall(lambda: assign_pearson(a, b, slots, already_normalized=False, return_info=True))
    print(f"fused, {name} counts: wall {w * 1e3:.1f} ms | upload+transform {info.ms_standardize:.1f} gemm blocks sum {info.ms_gemm:.1f} lap kernels {info.lap.ms_total:.1f}", flush=True)
for rep in range(3):
    w, ctx = wall(lambda: ExpressionContext(sc16, st16, already_normalized=False), reps=1)
    print(f"context (upload + transforms of both matrices, uint16), rep {rep}: {w * 1e3:.1f} ms", flush=True)
    if rep < 2:
        ctx.close()
ctx.assign_chunk(np.arange(C), slots)
w, (m2, t2, i2) = wall(lambda: ctx.assign_chunk(np.arange(C), slots, return_info=True))
print(f"one chunk = the whole problem on the resident operands: wall {w * 1e3:.1f} ms | gather {i2.ms_standardize:.2f} gemm {i2.ms_gemm:.2f} lap kernels {i2.lap.ms_total:.2f} "
      f"(colred {i2.lap.ms_colred:.2f} cache {i2.lap.ms_cache:.2f} arr {i2.lap.ms_arr:.2f} aug {i2.lap.ms_aug:.2f})")
ctx.close()
cost, N, ld, gemm_ms = common.pearson_cost_device(sc16, st16, np.ones(S, np.int64), 0, already_normalized=False)
loc = np.repeat(np.arange(S), slots).astype(np.int32)
lap_solve_rows(None, loc, device_ptr=cost.ptr, nu=S, ld=ld)
w, g = wall(lambda: lap_solve_rows(None, loc, device_ptr=cost.ptr, nu=S, ld=ld, return_info=True))
i = g["info"]
print(f"the LAP alone on the resident cost (row map): wall {w * 1e3:.2f} ms, kernels {i.ms_total:.2f} (colred {i.ms_colred:.2f} cache {i.ms_cache:.2f} arr {i.ms_arr:.2f} aug {i.ms_aug:.2f}); "
      f"split GEMM alone {gemm_ms:.2f} ms")
cost.free()
