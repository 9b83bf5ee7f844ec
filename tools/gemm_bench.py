"""c3-sized cost GEMM timing (20 000 genes x 5 000 spots x 50 000 cells, 10 slots) and a resident uniform LAP batch."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd import _lib, common
def gemm(G, S, C, slots_per, reps=5):
    rng = np.random.default_rng(1)
    sc = rng.random((G, C), dtype=np.float32); st = rng.random((G, S), dtype=np.float32)
    slots = np.full(S, slots_per, np.int64)
    zsc = common.StandardizedMatrix(sc, True); zst = common.StandardizedMatrix(st, True)
    N = int(slots.sum()); ld = -(-C // 4) * 4
    cost = _lib.DeviceBuffer(N * ld * 4)
    ms = ctypes.c_double(); best = 1e9
    for _ in range(reps):
        _lib.check(_lib.lib().cyto_cost_pearson(zst.Gpad, S, C, zst.buf.ptr, zst.ld, zsc.buf.ptr, zsc.ld, slots.ctypes.data, cost.ptr, ld, ctypes.byref(ms), 0, None))
        best = min(best, ms.value)
    fl = 2.0 * zst.Gpad * S * C
    print(f"GEMM G={G} S={S} C={C} slots={slots_per}: {best:.3f} ms  {fl/best/1e9:.1f} TFLOP/s", flush=True)
    cost.free()
gemm(20000, 5000, 50000, 10)
gemm(5000, 9270, 10000, 1)
gemm(8192, 4096, 16384, 1)
