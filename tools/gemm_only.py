"""One c3-sized cost GEMM (20 000 genes x 5 000 spots x 50 000 cells, 10 slots per spot), for profiling."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd import _lib, common
G, S, C, k = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (20000, 5000, 50000, 10)
rng = np.random.default_rng(1)
zsc = common.StandardizedMatrix(rng.random((G, C), dtype=np.float32), True)
zst = common.StandardizedMatrix(rng.random((G, S), dtype=np.float32), True)
slots = np.full(S, k, np.int64)
N = int(slots.sum()); ld = -(-C // 4) * 4
cost = _lib.DeviceBuffer(N * ld * 4)
ms = ctypes.c_double()
for _ in range(3):
    _lib.check(_lib.lib().cyto_cost_pearson(zst.Gpad, S, C, zst.buf.ptr, zst.ld, zsc.buf.ptr, zsc.ld, slots.ctypes.data,
                                            cost.ptr, ld, ctypes.byref(ms), 0, None))
print(f"GEMM G={G} S={S} C={C}: {ms.value:.2f} ms, {2.0 * zst.Gpad * S * C / ms.value / 1e9:.1f} TFLOP/s")
