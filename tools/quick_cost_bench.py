"""Scratch timing of the MFMA cost GEMM and of batched concurrent LAPs (developer tool)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd import _lib, common
from cytospace_amd.lap import lap_solve_batch, lap_solve

def gemm(G, S, C, slots_per):
    rng = np.random.default_rng(1)
    sc = rng.random((G, C), dtype=np.float32)
    st = rng.random((G, S), dtype=np.float32)
    slots = np.full(S, slots_per, np.int64)
    zsc = common.StandardizedMatrix(sc, True); zst = common.StandardizedMatrix(st, True)
    N = int(slots.sum()); ld = -(-C // 4) * 4
    cost = _lib.DeviceBuffer(N * ld * 4)
    ms = ctypes.c_double()
    for _ in range(3):
        _lib.check(_lib.lib().cyto_cost_pearson(zst.Gpad, S, C, zst.buf.ptr, zst.ld, zsc.buf.ptr, zsc.ld, slots.ctypes.data,
                                                cost.ptr, ld, ctypes.byref(ms), 0, None))
    fl = 2.0 * zst.Gpad * S * C
    print(f"GEMM G={G} S={S} C={C} slots={slots_per}: {ms.value:.3f} ms  {fl/ms.value/1e9:.1f} TFLOP/s  (write {N*C*4/1e9:.2f} GB)", flush=True)

def batch(nb, n, conc):
    costs = [np.random.default_rng(1000 + b).random((n, n)).astype(np.float32) for b in range(nb)]
    t = time.perf_counter(); r1 = lap_solve(costs[0], return_info=True); t1 = time.perf_counter() - t
    t = time.perf_counter(); res = lap_solve_batch(costs, max_concurrent=conc); tb = time.perf_counter() - t
    print(f"batch nb={nb} n={n} conc={conc}: single {t1*1e3:.0f} ms (kernels {r1['info'].ms_total:.0f} ms); batch wall {tb*1e3:.0f} ms "
          f"-> {nb*n/tb:.0f} assignments/s (incl. H2D of {nb*n*n*4/1e9:.1f} GB)", flush=True)

if __name__ == "__main__" and "--resident" not in sys.argv:
    gemm(2048, 1024, 4096, 4)
    gemm(8192, 4096, 16384, 4)
    gemm(20000, 5000, 20000, 4)
    batch(16, 4000, 16)
    batch(32, 6000, 32)
    batch(32, 10000, 32)


def batch_resident(nb, n, conc):
    L = _lib.lib()
    costs = [np.random.default_rng(1000 + b).random((n, n)).astype(np.float32) for b in range(nb)]
    bufs = [_lib.DeviceBuffer.from_numpy(c) for c in costs]
    ns = (ctypes.c_int * nb)(*([n] * nb)); lds = (ctypes.c_int64 * nb)(*([n] * nb))
    cptr = (ctypes.c_void_p * nb)(*[b.ptr for b in bufs])
    cols = [np.empty(n, np.int32) for _ in range(nb)]
    colp = (ctypes.c_void_p * nb)(*[c.ctypes.data for c in cols])
    infos = (_lib.LapInfo * nb)(); status = (ctypes.c_int * nb)(); totals = (ctypes.c_double * nb)()
    for rep in range(2):
        t = time.perf_counter()
        st = L.cyto_lap_batch_f32(nb, ns, cptr, lds, 1, None, colp, None, None, totals, infos, status, conc, 0)
        tb = time.perf_counter() - t
    _lib.check(st)
    ks = [infos[b].ms_total for b in range(nb)]
    print(f"resident batch nb={nb} n={n} conc={conc}: wall {tb*1e3:.0f} ms -> {nb*n/tb:.0f} assignments/s; "
          f"per-LAP kernel ms min/mean/max {min(ks):.0f}/{sum(ks)/nb:.0f}/{max(ks):.0f}", flush=True)
    for b in bufs: b.free()


if __name__ == "__main__" and "--resident" in sys.argv:
    batch_resident(8, 10000, 8)
    batch_resident(32, 10000, 32)
    batch_resident(64, 10000, 64)
    batch_resident(128, 5000, 128)
