"""Full-size parity at config c3: device-built Visium-like cost (50k x 50k, every spot row x10) solved by the
HIP path and by the CPU oracle; rowsol/colsol/u/v must be bit-identical.  Usage: check_c3_parity.py [G C S]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from c3_pipeline import synth
from cytospace_amd import common
from cytospace_amd.lap import lap_solve
from oracle.jv import jv_oracle

G, C, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20000, 50000, 5000)
sc, st, slots = synth(G, C, S)
cost, N, ld, gemm_ms = common.pearson_cost_device(sc.astype(np.float64), st.astype(np.float64), slots, already_normalized=False)
t = time.time(); g = lap_solve(None, np.float32, return_info=True, device_ptr=cost.ptr, n=N, ld=ld); tg = time.time() - t
c = np.ascontiguousarray(cost.to_numpy((N, ld), np.float32)[:, :C]); cost.free()
t = time.time(); o = jv_oracle(c, np.float32); to = time.time() - t
ok = all(np.array_equal(g[k], o[k]) for k in ("rowsol", "colsol", "u", "v"))
i = g["info"]
print(f"c3 parity G={G} N={N}: bit-identical={ok} scans equal={i.row_scans == o['stats'].row_scans} "
      f"gpu kernels {i.ms_total:.0f} ms (wall {tg:.2f}s) oracle {to:.1f}s gemm {gemm_ms:.1f} ms", flush=True)
