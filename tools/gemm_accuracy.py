"""Max |device cost - float64 reference| of the cost GEMM at c3 size on a sample of spot rows (the FOLD knob of pearson_gemm)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd import _lib, common
from oracle import cost as ocost
from tools import instances
G, C, S = 20000, 50000, 5000
sc, st, slots = instances.synth_expression(G, C, S, seed=1)
cost, N, ld, ms = common.pearson_cost_device(sc, st, np.ones(S, np.int64), already_normalized=False)
rs = np.random.default_rng(0).choice(S, 96, replace=False)
stn = ocost.normalize_data(st[:, rs].astype(np.float64))
ref = np.empty((len(rs), C))
for lo in range(0, C, 5000):
    ref[:, lo:lo + 5000] = -ocost.matrix_correlation_pearson(ocost.normalize_data(sc[:, lo:lo + 5000].astype(np.float64)), stn)
rows = np.empty((len(rs), ld), np.float32)
for k, s_ in enumerate(rs):
    _lib.check(_lib.lib().cyto_memcpy_d2h(rows[k].ctypes.data, cost.ptr + int(s_) * ld * 4, ld * 4, 0))
err = rows[:, :C].astype(np.float64) - ref
print(f"GEMM {ms:.2f} ms; max |err| {np.abs(err).max():.3e}, mean err {err.mean():+.3e}, entries above 1e-6: {(np.abs(err) > 1e-6).sum()} of {err.size}, above 2e-6: {(np.abs(err) > 2e-6).sum()}")
