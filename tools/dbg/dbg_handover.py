import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from cytospace_amd.lap import lap_solve
rng = np.random.default_rng(5)
n, types = 5400, 6
prof = rng.normal(size=(types, 64)).astype(np.float32)
rows = prof[rng.integers(0, types, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
cols = prof[rng.integers(0, types, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
c = -(rows @ cols.T).astype(np.float32)
g = lap_solve(c, np.float32, return_info=True)
i = g["info"]
print("ok", i.ms_total, i.scans_aug_relax, i.aug_dense_scans, i.augmentations, i.aug_handover, i.scans_arr, i.dense_refreshes)
