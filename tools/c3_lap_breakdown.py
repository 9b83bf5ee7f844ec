import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.cytospace import assign_pearson
from tools import instances
sc, st, slots = instances.synth_expression(20000, 50000, 5000, seed=1)
assign_pearson(sc, st, slots, already_normalized=False)
m, t, info = assign_pearson(sc, st, slots, already_normalized=False, return_info=True)
l = info.lap
print("LAP", {k: round(getattr(l, k), 2) for k in ("ms_colred", "ms_cache", "ms_chain", "ms_arr", "ms_aug", "ms_total")},
      {k: int(getattr(l, k)) for k in ("scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax", "augmentations", "free_after_colred", "free_after_arr1", "free_after_arr2", "dense_refreshes", "aug_dense_scans", "aug_sparse_inits", "row_groups", "hbm_row_reads")})
print("wide", {k: (round(getattr(l, k), 3) if isinstance(getattr(l, k), float) else int(getattr(l, k))) for k in (
    "wide", "wide_rounds", "wide_retired", "wide_trivial", "wide_aug_rounds", "wide_aug_settled", "wide_verify_passes", "wide_aug_launches", "wide_dense_aug",
    "wide_ms_aug_rounds", "wide_ms_aug_verify", "wide_ms_aug_finish", "wide_ms_aug_trivial", "wide_par_batches", "path_hops")})
