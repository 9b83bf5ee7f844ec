"""Golden vectors for apply_linear_assignment (cytospace/cytospace.py:354-469): the reference itself, imported
read-only from /root/reference, run with an injected exact solver in its two chunked modes.  Only arrays are
committed (inputs, chunk index lists, and the resulting (cell id, spot coordinates) pairs).

Run:  python tests/golden/make_golden_apply.py
"""
import os
import sys
import types

import numpy as np
import pandas as pd

for name in ("scanpy", "datatable", "ortools", "ortools.graph", "ortools.graph.pywrapgraph"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ortools"].graph = sys.modules["ortools.graph"]
sys.modules["ortools.graph"].pywrapgraph = sys.modules["ortools.graph.pywrapgraph"]
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cytospace.cytospace import apply_linear_assignment, partition_indices  # noqa: E402
from make_golden import synth_expression, exact_solver_lapjv_shape  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    gv = {}
    G = 120
    # --single-cell mode: slots == 1, ST split with the scRNA chunks
    S = C = 90
    slots = np.ones(S, dtype=np.int64)
    sc, st = synth_expression(G, C, S, slots, 201)
    sc_df = pd.DataFrame(sc, index=[f"g{i}" for i in range(G)], columns=[f"c{i}" for i in range(C)])
    st_df = pd.DataFrame(st, index=sc_df.index, columns=[f"s{i}" for i in range(S)])
    coords = pd.DataFrame({"row": np.arange(S) // 10, "col": np.arange(S) % 10}, index=st_df.columns)
    idx_sc = partition_indices(np.arange(C), split_by_interval_int=40, shuffle=False)
    idx_st = partition_indices(np.arange(S), split_by_interval_int=40, shuffle=False)
    loc, ids = apply_linear_assignment(sc_df, st_df, coords, slots, "lapjv", exact_solver_lapjv_shape, 1,
                                       "Pearson_correlation", 2, idx_sc, index_st_list=idx_st)
    gv["sc_counts"] = sc; gv["sc_st_counts"] = st; gv["sc_slots"] = slots
    gv["sc_idx_sc"] = np.concatenate(idx_sc); gv["sc_idx_sc_lens"] = np.array([len(x) for x in idx_sc])
    gv["sc_idx_st"] = np.concatenate(idx_st); gv["sc_idx_st_lens"] = np.array([len(x) for x in idx_st])
    gv["sc_out_cell"] = np.array([int(x[1:]) for x in ids]); gv["sc_out_rowcol"] = loc.to_numpy()
    # --sampling-sub-spots mode: full ST for every chunk, per-chunk slot counts
    S, per = 12, 10
    C = S * per
    slots = np.full(S, per, dtype=np.int64)
    sc, st = synth_expression(G, C, S, slots, 202)
    sc_df = pd.DataFrame(sc, index=[f"g{i}" for i in range(G)], columns=[f"c{i}" for i in range(C)])
    st_df = pd.DataFrame(st, index=sc_df.index, columns=[f"s{i}" for i in range(S)])
    coords = pd.DataFrame({"row": np.arange(S) // 4, "col": np.arange(S) % 4}, index=st_df.columns)
    idx_sc = partition_indices(np.arange(C), split_by_interval_int=40, shuffle=False)
    sub = [np.array([4, 3, 3, 4, 3, 3, 4, 3, 3, 4, 3, 3]), np.array([3, 4, 3, 3, 4, 3, 3, 4, 3, 3, 4, 3]),
           np.array([3, 3, 4, 3, 3, 4, 3, 3, 4, 3, 3, 4])]
    loc, ids = apply_linear_assignment(sc_df, st_df, coords, slots, "lapjv", exact_solver_lapjv_shape, 1,
                                       "Pearson_correlation", 2, idx_sc, subsampled_cell_number_to_node_assignment_list=sub)
    gv["ss_counts"] = sc; gv["ss_st_counts"] = st; gv["ss_slots"] = slots
    gv["ss_idx_sc"] = np.concatenate(idx_sc); gv["ss_idx_sc_lens"] = np.array([len(x) for x in idx_sc])
    gv["ss_sub"] = np.stack(sub)
    gv["ss_out_cell"] = np.array([int(x[1:]) for x in ids]); gv["ss_out_rowcol"] = loc.to_numpy()
    np.savez(os.path.join(OUT, "gv11_apply_linear_assignment.npz"), **gv)
    print("wrote gv11_apply_linear_assignment.npz", len(ids))
