"""Golden vectors for the functions upstream of apply_linear_assignment (cytospace/cytospace.py:116-147, 212-301):
estimate_cell_number_RNA_reads, get_cell_type_fraction, sample_single_cells -- the reference itself, imported
read-only from /root/reference; only arrays are committed.

Run:  python tests/golden/make_golden_upstream.py
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

from cytospace.cytospace import estimate_cell_number_RNA_reads, get_cell_type_fraction, sample_single_cells  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    rng = np.random.default_rng(707)
    gv = {}
    # GV7a: cells per spot
    st = rng.poisson(rng.lognormal(0, 1, (60, 1)) * rng.uniform(0.2, 3.0, (1, 25))).astype(float)
    st_df = pd.DataFrame(st, index=[f"g{i}" for i in range(60)], columns=[f"s{i}" for i in range(25)])
    gv["st_counts"] = st
    gv["cells_per_spot_mean5"] = estimate_cell_number_RNA_reads(st_df, 5)
    gv["cells_per_spot_mean20"] = estimate_cell_number_RNA_reads(st_df, 20)
    # GV7b: integer cell numbers per type
    frac = np.array([[0.31, 0.22, 0.47]])
    frac_df = pd.DataFrame(frac, index=["Fraction"], columns=["TYPE_A", "TYPE_B", "TYPE_C"])
    gv["fractions"] = frac
    gv["numbers_101"] = get_cell_type_fraction(101, frac_df.copy()).values[:, 0].astype(np.int64)
    gv["numbers_7"] = get_cell_type_fraction(7, frac_df.copy()).values[:, 0].astype(np.int64)
    # GV7c: the sampler, both methods, types with too few and with enough cells
    G, C = 8, 30
    sc = rng.poisson(2.0, (G, C)).astype(float)
    labels = np.array(["TYPE_A"] * 12 + ["TYPE_B"] * 5 + ["TYPE_C"] * 13)
    sc_df = pd.DataFrame(sc, index=[f"g{i}" for i in range(G)], columns=[f"CELL_{i}" for i in range(C)])
    ct_df = pd.DataFrame(labels, index=sc_df.columns, columns=["CellType"])
    need = pd.DataFrame([7, 9, 13], index=["TYPE_A", "TYPE_B", "TYPE_C"], columns=["Fraction"])
    gv["sc_counts"] = sc; gv["sc_labels"] = labels; gv["need"] = need.values[:, 0]
    for seed in (1, 4):
        d = sample_single_cells(sc_df, ct_df, need, "duplicates", seed)
        gv[f"dup_s{seed}_cells"] = np.array([int(x.split("_")[1]) for x in d.columns])
        p = sample_single_cells(sc_df, ct_df, need, "place_holders", seed)
        gv[f"ph_s{seed}_names"] = np.array(list(p.columns))
        gv[f"ph_s{seed}_values"] = p.to_numpy()
    np.savez(os.path.join(OUT, "gv7_upstream.npz"), **gv)
    print("wrote gv7_upstream.npz")
