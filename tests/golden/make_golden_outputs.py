"""gv12: the files the reference's own save_results (post_processing.py:10-116) and main_cytospace's unassigned-spot table
(cytospace.py:686-694) write for toy inputs, in both sampling methods.  Run here (imports /root/reference); the fixture is the
INPUT tables (as arrays) and the OUTPUT file texts -- no reference source."""
import io
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

class _Stub(types.ModuleType):          # absent third-party modules (plotting, readers): any attribute is a placeholder
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Stub(self.__name__ + "." + k)


for name in ("scanpy", "datatable", "ortools", "ortools.graph", "ortools.graph.pywrapgraph", "matplotlib", "matplotlib.pyplot",
             "matplotlib.patches", "matplotlib.collections", "matplotlib.colors", "matplotlib.gridspec", "seaborn"):
    try:
        __import__(name)
    except Exception:
        sys.modules[name] = _Stub(name)
sys.path.insert(0, "/root/reference")
from cytospace.post_processing.post_processing import save_results  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def toy(method):
    rng = np.random.default_rng(12)
    G, C, S = 7, 14, 6
    genes = [f"GENE_g{i}" for i in range(G)]
    cells = [f"CELL_c{i}" for i in range(C)]
    types_ = ["TYPE_B", "TYPE_T", "TYPE_Mono"]
    ctd = pd.DataFrame({"CellType": [types_[i % 3] for i in range(C)]}, index=cells)
    expr = pd.DataFrame(rng.poisson(2.0, (G, C)), index=genes, columns=cells)
    coords = pd.DataFrame({"row": np.arange(S) // 3, "col": np.arange(S) % 3}, index=[f"SPOT_s{i}" for i in range(S)])
    picked = [cells[i] for i in (3, 0, 7, 7, 12, 5, 9, 1, 3, 13, 2)]
    if method == "place_holders":
        extra = [f"CELL_{t[5:]}_new_{k + 1}" for k, t in enumerate(["TYPE_B", "TYPE_Mono"])]
        for e in extra:
            expr[e] = rng.poisson(2.0, G)
        picked = picked[:9] + extra
    spots = [coords.index[i] for i in (0, 0, 1, 3, 3, 3, 4, 1, 0, 4, 3)]
    return picked, expr, coords.loc[spots], ctd, coords


def main():
    gv = {}
    for method in ("duplicates", "place_holders"):
        for single in (False, True):
            picked, expr, assigned, ctd, coords = toy(method)
            tag = f"{method}_{'sc' if single else 'spot'}"
            with tempfile.TemporaryDirectory() as d:
                save_results(d, "p_", np.array(picked), expr, assigned, ctd, method, single)
                # cytospace.py:686-694
                unmapped = np.setdiff1d(list(coords.index), list(assigned.index)).tolist()
                ul = coords.loc[unmapped]
                ul.index = ul.index.str.replace("SPOT_", "")
                ul["Number of cells"] = 0
                ul.to_csv(f"{d}/p_unassigned_locations.csv", index=True)
                for root, _, files in os.walk(d):
                    for f in sorted(files):
                        rel = os.path.relpath(os.path.join(root, f), d)
                        gv[f"{tag}::{rel}"] = np.frombuffer(open(os.path.join(root, f), "rb").read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "gv12_outputs.npz"), **gv)
    print("gv12:", len(gv), "files")


if __name__ == "__main__":
    main()
