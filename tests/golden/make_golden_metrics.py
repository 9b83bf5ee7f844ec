"""Golden vectors for the other distance metrics of calculate_cost (linear_assignment_solvers.py:53-59), produced by
IMPORTING the reference (read-only, /root/reference) in this container; only arrays are committed.
Separate from make_golden.py so that the earlier fixtures are not rewritten.

Run:  python tests/golden/make_golden_metrics.py
"""
import os
import sys
import types

import numpy as np

for name in ("scanpy", "datatable", "ortools", "ortools.graph", "ortools.graph.pywrapgraph"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ortools"].graph = sys.modules["ortools.graph"]
sys.modules["ortools.graph"].pywrapgraph = sys.modules["ortools.graph.pywrapgraph"]
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cytospace.common import normalize_data  # noqa: E402
from cytospace.linear_assignment_solvers import calculate_cost  # noqa: E402
from cytospace.cytospace import solve_linear_assignment_problem  # noqa: E402
from make_golden import synth_expression, exact_solver_lapjv_shape  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    # GV9: calculate_cost with Spearman_correlation and Euclidean; slots with 0, 1 and > 1; ties in every column
    slots = np.array([2, 0, 1, 4, 3, 0, 2, 5, 1], dtype=np.int64)
    C = int(slots.sum())
    sc, st = synth_expression(150, C, len(slots), slots, 31)
    scn, stn = normalize_data(sc.copy()), normalize_data(st.copy())
    out = dict(sc_norm=scn, st_norm=stn, slots=slots)
    for tag, metric in (("spearman", "Spearman_correlation"), ("euclidean", "Euclidean")):
        dist, loc = calculate_cost(scn, stn, slots, "lapjv", metric)
        out[tag + "_distance_repeat"] = dist
        out[tag + "_location_repeat"] = loc
    np.savez(os.path.join(OUT, "gv9_metrics_cost.npz"), **out)

    # GV10: solve_linear_assignment_problem with the two metrics and an injected exact solver (spot level)
    gv = {}
    slots = np.full(10, 4, dtype=np.int64)
    C = int(slots.sum())
    sc, st = synth_expression(220, C, len(slots), slots, 77)
    scn, stn = normalize_data(sc.copy()), normalize_data(st.copy())
    gv["sc_norm"] = scn; gv["st_norm"] = stn; gv["slots"] = slots
    for tag, metric in (("spearman", "Spearman_correlation"), ("euclidean", "Euclidean")):
        mapped, _ = solve_linear_assignment_problem(scn, stn, slots, "lapjv", exact_solver_lapjv_shape, 1, metric, process_idx=None)
        gv[tag + "_mapped"] = np.asarray(mapped, dtype=np.int64)
    np.savez(os.path.join(OUT, "gv10_metrics_solve.npz"), **gv)
    print("wrote gv9_metrics_cost.npz, gv10_metrics_solve.npz")
