"""The `--solver-method` plug-in surface of the reference
(/root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:11-69) with one more
method, "lapjv_hip": the MI355X Jonker-Volgenant solver.  Same function names and signatures.
"""
import time

import numpy as np

from . import common
from .lap import lapjv_hip

SOLVER_METHODS = ("lapjv", "lapjv_compat", "lap_CSPR", "lapjv_hip")   # argparse `choices` (argument_parser.py:69-71)


def import_solver(solver_method):
    """linear_assignment_solvers.py:11-31.  "lapjv_hip" resolves to the in-tree HIP solver; the
    reference's own names still resolve to the third-party packages when those are installed."""
    if solver_method == "lapjv_hip":
        return lapjv_hip
    try:
        if solver_method == "lapjv_compat":
            from lap import lapjv
            return lapjv
        elif solver_method == "lapjv":
            from lapjv import lapjv
            return lapjv
        raise NotImplementedError(f"The solver {solver_method} is not a supported solver for the shortest "
                                  "augmenting path method, choose between 'lapjv_hip', 'lapjv' and 'lapjv_compat'.")
    except ModuleNotFoundError:
        raise ModuleNotFoundError("The Python package containing the solver_method option you have chosen "
                                  f"{solver_method} was not found; 'lapjv_hip' needs no extra package.")


def call_solver(solver, solver_method, cost_scaled):
    """linear_assignment_solvers.py:34-40: normalise the return tuple to y (row of each column)."""
    if solver_method == "lapjv_compat":
        _, _, y = solver(cost_scaled)
    elif solver_method in ("lapjv", "lapjv_hip"):
        _, y, _ = solver(cost_scaled)
    else:
        raise ValueError("Invalid solver_method provided")
    return y


def calculate_cost(expressions_tpm_scRNA_log, expressions_tpm_st_log, cell_number_to_node_assignment,
                   solver_method, distance_metric):
    """linear_assignment_solvers.py:42-69 for the shortest-augmenting-path solvers (Pearson_correlation,
    Spearman_correlation, Euclidean): returns (distance_repeat [N x C float32], location_repeat [N int])."""
    if solver_method == "lap_CSPR":
        raise NotImplementedError("the HIP cost build covers the lapjv-family branch (cost spots x cells)")
    if distance_metric not in common.METRICS:
        # the reference leaves `cost` unbound here and dies with UnboundLocalError; say what is wrong instead
        raise ValueError(f"unknown distance_metric {distance_metric!r}")
    print("Building cost matrix ...")
    t0 = time.perf_counter()
    slots = np.asarray(cell_number_to_node_assignment)
    cost, N, ld, _ = common.pearson_cost_device(expressions_tpm_scRNA_log, expressions_tpm_st_log, slots,
                                                metric=distance_metric)
    C = np.asarray(expressions_tpm_scRNA_log).shape[1]
    distance_repeat = np.ascontiguousarray(cost.to_numpy((N, ld), np.float32)[:, :C])
    cost.free()
    location_repeat = np.repeat(np.arange(len(slots)), slots).astype(int)
    print(f"Time to build cost matrix: {round(time.perf_counter() - t0, 2)} seconds")
    return distance_repeat, location_repeat
