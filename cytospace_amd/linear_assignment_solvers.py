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
    # (lap_CSPR, l.46-52: the reference contracts the operands the other way round and transposes the result -- the same
    #  spots x cells matrix up to the last bit of its float64 dot products; the device cost is built once, the usual way)
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


def match_solution(cost):
    """linear_assignment_solvers.py:72-96 (`lap_CSPR`): minimum-cost perfect matching of an INTEGER cost matrix
    (workers x tasks, as nested lists or an array); returns the reference's assignment_mat: [:, 0] = task of worker i
    (RightMate), [:, 1] = its cost.  The reference hands the matrix to OR-tools' cost-scaling push-relabel
    LinearSumAssignment and leaves out arcs whose cost is 0; here the same integer problem is solved exactly by the HIP
    Jonker-Volgenant kernels (float32 arithmetic is exact on these integers: |10^6 d + 10 rand + 1| < 2^24, and every JV
    quantity is a difference of costs and prices inside that range) and the optimality is VERIFIED in integer arithmetic
    from the duals (any failure falls back to the float64 solve).  An optimal matching of an integer problem need not be
    unique: like OR-tools' it is one of the optima, with the same total."""
    c = np.asarray(cost)
    if c.ndim != 2 or c.shape[0] != c.shape[1]:
        raise ValueError("The assignment failed")              # OR-tools reports INFEASIBLE for a non-square problem
    ci = c.astype(np.int64)
    big = int(np.abs(ci).max()) * 4 + 4
    work = np.where(ci == 0, big, ci)                           # `if cost[worker][task]:` -- a zero-cost arc is never added
    exact32 = big * 2 < (1 << 24)
    r = None
    if exact32:
        from .lap import lap_solve
        r = lap_solve(work.astype(np.float32), np.float32)
        u, v = np.rint(r["u"]).astype(np.int64), np.rint(r["v"]).astype(np.int64)
        red = work - u[:, None] - v[None, :]
        rows = np.arange(len(work))
        if red.min() < 0 or np.any(red[rows, r["rowsol"]] != 0):
            r = None                                            # (not observed) the duals do not certify it: solve in float64
    if r is None:
        from .lap import lap_solve
        r = lap_solve(work.astype(np.float64), np.float64)
    rowsol = r["rowsol"].astype(np.int64)
    if np.any(work[np.arange(len(work)), rowsol] >= big):
        print('No assignment is possible.')
        return np.zeros((len(work), 2))
    out = np.zeros((len(work), 2))
    out[:, 0] = rowsol
    out[:, 1] = ci[np.arange(len(ci)), rowsol]
    print('Total cost = ', int(out[:, 1].sum()))
    print()
    return out
