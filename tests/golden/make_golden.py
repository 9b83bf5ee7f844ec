"""Generates the golden vectors under tests/golden/ by IMPORTING the reference (read-only, at
/root/reference) in this container.  Only arrays are committed; no reference source is copied.

Run:  python tests/golden/make_golden.py        (needs /root/reference; not needed on the GPU box)

`import cytospace` fails as shipped because scanpy / datatable / ortools are not installed and
are imported at module top (common.py:3,5; linear_assignment_solvers.py:6).  None of them is
touched on the hot path, so empty stub modules are inserted first.
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

from cytospace.common import normalize_data, matrix_correlation_pearson, matrix_correlation_spearman  # noqa: E402
from cytospace.linear_assignment_solvers import calculate_cost, call_solver  # noqa: E402
from cytospace.cytospace import (solve_linear_assignment_problem, partition_indices,  # noqa: E402
                                 estimate_cell_number_RNA_reads, get_cell_type_fraction)
from scipy.optimize import linear_sum_assignment  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def exact_solver_lapjv_shape(cost):
    """An exact solver with lapjv's return shape: (row_ind, col_ind, extra)."""
    r, c = linear_sum_assignment(np.asarray(cost, dtype=np.float64))
    row_ind = np.empty(len(r), np.int64)
    row_ind[r] = c
    col_ind = np.empty(len(r), np.int64)
    col_ind[c] = r
    return row_ind, col_ind, None


def synth_expression(G, C, S, slots, seed):
    rng = np.random.default_rng(seed)
    m = rng.lognormal(0, 1.5, G)
    K = 6
    mult = rng.lognormal(0, 0.75, (K, G))
    typ = rng.integers(0, K, C)
    rate = 0.3 * m[None, :] * mult[typ]
    sc = rng.poisson(rate).T.astype(np.float64)          # G x C
    st = np.zeros((G, S))
    perm = rng.permutation(C)
    pos = 0
    for s_ in range(S):
        k = int(slots[s_])
        cells = perm[pos:pos + k] if k > 0 else perm[:1]
        pos += k
        st[:, s_] = rng.poisson(rate[cells].sum(0))
    return sc, st


def main():
    rng = np.random.default_rng(20260928)

    # GV1 normalize_data (common/common.py:142-147): all-zero column, a NaN, integer counts
    counts = rng.poisson(3.0, (64, 12)).astype(np.float64)
    counts[:, 4] = 0.0
    counts[7, 2] = np.nan
    np.savez(os.path.join(OUT, "gv1_normalize.npz"), counts=counts, out=normalize_data(counts.copy()))
    counts_int = rng.poisson(5.0, (40, 9)).astype(np.int64)
    np.savez(os.path.join(OUT, "gv1b_normalize_int.npz"), counts=counts_int, out=normalize_data(counts_int.copy()))

    # GV2 matrix_correlation_pearson (common/common.py:190-199)
    sc, st = synth_expression(128, 48, 16, np.full(16, 3), 11)
    scn, stn = normalize_data(sc.copy()), normalize_data(st.copy())
    np.savez(os.path.join(OUT, "gv2_pearson.npz"), sc_norm=scn, st_norm=stn, corr=matrix_correlation_pearson(scn, stn))
    np.savez(os.path.join(OUT, "gv2b_spearman.npz"), sc_norm=scn, st_norm=stn, corr=matrix_correlation_spearman(scn, stn))

    # GV3 calculate_cost (linear_assignment_solvers.py:42-69): slots with 0, 1 and >1
    slots = np.array([0, 1, 3, 2, 0, 5, 1, 4, 2, 6], dtype=np.int64)
    C = int(slots.sum())
    sc, st = synth_expression(96, C, len(slots), slots, 13)
    scn, stn = normalize_data(sc.copy()), normalize_data(st.copy())
    dist, loc = calculate_cost(scn, stn, slots, "lapjv", "Pearson_correlation")
    np.savez(os.path.join(OUT, "gv3_calculate_cost.npz"), sc_norm=scn, st_norm=stn, slots=slots,
             distance_repeat=dist, location_repeat=loc)

    # GV4 the tie-breaking perturbation stream (cytospace.py:325-327)
    np.random.seed(1)
    r1 = np.random.rand(4, 4)
    np.random.seed(7)
    r7 = np.random.rand(3, 5)
    np.savez(os.path.join(OUT, "gv4_rand.npz"), seed1_4x4=r1, seed7_3x5=r7)

    # GV5 solve_linear_assignment_problem (cytospace.py:304-351) with an injected exact solver
    gv5 = {}
    for tag, slots in (("visium", np.full(12, 5, dtype=np.int64)), ("single", np.ones(60, dtype=np.int64))):
        for seed in (1, 7):
            C = int(slots.sum())
            sc, st = synth_expression(200, C, len(slots), slots, 100 + seed)
            scn, stn = normalize_data(sc.copy()), normalize_data(st.copy())
            mapped, pidx = solve_linear_assignment_problem(scn, stn, slots, "lapjv", exact_solver_lapjv_shape, seed,
                                                           "Pearson_correlation", process_idx=3)
            assert pidx == 3
            key = f"{tag}_s{seed}"
            gv5[key + "_sc_norm"] = scn
            gv5[key + "_st_norm"] = stn
            gv5[key + "_slots"] = slots
            gv5[key + "_mapped"] = np.asarray(mapped, dtype=np.int64)
    np.savez(os.path.join(OUT, "gv5_solve_lap.npz"), **gv5)

    # GV6 partition_indices (cytospace.py:150-209): the docstring examples + a seeded shuffle
    gv6 = {}
    p = partition_indices(np.arange(0, 2500), split_by_interval_int=1000, shuffle=False)
    gv6["ex1_lens"] = np.array([len(x) for x in p]); gv6["ex1_first"] = np.array([x[0] for x in p])
    p = partition_indices(np.arange(0, 1800), split_by_category_list=np.array([500, 1000, 300]),
                          split_by_interval_int=400, shuffle=False)
    gv6["ex2_lens"] = np.array([len(x) for x in p]); gv6["ex2_first"] = np.array([x[0] for x in p])
    p = partition_indices(np.arange(0, 8000), split_by_category_list=np.array([3000, 5000]),
                          split_by_interval_int=2000, shuffle=False)
    gv6["ex3_lens"] = np.array([len(x) for x in p]); gv6["ex3_first"] = np.array([x[0] for x in p])
    np.random.seed(5)
    p = partition_indices(np.arange(0, 37), split_by_interval_int=10, shuffle=True)
    gv6["shuf_concat"] = np.concatenate(p); gv6["shuf_lens"] = np.array([len(x) for x in p])
    np.savez(os.path.join(OUT, "gv6_partition.npz"), **gv6)

    # GV8 LAP known answers from an independent exact solver (scipy), unique optimum checked by
    # re-solving a +-1 ulp perturbed copy
    gv8 = {}
    for n in (1, 2, 3, 7, 64, 256):
        seed = n
        while True:
            c = np.random.default_rng(seed).random((n, n)).astype(np.float32)
            r, col = linear_sum_assignment(c.astype(np.float64))
            pert = np.nextafter(c, c + np.where(np.random.default_rng(seed + 1).random((n, n)) < 0.5, -1, 1).astype(np.float32))
            r2, col2 = linear_sum_assignment(pert.astype(np.float64))
            if np.array_equal(col, col2):
                break
            seed += 1000
        colsol = np.empty(n, np.int64); colsol[col] = r
        gv8[f"n{n}_cost"] = c
        gv8[f"n{n}_rowsol"] = col.astype(np.int64)
        gv8[f"n{n}_colsol"] = colsol
        gv8[f"n{n}_total"] = np.float64(c.astype(np.float64)[r, col].sum())
    # duplicated rows: only the spot-level answer and the total are pinned
    base = -np.random.default_rng(77).random((8, 40)).astype(np.float32)
    dup = np.repeat(base, 5, axis=0)
    r, col = linear_sum_assignment(dup.astype(np.float64))
    gv8["dup_cost"] = dup
    gv8["dup_total"] = np.float64(dup.astype(np.float64)[r, col].sum())
    gv8["dup_spot_of_col"] = (np.argsort(col) // 5).astype(np.int64)   # spot (row // 5) given to each column
    np.savez(os.path.join(OUT, "gv8_lap.npz"), **gv8)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
