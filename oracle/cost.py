"""numpy (float64) restatement of the reference's cost build and per-chunk solve.

TEST INFRASTRUCTURE ONLY -- never imported by cytospace_amd/.  Each function cites the reference
lines it follows (paths relative to /root/reference).  Pinned by tests/golden/gv1..gv6 (arrays
produced by importing the reference itself: tests/golden/make_golden.py).
"""
import numpy as np

from .jv import jv_oracle


def normalize_data(data):
    """cytospace/common/common.py:142-147 -- per-column CPM then log2(x+1), NaN -> 0.

    nan_to_num first; every column is scaled to 1e6 total (axis 0 = genes); an all-zero
    column gives 0/0 = NaN which the final nan_to_num turns into 0."""
    x = np.nan_to_num(np.asarray(data)).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        x *= 1e6 / x.sum(axis=0, dtype=np.float64)
    x = np.log2(x + 1.0)
    return np.nan_to_num(x)


def matrix_correlation_pearson(v1, v2):
    """cytospace/common/common.py:190-199 -- corr[s, c] between column s of v2 and column c of v1.

    Population std (ddof=0); no zero-variance guard (NaN/Inf propagate, as in the reference)."""
    if v1.shape[0] != v2.shape[0]:
        raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                         "ST and scRNA data must have the same genes")
    n = v1.shape[0]
    sums = np.multiply.outer(v2.sum(0), v1.sum(0))
    stds = np.multiply.outer(v2.std(0), v1.std(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (v2.T.dot(v1) - sums / n) / stds / n


def rank_columns(v):
    """pandas.DataFrame(v).rank() with its defaults (ascending, ties averaged, 1-based), per column."""
    v = np.asarray(v, dtype=np.float64)
    out = np.empty_like(v)
    for c in range(v.shape[1]):
        col = v[:, c]
        order = np.argsort(col, kind="stable")
        s = col[order]
        starts = np.flatnonzero(np.r_[True, s[1:] != s[:-1]])
        ends = np.r_[starts[1:], len(s)]
        r = np.empty(len(s))
        for a, b in zip(starts, ends):
            r[a:b] = 0.5 * (a + 1 + b)
        out[order, c] = r
    return out


def matrix_correlation_spearman(v1, v2):
    """cytospace/common/common.py:202-215 -- the Pearson formula on the per-column ranks."""
    if v1.shape[0] != v2.shape[0]:
        raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                         "ST and scRNA data must have the same genes")
    return matrix_correlation_pearson(rank_columns(v1), rank_columns(v2))


def euclidean_cost(sc_norm, st_norm):
    """np.transpose(scipy cdist(sc.T, st.T, 'euclidean')) (linear_assignment_solvers.py:58-59): spots x cells."""
    a = np.asarray(st_norm, dtype=np.float64).T[:, None, :]
    b = np.asarray(sc_norm, dtype=np.float64).T[None, :, :]
    return np.sqrt(((a - b) ** 2).sum(-1))


def calculate_cost(sc_norm, st_norm, slots, solver_method="lapjv", distance_metric="Pearson_correlation"):
    """cytospace/linear_assignment_solvers/linear_assignment_solvers.py:42-69, the non-CSPR branch:
    cost (spots x cells) = -Pearson, -Spearman or Euclidean distance; rows repeated slots[s] times in spot order."""
    if solver_method == "lap_CSPR":
        raise NotImplementedError("oracle restates the lapjv branch only")
    if distance_metric == "Pearson_correlation":
        cost = -matrix_correlation_pearson(sc_norm, st_norm)
    elif distance_metric == "Spearman_correlation":
        cost = -matrix_correlation_spearman(sc_norm, st_norm)
    elif distance_metric == "Euclidean":
        cost = euclidean_cost(sc_norm, st_norm)
    else:
        raise ValueError(distance_metric)
    location_repeat = np.repeat(np.arange(len(slots)), slots).astype(int)
    return cost[location_repeat, :], location_repeat


def perturb(distance_repeat, seed):
    """cytospace/cytospace.py:325-327 -- legacy RandomState stream, row-major fill."""
    np.random.seed(seed)
    return distance_repeat + 1e-16 * np.random.rand(distance_repeat.shape[0], distance_repeat.shape[1])


def solve_linear_assignment_problem(sc_norm, st_norm, slots, seed=1, dtype=np.float32, process_idx=None):
    """cytospace/cytospace.py:304-351 for solver_method == 'lapjv': cost -> perturb -> JV ->
    location_repeat[colsol].  `dtype` is the precision the JV solve runs in (lapjv 1.3.14 is
    recalled to use float32 unless force_doubles -- UNVERIFIED)."""
    distance_repeat, location_repeat = calculate_cost(sc_norm, st_norm, slots)
    cost_scaled = perturb(distance_repeat, seed)
    r = jv_oracle(cost_scaled, dtype)
    mapped = location_repeat[r["colsol"]]
    return mapped.tolist(), process_idx


def partition_indices(indices, split_by_category_list=None, split_by_interval_int=None, shuffle=True):
    """cytospace/cytospace.py:150-209 -- break points at category boundaries and every
    `interval` inside a category that is longer than the interval; np.array_split."""
    indices = np.asarray(indices)
    n = len(indices)
    if shuffle:
        np.random.shuffle(indices)
    bps = {0, n}
    if split_by_category_list is not None:
        bps.update(int(b) for b in np.cumsum(split_by_category_list))
    base = sorted(bps)
    if split_by_interval_int is not None:
        for a, b in zip(base[:-1], base[1:]):
            if b - a > split_by_interval_int:
                bps.update(range(a, b, split_by_interval_int))
    cuts = sorted(bps)[1:-1]
    return np.array_split(indices, cuts)
