"""CPU tests: the oracle (oracle/) against the golden vectors captured from the reference and
against an independent exact LAP solver.  No GPU needed."""
import os

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from oracle import cost as ocost
from oracle.jv import jv_oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_gv1_normalize():
    for f in ("gv1_normalize.npz", "gv1b_normalize_int.npz"):
        d = load(f)
        out = ocost.normalize_data(d["counts"])
        assert out.dtype == np.float64
        np.testing.assert_allclose(out, d["out"], rtol=1e-12, atol=0)
    d = load("gv1_normalize.npz")
    assert np.all(ocost.normalize_data(d["counts"])[:, 4] == 0.0)      # all-zero column -> zeros


def test_gv2_pearson():
    d = load("gv2_pearson.npz")
    c = ocost.matrix_correlation_pearson(d["sc_norm"], d["st_norm"])
    assert c.shape == (16, 48)
    np.testing.assert_allclose(c, d["corr"], rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        ocost.matrix_correlation_pearson(d["sc_norm"][:-1], d["st_norm"])


def test_gv3_calculate_cost():
    d = load("gv3_calculate_cost.npz")
    dist, loc = ocost.calculate_cost(d["sc_norm"], d["st_norm"], d["slots"])
    assert np.array_equal(loc, d["location_repeat"])
    np.testing.assert_allclose(dist, d["distance_repeat"], rtol=0, atol=1e-12)
    assert dist.shape[0] == d["slots"].sum() == dist.shape[1]


def test_gv4_perturbation_stream():
    d = load("gv4_rand.npz")
    z = np.zeros((4, 4))
    assert np.array_equal(ocost.perturb(z, 1), 1e-16 * d["seed1_4x4"])
    assert np.array_equal(ocost.perturb(np.zeros((3, 5)), 7), 1e-16 * d["seed7_3x5"])
    # the perturbation is a no-op once the cost is cast to float32 (|cost| <= 1)
    c = -np.random.default_rng(0).random((50, 50))
    assert np.array_equal(ocost.perturb(c, 1).astype(np.float32), c.astype(np.float32))


@pytest.mark.parametrize("key", ["visium_s1", "visium_s7", "single_s1", "single_s7"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gv5_solve_lap_spot_level(key, dtype):
    d = load("gv5_solve_lap.npz")
    seed = int(key[-1])
    mapped, pidx = ocost.solve_linear_assignment_problem(d[key + "_sc_norm"], d[key + "_st_norm"], d[key + "_slots"],
                                                         seed=seed, dtype=dtype, process_idx=3)
    assert pidx == 3
    assert np.array_equal(np.bincount(mapped, minlength=len(d[key + "_slots"])), d[key + "_slots"])
    assert np.array_equal(np.asarray(mapped), d[key + "_mapped"])


def test_gv6_partition_indices():
    d = load("gv6_partition.npz")
    p = ocost.partition_indices(np.arange(0, 2500), split_by_interval_int=1000, shuffle=False)
    assert np.array_equal([len(x) for x in p], d["ex1_lens"]) and np.array_equal([x[0] for x in p], d["ex1_first"])
    p = ocost.partition_indices(np.arange(0, 1800), np.array([500, 1000, 300]), 400, shuffle=False)
    assert np.array_equal([len(x) for x in p], d["ex2_lens"]) and np.array_equal([x[0] for x in p], d["ex2_first"])
    p = ocost.partition_indices(np.arange(0, 8000), np.array([3000, 5000]), 2000, shuffle=False)
    assert np.array_equal([len(x) for x in p], d["ex3_lens"]) and np.array_equal([x[0] for x in p], d["ex3_first"])
    np.random.seed(5)
    p = ocost.partition_indices(np.arange(0, 37), split_by_interval_int=10, shuffle=True)
    assert np.array_equal(np.concatenate(p), d["shuf_concat"]) and np.array_equal([len(x) for x in p], d["shuf_lens"])


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 256])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gv8_lap_known_answers(n, dtype):
    d = load("gv8_lap.npz")
    r = jv_oracle(d[f"n{n}_cost"], dtype)
    assert np.array_equal(r["rowsol"], d[f"n{n}_rowsol"])
    assert np.array_equal(r["colsol"], d[f"n{n}_colsol"])
    assert abs(r["total"] - float(d[f"n{n}_total"])) <= 1e-5 * max(1.0, abs(float(d[f"n{n}_total"])))


def test_gv8_duplicate_rows_spot_level():
    d = load("gv8_lap.npz")
    r = jv_oracle(d["dup_cost"], np.float32)
    assert np.array_equal(r["colsol"] // 5, d["dup_spot_of_col"])
    assert abs(r["total"] - float(d["dup_total"])) <= 1e-5


@pytest.mark.parametrize("n", [5, 33, 128, 500, 1200, 5000])
def test_jv_oracle_vs_scipy_random(n):
    c = np.random.default_rng(1000 + n).random((n, n)).astype(np.float32)
    r = jv_oracle(c, np.float32)
    ri, ci = linear_sum_assignment(c.astype(np.float64))
    assert np.array_equal(r["rowsol"], ci)
    # dual feasibility and complementary slackness in the solve precision
    u, v = r["u"].astype(np.float64), r["v"].astype(np.float64)
    red = c.astype(np.float64) - u[:, None] - v[None, :]
    assert red.min() > -1e-5
    assert np.abs(red[np.arange(n), r["rowsol"]]).max() < 1e-5


def test_jv_oracle_rejects_nan_and_nonsquare():
    c = np.ones((4, 4), np.float32)
    c[1, 2] = np.nan
    with pytest.raises(ValueError):
        jv_oracle(c)
    with pytest.raises(ValueError):
        jv_oracle(np.ones((3, 4), np.float32))


def test_jv_oracle_scan_counters():
    n = 300
    c = np.random.default_rng(9).random((n, n)).astype(np.float32)
    st = jv_oracle(c)["stats"]
    assert st.scans_colred == n
    assert st.row_scans == (st.scans_colred + st.scans_redtransfer + st.scans_arr + st.scans_aug_init + st.scans_aug_relax)
    assert st.scans_aug_init == st.augmentations == st.free_after_arr2


def test_gv9_other_metrics_cost():
    # oracle restatement of the Spearman / Euclidean branches of calculate_cost vs arrays captured from the reference
    d = np.load(os.path.join(G, "gv9_metrics_cost.npz"))
    for tag, metric in (("spearman", "Spearman_correlation"), ("euclidean", "Euclidean")):
        dist, loc = ocost.calculate_cost(d["sc_norm"], d["st_norm"], d["slots"], "lapjv", metric)
        assert np.array_equal(loc, d[tag + "_location_repeat"])
        np.testing.assert_allclose(dist, d[tag + "_distance_repeat"], rtol=1e-12, atol=1e-12)
    d2 = np.load(os.path.join(G, "gv2b_spearman.npz"))
    np.testing.assert_allclose(ocost.matrix_correlation_spearman(d2["sc_norm"], d2["st_norm"]), d2["corr"], rtol=0, atol=1e-12)


def test_rank_columns_average_ties():
    v = np.array([[3.0, 0.0], [1.0, 0.0], [3.0, 5.0], [0.0, 0.0], [3.0, 5.0]])
    np.testing.assert_array_equal(ocost.rank_columns(v), np.array([[4.0, 2.0], [2.0, 2.0], [4.0, 4.5], [1.0, 2.0], [4.0, 4.5]]))


# ---- the seeded instance generators behind the large goldens (tools/instances.py) are platform independent ----

def test_instance_generators_are_pinned():
    import hashlib
    from tools import instances

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    c, loc = instances.c4_chunk_cost(300, seed=4)
    assert c.shape == (300, 300) and c.dtype == np.float32 and len(loc) == 300
    c3, loc3 = instances.c3_shaped_cost(200, 10, seed=3)
    assert np.array_equal(c3[0], c3[9]) and not np.array_equal(c3[9], c3[10])
    u = instances.uniform_cost(257)
    assert np.array_equal(u, np.random.default_rng(257).random((257, 257)).astype(np.float32))
    pinned = {"c4": sha(c), "c3": sha(c3), "u": sha(u)}
    assert pinned == {"c4": "617df22ab5811ead", "c3": "6a725449cb4d96a5", "u": "79dec35b4da37fe0"}, pinned


@pytest.mark.parametrize("tag", ["u20000", "u33000", "u50000", "c3s50000", "c4s10000"])
def test_large_goldens_are_certified(tag):
    path = os.path.join(G, f"large_{tag}.npz")
    assert os.path.exists(path), "run tests/golden/make_golden_large.py"
    d = np.load(path)
    n = int(d["n"])
    assert np.array_equal(np.sort(d["colsol"]), np.arange(n))
    assert bool(d["unique"])
    mn, tight, gap = d["dual_certificate"]
    assert mn >= -1e-6 and tight <= 1e-6 and abs(gap) <= 1e-6 * n
    assert bool(d["scipy_checked"]) or bool(d["spot_level"])   # scipy may time out only on the 10x duplicated rows


def test_small_typed_instances_oracle_vs_scipy():
    # the typed (few cell types, deep searches) family at a size scipy solves in a second
    from tools import instances
    c, loc = instances.c4_chunk_cost(600, seed=9)
    r = jv_oracle(c, np.float32)
    ri, ci = linear_sum_assignment(c.astype(np.float64))
    sp = np.empty(600, np.int64)
    sp[ci] = ri
    assert np.array_equal(loc[r["colsol"]], loc[sp])


# ---- the WIDE-mode restatement (oracle/jv_oracle_impl.h, second half): same optimum, pinned the same way ----

@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 256])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_wide_oracle_gv8_known_answers(n, dtype):
    from oracle.jv import jv_oracle_wide
    d = load("gv8_lap.npz")
    r = jv_oracle_wide(d[f"n{n}_cost"], dtype)
    assert np.array_equal(r["rowsol"], d[f"n{n}_rowsol"]) and np.array_equal(r["colsol"], d[f"n{n}_colsol"])
    assert abs(r["total"] - float(d[f"n{n}_total"])) <= 1e-5 * max(1.0, abs(float(d[f"n{n}_total"])))


@pytest.mark.parametrize("n", [5, 33, 128, 500, 1200, 3000])
@pytest.mark.parametrize("rounds", [-1, 0, 3])
def test_wide_oracle_vs_scipy_random(n, rounds):
    # rounds: the budget of Jacobi row-reduction rounds (-1: the default; 0: none -- every free row is augmented; 3: cut short).
    # Whatever the budget, the augmentation finishes the optimum.
    from oracle.jv import jv_oracle_wide
    c = np.random.default_rng(1000 + n).random((n, n)).astype(np.float32)
    r = jv_oracle_wide(c, np.float32, max_rounds=rounds)
    ri, ci = linear_sum_assignment(c.astype(np.float64))
    assert np.array_equal(r["rowsol"], ci)
    assert np.array_equal(r["colsol"], jv_oracle(c, np.float32)["colsol"])
    u, v = r["u"].astype(np.float64), r["v"].astype(np.float64)
    red = c.astype(np.float64) - u[:, None] - v[None, :]
    assert red.min() > -1e-5 and np.abs(red[np.arange(n), r["rowsol"]]).max() < 1e-5
    st = r["stats"]
    assert st.arr_rounds <= (4096 + n // 4 if rounds < 0 else rounds)
    assert st.scans_aug_init == st.augmentations == st.free_after_arr


@pytest.mark.parametrize("n", [300, 1000, 2500])
@pytest.mark.parametrize("rounds", [-1, 8, 9, 20, 100])
def test_wide_oracle_scaled_phases_any_budget(n, rounds):
    # generic costs: after eight eps = 0 rounds the active list is still long and the row reduction goes through the eps-scaled phases.
    # A budget of rounds may end the scaled phases at any point -- never the final eps = 0 phase: the assignments a scaled phase leaves
    # satisfy eps-complementary slackness only, and the augmentation would finish something that is NOT the optimum (this test failed
    # for rounds = 20 and 100 before the rule)
    from oracle.jv import jv_oracle_wide
    c = np.random.default_rng(4000 + n).random((n, n)).astype(np.float32)
    r = jv_oracle_wide(c, np.float32, max_rounds=rounds)
    ri, ci = linear_sum_assignment(c.astype(np.float64))
    assert np.array_equal(r["rowsol"], ci)
    st = r["stats"]
    assert st.arr_scaled == (0 if rounds == 8 else 1) and (st.arr_phases >= 2) == bool(st.arr_scaled)
    assert st.gap_exp > 0
    u, v = r["u"].astype(np.float64), r["v"].astype(np.float64)
    red = c.astype(np.float64) - u[:, None] - v[None, :]
    assert red.min() > -1e-5 and np.abs(red[np.arange(n), r["rowsol"]]).max() < 1e-5
    if rounds < 0:                                     # the point of the phases: few rounds, shallow searches
        cold = jv_oracle(c, np.float32)["stats"]
        assert st.arr_rounds < 600 and st.scans_aug_relax < cold.scans_arr // 10


def test_wide_oracle_duplicated_rows_and_ties_never_scale():
    from oracle.jv import jv_oracle_wide
    from tools import instances
    rng = np.random.default_rng(31)
    for c in (np.repeat(rng.random((200, 1000)), 5, axis=0).astype(np.float32), rng.integers(0, 10, (600, 600)).astype(np.float32),
              instances.c3_shaped_cost(1000, 10, 3)[0]):
        st = jv_oracle_wide(c, np.float32)["stats"]
        assert st.arr_scaled == 0 and st.arr_phases == 0


@pytest.mark.parametrize("n", [2, 3, 50, 400, 1500])
def test_warm_float64_oracle(n):
    # float64 warm-started from the float32 wide solve of the narrowed matrix (what the HIP float64 path computes by default): the
    # float64 problem's optimum -- scipy on the float64 costs --, far fewer row-reduction steps than the cold classic solve
    rng = np.random.default_rng(5000 + n)
    c = rng.random((n, n)) + rng.random((n, n)) * 2.0 ** -30          # (a term below float32's resolution)
    w = jv_oracle(c, np.float64, warm=True)
    ri, ci = linear_sum_assignment(c)
    assert np.array_equal(w["rowsol"], ci)
    cold = jv_oracle(c, np.float64)
    assert np.array_equal(w["colsol"], cold["colsol"]) and abs(w["total"] - cold["total"]) <= 1e-12 * max(1.0, abs(cold["total"]))
    red = c - w["u"][:, None] - w["v"][None, :]
    assert red.min() > -1e-12 and np.abs(red[np.arange(n), w["rowsol"]]).max() < 1e-12
    if n >= 400:
        assert w["stats"].scans_arr < 3 * n < cold["stats"].scans_arr
    assert w["stats"].scans_redtransfer == 0 and w["stats"].free_after_colred == n


def test_warm_float64_oracle_on_the_perturbed_reference_golden_and_beyond_float32_range():
    d = load("gv5_solve_lap.npz")
    dist, loc = ocost.calculate_cost(d["visium_s1_sc_norm"], d["visium_s1_st_norm"], d["visium_s1_slots"])
    c = ocost.perturb(dist, 1)                                          # cytospace.py:325-327 on duplicated spot rows
    w = jv_oracle(c, np.float64, warm=True)
    assert np.array_equal(loc[w["colsol"]], d["visium_s1_mapped"])
    big = np.random.default_rng(1).random((40, 40)) * 1e300            # narrows to inf: the cold start
    a, b = jv_oracle(big, np.float64, warm=True), jv_oracle(big, np.float64)
    assert all(np.array_equal(a[k], b[k]) for k in ("rowsol", "colsol", "u", "v"))


def test_wide_oracle_ties_duplicates_and_typed_instances():
    from oracle.jv import jv_oracle_wide
    from tools import instances
    rng = np.random.default_rng(3)
    cases = [rng.integers(0, 10, (200, 200)).astype(np.float32),                       # heavy exact ties
             np.repeat(rng.random((60, 300)), 5, axis=0).astype(np.float32),           # every spot row five times
             instances.c4_chunk_cost(600, seed=9)[0], instances.c3_shaped_cost(400, 10, 3)[0]]
    for c in cases:
        n = len(c)
        r = jv_oracle_wide(c, np.float32)
        ri, ci = linear_sum_assignment(c.astype(np.float64))
        best = c.astype(np.float64)[ri, ci].sum()
        assert np.array_equal(np.sort(r["colsol"]), np.arange(n)) and np.array_equal(r["rowsol"][r["colsol"]], np.arange(n))
        assert abs(r["total"] - best) <= 1e-5 * max(1.0, abs(best))
        u, v = r["u"].astype(np.float64), r["v"].astype(np.float64)
        red = c.astype(np.float64) - u[:, None] - v[None, :]
        assert red.min() > -1e-5 and np.abs(red[np.arange(n), r["rowsol"]]).max() < 1e-5
    d = load("gv8_lap.npz")
    r = jv_oracle_wide(d["dup_cost"], np.float32)
    assert np.array_equal(r["colsol"] // 5, d["dup_spot_of_col"])


@pytest.mark.parametrize("key", ["visium_s1", "single_s7"])
def test_wide_oracle_on_the_reference_goldens_spot_level(key):
    # the reference's own solve_linear_assignment_problem outputs (gv5): the wide restatement maps every cell to the same spot
    from oracle.jv import jv_oracle_wide
    d = load("gv5_solve_lap.npz")
    dist, loc = ocost.calculate_cost(d[key + "_sc_norm"], d[key + "_st_norm"], d[key + "_slots"])
    r = jv_oracle_wide(dist, np.float32)
    assert np.array_equal(loc[r["colsol"]], d[key + "_mapped"])


def test_wide_oracle_stop_phases_expose_the_intermediate_state():
    from oracle.jv import jv_oracle_wide
    c = np.random.default_rng(8).random((400, 400)).astype(np.float32)
    a = jv_oracle_wide(c, np.float32, stop_phase=2)
    free = np.flatnonzero(a["rowsol"] < 0)
    assert len(free) == a["stats"].free_after_arr and (a["colsol"] < 0).sum() == len(free)
    asg = np.flatnonzero(a["rowsol"] >= 0)
    h = c[asg] - a["v"][None, :]                       # every assigned row sits on a minimum of its reduced costs
    assert np.all(h[np.arange(len(asg)), a["rowsol"][asg]] <= h.min(1) + 1e-7)


@pytest.mark.parametrize("tag", ["u20000", "u50000", "u70000", "c3s50000", "c4s10000", "t20000"])
def test_wide_large_goldens_are_certified(tag):
    # the wide restatement's answers at true size: certified by their own duals on ALL n^2 entries and equal (slot level; spot
    # level where spot rows are duplicated) to the classic goldens, which scipy and the perturbation re-solve certify
    path = os.path.join(G, f"large_{tag}_wide.npz")
    assert os.path.exists(path), "run tests/golden/make_golden_large.py --wide"
    d, c = np.load(path), np.load(os.path.join(G, f"large_{tag}.npz"))
    n = int(d["n"])
    assert np.array_equal(np.sort(d["colsol"]), np.arange(n)) and int(d["same_as_classic"]) == 1
    assert str(d["cost_sha256"]) == str(c["cost_sha256"])
    mn, tight, gap = d["dual_certificate"]
    assert mn >= -1e-6 and tight <= 1e-6 and abs(gap) <= 1e-6 * n
    if not bool(d["spot_level"]):
        assert np.array_equal(d["colsol"], c["colsol"])
    assert abs(float(d["total"]) - float(c["total"])) <= 1e-5 * max(1.0, abs(float(c["total"])))
    st = dict(zip([str(k) for k in d["stats_keys"]], d["stats_vals"].tolist()))
    assert st["arr_scaled"] == (0 if tag.startswith("c3s") else 1)     # made with the eps-scaled restatement (round 4)
