"""GPU parity: HIP Jonker-Volgenant (through the C ABI) vs the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

from cytospace_amd.lap import lap_solve, lap_solve_rows, lapjv_hip
from oracle.jv import jv_oracle, jv_oracle_wide

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STAT_KEYS = ["scans_colred", "scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax",
             "augmentations", "path_hops", "free_after_colred", "free_after_arr1", "free_after_arr2"]


CHAIN = dict(mode=1)      # the chain solver: classic Gauss-Seidel order, oracle jv_oracle
WIDE = dict(mode=2)       # the wide solver: order-free restatement, oracle jv_oracle_wide (the default for one float32 problem)
WIDE_KEYS = [("scans_redtransfer", "scans_redtransfer"), ("scans_arr", "scans_arr"), ("scans_aug_init", "scans_aug_init"),
             ("scans_aug_relax", "scans_aug_relax"), ("augmentations", "augmentations"), ("path_hops", "path_hops"),
             ("free_after_colred", "free_after_colred"), ("free_after_arr2", "free_after_arr"), ("wide_rounds", "arr_rounds"),
             ("wide_retired", "arr_retired"), ("wide_scaled", "arr_scaled"), ("wide_phases", "arr_phases")]


def _check_wide(c, opts=None, rounds=0):
    """The wide solver against ITS restatement (oracle/jv_oracle_impl.h, WIDE MODE): indices, duals and the semantic
    counters bit for bit -- although the kernel settles columns speculatively, 16 at a time, in no fixed order."""
    o = jv_oracle_wide(c, np.float32, max_rounds=-1 if rounds == 0 else max(rounds, 0))
    g = lap_solve(c, np.float32, return_info=True, opts=dict(WIDE, wide_rounds=rounds, **(opts or {})))
    for k in ("rowsol", "colsol", "v", "u"):
        assert np.array_equal(g[k], o[k]), k
    assert abs(g["total"] - o["total"]) <= 1e-5 * max(1.0, abs(o["total"]))
    n = c.shape[0]
    assert np.array_equal(np.sort(g["colsol"]), np.arange(n)) and np.array_equal(g["rowsol"][g["colsol"]], np.arange(n))
    od, gd = o["stats"].as_dict(), g["info"].as_dict()
    assert gd["wide"] == 1
    for kg, ko in WIDE_KEYS:
        assert gd[kg] == od[ko], (kg, gd[kg], od[ko])
    return g, o


def _check(c, dtype, opts=None, unique=False):
    """unique: the instance has ONE optimal assignment (generic costs), so the two solvers -- different restatements, different
    duals -- must return the same indices element for element."""
    o = jv_oracle(c, dtype)
    g = lap_solve(c, dtype, return_info=True, opts=dict(CHAIN, **(opts or {})))
    assert np.array_equal(g["rowsol"], o["rowsol"])
    assert np.array_equal(g["colsol"], o["colsol"])
    assert np.array_equal(g["v"], o["v"]), "dual prices v differ"
    assert np.array_equal(g["u"], o["u"]), "duals u differ"
    assert abs(g["total"] - o["total"]) <= 1e-5 * max(1.0, abs(o["total"]))   # tolerance from BASELINE.json
    n = c.shape[0]
    assert np.array_equal(np.sort(g["colsol"]), np.arange(n))
    assert np.array_equal(g["rowsol"][g["colsol"]], np.arange(n))
    od = o["stats"].as_dict()
    gd = g["info"].as_dict()
    for k in STAT_KEYS:
        assert gd[k] == od[k], (k, gd[k], od[k])
    if dtype == np.float32 and not opts:
        # the same instance through the wide solver; both solvers reach the same optimum
        gw, ow = _check_wide(c)
        assert abs(ow["total"] - o["total"]) <= 1e-5 * max(1.0, abs(o["total"]))
        if unique:
            assert np.array_equal(gw["colsol"], o["colsol"]) and np.array_equal(gw["rowsol"], o["rowsol"])
        d = lap_solve(c, np.float32, return_info=True)              # what a caller gets by default: the wide solver
        assert d["info"].wide == 1 and all(np.array_equal(d[k], gw[k]) for k in ("rowsol", "colsol", "u", "v"))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 100, 256, 1000, 1023, 1024, 1025, 2500])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_uniform(n, dtype):
    c = np.random.default_rng(n).random((n, n)).astype(np.float32)
    _check(c, dtype, unique=True)


@pytest.mark.parametrize("n", [4100, 8200, 9000])
def test_uniform_larger_variants(n):
    # crosses the per-thread column-chunk variants (CH = 2 / 5 for float32)
    c = np.random.default_rng(n).random((n, n)).astype(np.float32)
    _check(c, np.float32, unique=True)


@pytest.mark.parametrize("n,slots", [(60, 5), (500, 5), (1000, 10), (2000, 4)])
def test_duplicate_rows(n, slots):
    # Visium-like: every spot row repeated `slots` times (linear_assignment_solvers.py:63-66)
    rng = np.random.default_rng(n + slots)
    base = -rng.random((n // slots, n)).astype(np.float32)
    c = np.repeat(base, slots, axis=0)
    _check(c, np.float32)


@pytest.mark.parametrize("n", [16, 200, 1000])
def test_integer_ties(n):
    # heavy exact ties: small integer costs
    c = np.random.default_rng(n).integers(0, 10, (n, n)).astype(np.float32)
    _check(c, np.float32)
    _check(c, np.float64)


def test_negative_and_large_values():
    rng = np.random.default_rng(5)
    c = (rng.standard_normal((300, 300)) * 1e3).astype(np.float32)
    _check(c, np.float32, unique=True)


def test_nan_rejected():
    c = np.random.default_rng(0).random((64, 64)).astype(np.float32)
    c[3, 5] = np.nan
    with pytest.raises(ValueError):
        lap_solve(c)
    c[3, 5] = np.inf
    with pytest.raises(ValueError):
        lap_solve(c)


def test_non_square_rejected():
    with pytest.raises(ValueError):
        lap_solve(np.zeros((3, 4), np.float32))


def test_lapjv_call_shape():
    # `_, y, _ = solver(cost)` (linear_assignment_solvers.py:38): y[j] = row of column j
    c = np.random.default_rng(3).random((50, 50))
    row_ind, col_ind, (total, u, v) = lapjv_hip(c)
    o = jv_oracle(c, np.float32)
    assert np.array_equal(col_ind, o["colsol"]) and np.array_equal(row_ind, o["rowsol"])
    assert len(u) == 50 and len(v) == 50 and np.isfinite(total)


def test_arr_step_budget_price_war():
    # float64 solve of duplicated spot rows whose ties were broken by CytoSPACE's 1e-16 perturbation
    # (cytospace.py:325-327): the augmenting row reduction degenerates into a one-ulp price war; the
    # oracle and the kernels cut it at the same step budget and must still agree bit for bit.
    from oracle import cost as ocost
    d = np.load(os.path.join(GOLD, "gv5_solve_lap.npz"))
    dist, loc = ocost.calculate_cost(d["visium_s1_sc_norm"], d["visium_s1_st_norm"], d["visium_s1_slots"])
    c = ocost.perturb(dist, 1)
    o = jv_oracle(c, np.float64)
    assert o["stats"].arr_budget_hit == 1
    _check(c, np.float64)
    _check(c, np.float32)
    g = lap_solve(c, np.float64)
    assert np.array_equal(loc[g["colsol"]], d["visium_s1_mapped"])


@pytest.mark.parametrize("n", [64, 256])
def test_golden_known_answers(n):
    d = np.load(os.path.join(GOLD, "gv8_lap.npz"))
    g = lap_solve(d[f"n{n}_cost"], np.float32)
    assert np.array_equal(g["colsol"], d[f"n{n}_colsol"]) and np.array_equal(g["rowsol"], d[f"n{n}_rowsol"])
    assert abs(g["total"] - float(d[f"n{n}_total"])) <= 1e-5


def test_batch_of_independent_laps():
    from cytospace_amd.lap import lap_solve_batch
    sizes = [5, 300, 1200, 64, 2100, 700, 33]
    costs = [np.random.default_rng(100 + n).random((n, n)).astype(np.float32) for n in sizes]
    # a batch runs a workgroup per problem through one solver: the wide one by default, the chain solver on request
    for opts, oracle in ((None, jv_oracle_wide), (CHAIN, jv_oracle)):
        res = lap_solve_batch(costs, max_concurrent=4, return_info=True, opts=opts)
        assert len(res) == len(sizes)
        for c, r in zip(costs, res):
            o = oracle(c, np.float32)
            assert np.array_equal(r["colsol"], o["colsol"]) and np.array_equal(r["rowsol"], o["rowsol"])
            assert np.array_equal(r["u"], o["u"]) and np.array_equal(r["v"], o["v"])
            assert r["info"].wide == (0 if opts else 1)
            if opts:
                assert r["info"].row_scans == o["stats"].row_scans
            else:
                assert r["info"].scans_arr == o["stats"].scans_arr and r["info"].scans_aug_relax == o["stats"].scans_aug_relax
    bad = [costs[0], np.full((4, 4), np.nan, np.float32)]
    with pytest.raises(ValueError):
        lap_solve_batch(bad)


def test_rccl_broadcast_single_rank():
    import ctypes
    from cytospace_amd import _lib
    L = _lib.lib()
    idb = ctypes.create_string_buffer(128)
    _lib.check(L.cyto_comm_unique_id(idb))
    comm = ctypes.c_void_p()
    _lib.check(L.cyto_comm_init(idb, 0, 1, 0, ctypes.byref(comm)))
    x = np.arange(4096, dtype=np.float32)
    buf = _lib.DeviceBuffer.from_numpy(x)
    _lib.check(L.cyto_comm_bcast_f32(comm, buf.ptr, x.size, 0, 0, None))
    assert np.array_equal(buf.to_numpy(x.shape, np.float32), x)
    _lib.check(L.cyto_comm_destroy(comm))
    buf.free()


@pytest.mark.parametrize("n,slots", [(3000, 5), (6000, 10)])
def test_duplicate_rows_skip_is_exact_and_deterministic(n, slots):
    # Visium-like: runs of identical rows trigger the exact no-op-scan elision in the augmentation;
    # results, duals and the algorithmic scan counters must still equal the oracle's, run after run
    rng = np.random.default_rng(n)
    base = -(rng.random((n // slots, n)) ** 3).astype(np.float32)
    c = np.repeat(base, slots, axis=0)
    o = jv_oracle(c, np.float32)
    for _ in range(3):
        g = lap_solve(c, np.float32, return_info=True, opts=CHAIN)
        assert np.array_equal(g["rowsol"], o["rowsol"]) and np.array_equal(g["colsol"], o["colsol"])
        assert np.array_equal(g["u"], o["u"]) and np.array_equal(g["v"], o["v"])
        assert g["info"].scans_aug_relax == o["stats"].scans_aug_relax
        assert g["info"].row_groups == n // slots and g["info"].aug_scans_skipped > 0


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_large_n_code_path_forced_at_small_n(mode):
    # cyto_lap_opts.chain_variant routes any size through the large-n kernels (prices in L2-resident global memory,
    # u16 colsol in LDS, cache-certified augmentation without LDS state; 2: also the streaming dense refresh used beyond
    # 32768 columns; 3: the same with colsol in global memory as well, what n > 65535 uses); they must be bit-identical too.  The real switch-overs are at n > 26624 and n > 32768, where
    # tests/test_large_gpu.py runs them at true size against goldens.
    opts = dict(chain_variant=mode)
    for n in (5, 64, 700, 2300):
        c = np.random.default_rng(n).random((n, n)).astype(np.float32)
        _check(c, np.float32, opts)
    rng = np.random.default_rng(8)
    base = -(rng.random((300, 1500)) ** 3).astype(np.float32)
    c = np.repeat(base, 5, axis=0)
    g = lap_solve(c, np.float32, return_info=True, opts=opts)
    _check(c, np.float32, opts)
    assert g["info"].aug_scans_skipped > 0
    c = np.random.default_rng(3).integers(0, 10, (400, 400)).astype(np.float32)
    _check(c, np.float32, opts)
    c = np.random.default_rng(12).random((900, 900))
    _check(c, np.float64, opts)          # float64: the streaming chain at a size the register chain would take


def test_cache_certified_augmentation_forced():
    # augmentation=2: the augmentation relaxes only the cached columns of a row whenever the cache floor
    # certifies that no other column can matter; results must stay bit-identical to the oracle.
    opts = dict(augmentation=2, no_handover=1)
    for n in (1, 2, 5, 64, 65, 257, 700, 2300, 5000):
        c = np.random.default_rng(n).random((n, n)).astype(np.float32)
        _check(c, np.float32, opts)
    rng = np.random.default_rng(8)
    base = -(rng.random((300, 1500)) ** 3).astype(np.float32)
    c = np.repeat(base, 5, axis=0)
    g = lap_solve(c, np.float32, return_info=True, opts=opts)
    _check(c, np.float32, opts)
    assert g["info"].aug_scans_skipped > 0 and g["info"].aug_dense_scans < g["info"].scans_aug_relax
    c = np.random.default_rng(3).integers(0, 10, (400, 400)).astype(np.float32)
    _check(c, np.float32, opts)
    for n in (130, 1500):
        c = np.random.default_rng(n + 1).random((n, n)).astype(np.float32)
        _check(c, np.float32, dict(augmentation=2, no_handover=1, chain_variant=2))


def test_dense_augmentation_forced_above_its_default_range():
    # augmentation=1: the register-resident dense search at a size where the cache-certified one is the default
    c = np.random.default_rng(6100).random((6100, 6100)).astype(np.float32)
    _check(c, np.float32, dict(augmentation=1))


def test_bad_options_are_rejected():
    c = np.random.default_rng(1).random((8, 8)).astype(np.float32)
    for bad in (dict(chain_variant=4), dict(group_state_global=2), dict(aux_state_global=-1), dict(augmentation=-1), dict(no_handover=2), dict(inject_exceptions=-5)):
        with pytest.raises(ValueError):
            lap_solve(c, np.float32, opts=bad)


def test_augmentation_handover_to_dense_kernel_on_deep_searches():
    # few cell types, many near-equal columns: searches run deeper than the 63-column caches reach, the
    # cache-certified augmentation gives up (>= 10 % full-row scans) and the dense kernel finishes; bit-identical
    rng = np.random.default_rng(5)
    n, types = 5400, 6
    prof = rng.normal(size=(types, 64)).astype(np.float32)
    rows = prof[rng.integers(0, types, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
    cols = prof[rng.integers(0, types, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
    c = -(rows @ cols.T).astype(np.float32)
    g = lap_solve(c, np.float32, return_info=True, opts=CHAIN)
    o = jv_oracle(c, np.float32)
    for k in ("rowsol", "colsol", "u", "v"):
        assert np.array_equal(g[k], o[k]), k
    assert g["info"].scans_aug_relax == o["stats"].scans_aug_relax and g["info"].augmentations == o["stats"].augmentations
    assert g["info"].aug_handover >= 0


@pytest.mark.parametrize("k", [5, 100])
def test_exception_columns_of_the_cache_certified_augmentation(k):
    # a price that a rounding pushed UP takes its column out of the cache certificates: such columns are relaxed
    # explicitly in every cached step (k = 5), and when the list overflows the certificates are abandoned (k = 100:
    # every scan reads its row).  cyto_lap_opts.inject_exceptions pretends the first k columns are such columns; results stay exact.
    opts = dict(augmentation=2, no_handover=1, inject_exceptions=k)
    for n in (300, 2300):
        c = np.random.default_rng(n + 7).random((n, n)).astype(np.float32)
        _check(c, np.float32, opts)
    base = -(np.random.default_rng(9).random((200, 1000)) ** 3).astype(np.float32)
    _check(np.repeat(base, 5, axis=0), np.float32, opts)
    _check(np.repeat(base, 5, axis=0), np.float32, dict(augmentation=2, inject_exceptions=k))     # with the hand-over allowed


def test_full_size_properties_bench_config():
    # BASELINE.json configs[1] (20000 x 20000 dense): size-independent properties of an optimal assignment -- a
    # permutation, rowsol/colsol mutually inverse, dual feasibility u_i + v_j <= c_ij with equality on the assignment
    # (complementary slackness certifies optimality), total == sum of the assigned costs.  (bench.py additionally compares
    # this instance with the CPU oracle bit for bit; that takes the oracle 14 s.)
    n = 20000
    c = np.random.default_rng(n).random((n, n)).astype(np.float32)
    g = lap_solve(c, np.float32, return_info=True)
    rowsol, colsol = g["rowsol"], g["colsol"]
    assert np.array_equal(np.sort(colsol), np.arange(n)) and np.array_equal(rowsol[colsol], np.arange(n))
    tot = float(c[np.arange(n), rowsol].astype(np.float64).sum())
    assert abs(tot - g["total"]) <= 1e-5 * max(1.0, abs(tot))
    rows = np.random.default_rng(1).choice(n, 1024, replace=False)
    red = c[rows].astype(np.float64) - g["u"][rows].astype(np.float64)[:, None] - g["v"].astype(np.float64)[None, :]
    assert red.min() > -1e-5 and np.abs(red[np.arange(len(rows)), rowsol[rows]]).max() < 1e-5
    assert g["info"].wide == 1 and g["info"].wide_dense_aug < 0.05 * g["info"].scans_aug_relax      # the augmentation ran from the row caches


def test_float64_host_matrix_is_narrowed_on_the_device():
    # the reference calls lapjv(cost_scaled) with a float64 array that is solved in float32: same answer as casting first
    from cytospace_amd.lap import lapjv_hip
    for n in (1, 7, 130, 2500):
        c64 = np.random.default_rng(n).random((n, n)) * 3.0 - 1.0
        row, col, (tot, u, v) = lapjv_hip(c64)
        ref = lap_solve(c64.astype(np.float32), np.float32)
        assert np.array_equal(row, ref["rowsol"]) and np.array_equal(col, ref["colsol"])
        assert np.array_equal(u, ref["u"]) and np.array_equal(v, ref["v"]) and tot == ref["total"]
    with pytest.raises(ValueError):
        lapjv_hip(np.array([[1.0, np.nan], [0.0, 1.0]]))
    with pytest.raises(ValueError):
        lapjv_hip(np.array([[1e300, 0.0], [0.0, 1.0]]))        # overflows to inf in float32, like astype(float32)


# ---- SURVEY 8f rank 3 (first half): row indirection -- the cost holds every distinct spot row once ----

@pytest.mark.parametrize("opts", [CHAIN, dict(chain_variant=1), dict(chain_variant=2), dict(chain_variant=3), dict(augmentation=1), dict(augmentation=2, no_handover=1)])
def test_row_map_equals_the_materialised_matrix(opts):
    # cyto_lap_f32_rowmap(rows, np.repeat(arange(S), slots)) == cyto_lap_f32(rows[rowmap]) == the oracle, bit for bit,
    # with equal slots, ragged slots, stored rows nobody uses (slots == 0) and a single spot
    rng = np.random.default_rng(31)
    cases = []
    for S, slots in ((12, np.full(12, 5)), (300, rng.integers(0, 6, 300)), (1, np.array([7])), (500, np.full(500, 4)),
                     (900, np.where(rng.random(900) < 0.3, 0, rng.integers(1, 4, 900)))):
        n = int(slots.sum())
        rows = -(rng.random((S, n)) ** 3).astype(np.float32)
        cases.append((rows, np.repeat(np.arange(S), slots).astype(np.int32)))
    for rows, rowmap in cases:
        full = rows[rowmap]
        a = lap_solve_rows(rows, rowmap, return_info=True, opts=opts)
        b = lap_solve(full, np.float32, return_info=True, opts=opts)
        o = jv_oracle(full, np.float32)
        for k in ("rowsol", "colsol", "u", "v"):
            assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], o[k]), k
        assert a["total"] == b["total"]
        for k in STAT_KEYS:
            assert a["info"].as_dict()[k] == o["stats"].as_dict()[k], k


def test_row_map_argument_validation():
    rows = np.random.default_rng(1).random((3, 6)).astype(np.float32)
    for bad in ([0, 0, 1, 1, 2, 3], [0, 1, 0, 1, 2, 2], [-1, 0, 0, 1, 1, 2]):        # out of range, not monotone, negative
        with pytest.raises(ValueError):
            lap_solve_rows(rows, np.array(bad, np.int32))
    with pytest.raises(ValueError):
        lap_solve_rows(rows, np.array([0, 0, 1, 1, 2], np.int32))                    # n != number of columns


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_float64_streaming_chain_with_and_without_row_caches(variant):
    # float64 beyond n = 4096 runs jv_chain_stream; chain_variant 1 = with the 63-column row caches for reduction transfer and
    # augmenting row reduction (the default there), 2 = every scan reads its row, 3 = the caches with colsol in global memory
    # (what n > 65 535 uses).  Forced here at sizes the oracle solves quickly.
    opts = dict(chain_variant=variant)
    for n in (2, 5, 63, 64, 65, 300, 1500):
        c = np.random.default_rng(n).random((n, n))
        _check(c, np.float64, opts)
    c = np.random.default_rng(5).integers(0, 10, (400, 400)).astype(np.float64)      # heavy ties: the floor ties with cached values
    _check(c, np.float64, opts)
    rng = np.random.default_rng(8)
    c = np.repeat(-(rng.random((100, 500)) ** 3), 5, axis=0)                           # duplicated rows
    _check(c, np.float64, opts)
    c = c + 1e-16 * np.random.RandomState(1).rand(*c.shape)                            # CytoSPACE's perturbation (cytospace.py:325-327)
    _check(c, np.float64, opts)


def test_float64_default_path_above_4096():
    n = 4500
    c = np.random.default_rng(n).random((n, n))
    g = lap_solve(c, np.float64, return_info=True, opts=CHAIN)          # (the cold classic chain; the default is warm-started)
    _check(c, np.float64)
    assert g["info"].hbm_row_reads < g["info"].scans_redtransfer + g["info"].scans_arr      # most scans were served by the caches


@pytest.mark.parametrize("aug", [1, 2])
def test_duplicate_row_group_state_in_global_memory(aug):
    # large problems with thousands of row groups keep the groups' best offsets / search stamps in global memory (L2) instead
    # of LDS; cyto_lap_opts.group_state_global forces that form at a size the oracle checks -- dense (1) and cache-certified (2)
    rng = np.random.default_rng(21)
    base = -(rng.random((360, 1800)) ** 3).astype(np.float32)
    c = np.repeat(base, 5, axis=0)
    opts = dict(augmentation=aug, no_handover=1, group_state_global=1)
    g = lap_solve(c, np.float32, return_info=True, opts=opts)
    _check(c, np.float32, opts)
    assert g["info"].aug_scans_skipped > 0 and g["info"].row_groups == 360
    _check(c, np.float32, dict(chain_variant=2, group_state_global=1))
    # ... and the dense kernel's per-column auxiliaries in global memory as well (what it uses beyond ~13 000 columns)
    _check(c, np.float32, dict(augmentation=1, aux_state_global=1, group_state_global=aug - 1))
    _check(np.random.default_rng(4).random((700, 700)).astype(np.float32), np.float32, dict(augmentation=1, aux_state_global=1))


def test_chain_solver_tie_between_two_groups_of_one_lane():
    # Regression (tools/stress_lap.py seed 2407: n = 4 750, rows in runs of identical copies; found by tools/trace_aug_scans.py).  In
    # search 92 an assigned and an UNASSIGNED column sit at the same (final) distance in two different groups of four of the SAME lane
    # of the dense augmentation kernel: the pick's one-holder shortcut looked at the first such group only and scanned the assigned
    # column before ending at the unassigned one -- 126 544 scans against the oracle's 126 543, everything else identical (a scan AT
    # the final distance changes no price).  The shortcut now needs ONE group at the value; counters equal the oracle's.
    from tools.stress_lap import make
    rng = np.random.default_rng(1000 + 2407)
    n = int(rng.integers(3000, 6000))
    c = make("dup", n, rng)
    _check(c, np.float32)                                    # every counter equal; the wide solver too
    g = lap_solve(c, np.float32, return_info=True, opts=CHAIN)
    assert g["info"].scans_aug_relax == 126543


@pytest.mark.parametrize("par", [2, 5, 16])
def test_wide_several_searches_at_once(par):
    # cyto_lap_opts.wide_par: the searches of `par` consecutive free rows run at once, a workgroup each, from ONE state; the longest
    # prefix (in row order) whose settled sets and sinks are pairwise disjoint is committed, the rest runs again in the next batch --
    # exactly what one search after the other computes: the oracle's answer and its semantic counters bit for bit, whatever `par`
    rng = np.random.default_rng(90 + par)
    cases = [rng.random((n, n)).astype(np.float32) for n in (3, 64, 700, 2300, 4300)]
    cases.append(np.repeat(rng.random((150, 600)), 4, axis=0).astype(np.float32))                     # duplicated rows
    cases.append(rng.integers(0, 10, (400, 400)).astype(np.float32))                                    # heavy ties: most searches collide
    prof = rng.normal(size=(6, 64)).astype(np.float32)                                                  # few cell types: full-row relaxations
    rows = prof[rng.integers(0, 6, 1200)] + 0.05 * rng.normal(size=(1200, 64)).astype(np.float32)
    cols = prof[rng.integers(0, 6, 1200)] + 0.05 * rng.normal(size=(1200, 64)).astype(np.float32)
    cases.append(-(rows @ cols.T).astype(np.float32))
    seen = 0
    for c in cases:
        for rounds in (0, 2, -1):                                # (a short or no row reduction leaves many searches)
            g, o = _check_wide(c, opts=dict(wide_par=par), rounds=rounds)
            seen += g["info"].wide_par_batches
            assert g["info"].wide_par_batches <= max(1, g["info"].augmentations)
    assert seen > 0


def test_wide_single_solves_from_several_threads():
    # a single problem's searches run 16 at a time on 16 workgroups that wait for each other at grid barriers: only ONE such kernel may
    # be in flight per device (three of them could each get a part of their workgroups scheduled); a solve that finds the slot taken
    # runs its searches one at a time -- the same answer either way
    import threading
    n = 2600
    cs = [np.random.default_rng(300 + k).random((n, n)).astype(np.float32) for k in range(3)]
    want = [jv_oracle_wide(c, np.float32) for c in cs]
    bad, batches = [], []

    def work(k):
        for _ in range(4):
            g = lap_solve(cs[k], np.float32, return_info=True)
            batches.append(g["info"].wide_par_batches)
            if not all(np.array_equal(g[key], want[k][key]) for key in ("rowsol", "colsol", "u", "v")):
                bad.append(k)

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad and len(batches) == 12 and max(batches) > 0


# ---- float64 by default: warm-started from the float32 wide solve of the narrowed matrix (oracle: jv_oracle_warm_f64) ----

def _check_warm(c64):
    o = jv_oracle(c64, np.float64, warm=True)
    g = lap_solve(c64, np.float64, return_info=True)
    assert g["info"].f64_warm == 1
    for k in ("rowsol", "colsol", "v", "u"):
        assert np.array_equal(g[k], o[k]), k
    assert abs(g["total"] - o["total"]) <= 1e-9 * max(1.0, abs(o["total"]))
    od, gd = o["stats"].as_dict(), g["info"].as_dict()
    for k in STAT_KEYS:
        assert gd[k] == od[k], (k, gd[k], od[k])
    return g, o


@pytest.mark.parametrize("n", [2, 3, 64, 300, 1000, 2500, 4500, 6000])
def test_float64_warm_start_uniform(n):
    # the precision of lapjv(cost, force_doubles=True) / lap.lapjv (linear_assignment_solvers.py:13-15, 36): the float64 chain starts
    # from the prices of the float32 wide solve of the narrowed matrix, every row free -- ~1.2 n row-reduction steps instead of ~100 n;
    # the same optimum as the cold classic solve (unique: the same indices), its own duals
    c = np.random.default_rng(7000 + n).random((n, n))
    g, o = _check_warm(c)
    cold = jv_oracle(c, np.float64)
    assert np.array_equal(g["colsol"], cold["colsol"])
    if n >= 300:
        assert g["info"].scans_arr < 4 * n < cold["stats"].scans_arr


def test_float64_warm_start_ties_duplicates_and_the_perturbation():
    rng = np.random.default_rng(71)
    _check_warm(rng.integers(0, 10, (400, 400)).astype(np.float64))                                   # heavy ties
    base = -(rng.random((160, 800)) ** 3)
    dup = np.repeat(base, 5, axis=0)
    g, o = _check_warm(dup)                                                                            # duplicated spot rows
    np.random.seed(1)
    pert = dup + 1e-16 * np.random.rand(*dup.shape)                                                    # cytospace.py:325-327
    g2, o2 = _check_warm(pert)
    loc = np.repeat(np.arange(160), 5)
    assert np.array_equal(loc[g["colsol"]], loc[jv_oracle(dup, np.float64)["colsol"]])                # spot level: any exact solver's
    c = (rng.standard_normal((300, 300)) * 1e3)
    _check_warm(c)
    big = rng.random((50, 50)) * 1e300                                                                 # beyond float32's range: cold start
    gb = lap_solve(big, np.float64, return_info=True)
    ob = jv_oracle(big, np.float64, warm=True)
    assert gb["info"].f64_warm == 0 and all(np.array_equal(gb[k], ob[k]) for k in ("rowsol", "colsol", "u", "v"))


# ---- the wide solver (cyto_lap_opts.mode = 2; the default for one float32 problem): lap_wide.hip vs oracle WIDE MODE ----

@pytest.mark.parametrize("rounds", [-1, 1, 3, 40, 600])
def test_wide_round_budget(rounds):
    # the budget of Jacobi row-reduction rounds cuts the rounds short at the same point in the kernel and in the oracle; the rows
    # still active go to the augmentation (wide_rounds = -1: no round at all -- every free row is augmented).  A budget that ends
    # inside a SCALED phase still runs the final eps = 0 phase: only its assignments satisfy what the searches need
    c = np.random.default_rng(900 + rounds).random((900, 900)).astype(np.float32)
    g, o = _check_wide(c, rounds=rounds)
    assert g["info"].wide_rounds <= max(rounds, 0) or g["info"].wide_scaled == 1
    assert (g["info"].wide_scaled == 1) == (rounds > 8)
    assert np.array_equal(g["colsol"], jv_oracle(c, np.float32)["colsol"])          # a unique optimum: the chain's indices too


def test_wide_row_map_and_duplicated_rows():
    rng = np.random.default_rng(12)
    for S, slots in ((12, np.full(12, 5)), (300, rng.integers(0, 6, 300)), (1, np.array([7])), (500, np.full(500, 4))):
        n = int(slots.sum())
        rows = -(rng.random((S, n)) ** 3).astype(np.float32)
        rowmap = np.repeat(np.arange(S), slots).astype(np.int32)
        full = rows[rowmap]
        a = lap_solve_rows(rows, rowmap, return_info=True, opts=WIDE)
        g, o = _check_wide(full)
        for k in ("rowsol", "colsol", "u", "v"):
            assert np.array_equal(a[k], g[k]), k
        # the chain solver reaches the same optimum; which slot of a spot a cell takes is arbitrary, the spot is not
        oc = jv_oracle(full, np.float32)
        assert abs(oc["total"] - o["total"]) <= 1e-5 * max(1.0, abs(o["total"]))


def test_wide_one_edge_searches_on_the_whole_chip():
    # repeated spot rows: the searches that end at the free row's own best column are disposed of by deferred acceptance on the whole
    # chip before the search kernel (wide_claim_*) -- the same rows take the same columns as in the serial loop they replace, and the
    # loop goes on behind the first row that needs a search proper.  (a) every search is one edge (the c3 shape); (b) real searches
    # among them; (c) a batch that mixes repeated-row problems with plain ones; (d) runs longer than a wave looks ahead
    from cytospace_amd.lap import lap_solve_batch
    from tools import instances
    c, _ = instances.c3_shaped_cost(3000, 10, 3)
    g, o = _check_wide(c)
    assert g["info"].wide_trivial == 2100 and g["info"].wide_aug_rounds == 0
    rng = np.random.default_rng(91)
    dup = np.repeat(rng.random((250, 1000)), 4, axis=0).astype(np.float32)
    g, o = _check_wide(dup)
    assert g["info"].wide_trivial > 0 and g["info"].wide_aug_rounds > 0
    long_runs = np.repeat(-(rng.random((12, 1200)) ** 3), 100, axis=0).astype(np.float32)
    _check_wide(long_runs)
    cs = [dup, rng.random((1000, 1000)).astype(np.float32), c[:1000, :1000].copy(), np.repeat(rng.random((100, 1000)), 10, axis=0).astype(np.float32)]
    oracle = [jv_oracle_wide(x, np.float32) for x in cs]
    res = lap_solve_batch(cs, return_info=True, opts=dict(mode=2))
    for x, g, o in zip(cs, res, oracle):
        for key in ("rowsol", "colsol", "u", "v"):
            assert np.array_equal(g[key], o[key]), key
        assert g["info"].path_hops == o["stats"].path_hops and g["info"].scans_aug_relax == o["stats"].scans_aug_relax


def test_wide_full_row_fallbacks_on_near_equal_columns():
    # few cell types: hundreds of near-equal columns per row, the 63-column caches cannot certify -- bids and relaxations read
    # the full cost row, the search re-converges after every certificate pass; still the oracle's answer bit for bit
    rng = np.random.default_rng(5)
    n, types = 1500, 6
    prof = rng.normal(size=(types, 64)).astype(np.float32)
    rows = prof[rng.integers(0, types, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
    cols = prof[rng.integers(0, types, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
    c = -(rows @ cols.T).astype(np.float32)
    g, o = _check_wide(c)
    assert g["info"].wide_dense_aug > 0 and g["info"].wide_dense_arr > 0
    assert abs(jv_oracle(c, np.float32)["total"] - o["total"]) <= 1e-5 * max(1.0, abs(o["total"]))


@pytest.mark.parametrize("n", [4100, 9000])
def test_wide_uniform_larger(n):
    c = np.random.default_rng(n).random((n, n)).astype(np.float32)
    g, o = _check_wide(c)
    assert g["info"].wide_aug_settled >= g["info"].scans_aug_relax        # speculative: a column may be settled more than once


@pytest.mark.parametrize("groups", [2, 5, 16])
def test_wide_search_on_several_workgroups(groups):
    # cyto_lap_opts.wide_groups: the augmentation's searches run on several workgroups at once, asynchronously (no barrier inside a
    # search's steady state; labels, dirty bits and block minima exchanged through agent-scope atomics in L2).  The labels are the
    # fixed point of a monotone system, so who settles what when cannot matter: the oracle's answer bit for bit, every time.
    rng = np.random.default_rng(40 + groups)
    cases = [rng.random((n, n)).astype(np.float32) for n in (3, 64, 700, 2300)]
    cases.append(np.repeat(rng.random((150, 600)), 4, axis=0).astype(np.float32))                     # duplicated rows: tight cycles
    cases.append(rng.integers(0, 10, (400, 400)).astype(np.float32))                                    # heavy ties
    prof = rng.normal(size=(6, 64)).astype(np.float32)                                                  # few cell types: full-row relaxations
    rows = prof[rng.integers(0, 6, 1200)] + 0.05 * rng.normal(size=(1200, 64)).astype(np.float32)
    cols = prof[rng.integers(0, 6, 1200)] + 0.05 * rng.normal(size=(1200, 64)).astype(np.float32)
    cases.append(-(rows @ cols.T).astype(np.float32))
    for c in cases:
        for rounds in (0, 2):                                    # (a short row reduction leaves many searches)
            g, o = _check_wide(c, opts=dict(wide_groups=groups), rounds=rounds)
            assert g["info"].wide == 1


@pytest.mark.parametrize("waves", ["0", "1", "8", "32", "8-guess"])
def test_row_cache_builders_agree(waves):
    # The full-chip cache build: build_row_caches_wave (a wave per row, guessed floor, one sweep; cyto_lap_opts.cache_waves waves per CU --
    # with 1 a wave takes many rows and its guesses matter, with 32 most rows are a wave's first and take the lane-minima floor) or, with
    # "0" (cache_waves = -1), the workgroup-per-row builders of rounds 1-3.  Which columns a cache holds is a matter of speed only: both
    # solvers give the oracle's answers bit for bit with every builder -- on uniform, few-cell-type, tie-heavy (the floor search cannot
    # separate: caches without entries) and duplicated-row instances, ragged sizes (n % 4 != 0, n < 64) included.
    # ("8-guess": cache_stream = -1, a neighbouring row's floor as the guess and the lane minima's 35th as the fallback, instead of the
    #  guess-free streaming selection of rows of >= 2 048 columns)
    stream = -1 if waves.endswith("guess") else 1
    waves = int(waves.split("-")[0])
    bopts = dict(cache_waves=waves if waves > 0 else -1, cache_unroll=4 if waves == 1 else 8, cache_stream=stream)
    rng = np.random.default_rng(91)
    prof = rng.normal(size=(5, 48)).astype(np.float32)
    typed = -((prof[rng.integers(0, 5, 3001)] + 0.05 * rng.normal(size=(3001, 48)).astype(np.float32)) @
              (prof[rng.integers(0, 5, 3001)] + 0.05 * rng.normal(size=(3001, 48)).astype(np.float32)).T).astype(np.float32)
    cases = [rng.random((6002, 6002)).astype(np.float32), typed, rng.integers(0, 6, (1500, 1500)).astype(np.float32),
             np.repeat(rng.random((250, 1250)), 5, axis=0).astype(np.float32), rng.random((37, 37)).astype(np.float32),
             rng.random((5, 5)).astype(np.float32)]
    for c in cases:
        _check_wide(c, opts=bopts)
        if c.shape[0] >= 5200 or c.shape[0] == 3001:             # (the chain solver builds caches from n = 5 121 on, and where forced)
            _check(c, np.float32, opts=dict(bopts, augmentation=2) if c.shape[0] == 3001 else bopts)


@pytest.mark.parametrize("rebuild", [0, 1, 3, -1])
def test_wide_row_caches_rebuilt_between_searches(rebuild):
    # cyto_lap_opts.wide_rebuild: the search kernel returns to the driver when its row caches have gone stale (every search lowers
    # prices; a floor is a bound as of the build), the whole chip rebuilds them against the prices reached and the searches go on
    # where they stopped (few cell types: without it most settlements fall back to full cost rows).  The labels of a search do not
    # depend on which rows were read in full: the oracle's answer bit for bit, whatever the schedule.
    rng = np.random.default_rng(77)
    prof = rng.normal(size=(6, 64)).astype(np.float32)
    cases = []
    for n in (1200, 4200):
        rows = prof[rng.integers(0, 6, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
        cols = prof[rng.integers(0, 6, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
        cases.append(-(rows @ cols.T).astype(np.float32))
    cases.append(rng.random((2500, 2500)).astype(np.float32))
    cases.append(np.repeat(rng.random((300, 1200)), 4, axis=0).astype(np.float32))
    launches = []
    for c in cases:
        for rounds in (0, 2):
            g, o = _check_wide(c, opts=dict(wide_rebuild=rebuild), rounds=rounds)
            launches.append(g["info"].wide_aug_launches)
    assert all(k >= 1 for k in launches)
    if rebuild == -1:
        assert all(k == 1 for k in launches)
    if rebuild == 1:
        assert max(launches) > 10                              # one launch per search (one-edge searches aside)


def test_wide_batch_rebuilds_only_what_is_unfinished():
    # a batch in one launch per phase: problems that need fresh caches and problems that are done long before share the launches
    from cytospace_amd.lap import lap_solve_batch
    # (n >= 4096 and more than four problems: the first eight rounds on the whole chip, the other long-list rounds in wide_arr)
    rng = np.random.default_rng(78)
    prof = rng.normal(size=(5, 48)).astype(np.float32)
    n = 4200
    cs = []
    for k in range(6):
        if k % 2:
            cs.append(rng.random((n, n)).astype(np.float32))
        else:
            rows = prof[rng.integers(0, 5, n)] + 0.05 * rng.normal(size=(n, 48)).astype(np.float32)
            cols = prof[rng.integers(0, 5, n)] + 0.05 * rng.normal(size=(n, 48)).astype(np.float32)
            cs.append(-(rows @ cols.T).astype(np.float32))
    oracle = [jv_oracle_wide(c, np.float32) for c in cs]
    for rebuild in (0, 2):
        res = lap_solve_batch(cs, return_info=True, opts=dict(mode=2, wide_rebuild=rebuild))
        for c, g, o in zip(cs, res, oracle):
            for key in ("rowsol", "colsol", "u", "v"):
                assert np.array_equal(g[key], o[key]), key
            assert g["info"].scans_aug_relax == o["stats"].scans_aug_relax and g["info"].path_hops == o["stats"].path_hops


@pytest.mark.parametrize("n", [300, 1000, 2500, 4300])
def test_wide_scaled_row_reduction(n):
    # generic costs: after eight eps = 0 rounds the active list is still long, the instance goes through the eps-scaled phases
    # (every row unassigned at each phase's start, prices kept, a phase's sequential tail cut) and a final eps = 0 phase; a
    # duplicated-row instance never scales (its rows retire on ties at once).  The phase machine runs on the whole chip, two
    # launches per round: the oracle's state bit for bit
    rng = np.random.default_rng(n)
    g, o = _check_wide(rng.random((n, n)).astype(np.float32))
    assert g["info"].wide_scaled == 1 and g["info"].wide_phases >= 2
    prof = rng.normal(size=(6, 64)).astype(np.float32)                          # few cell types: many full-row bids in the coarse phases
    rows = prof[rng.integers(0, 6, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
    cols = prof[rng.integers(0, 6, n)] + 0.05 * rng.normal(size=(n, 64)).astype(np.float32)
    g, o = _check_wide(-(rows @ cols.T).astype(np.float32))
    assert g["info"].wide_scaled == 1
    g, o = _check_wide(np.repeat(rng.random((n // 5, n)), 5, axis=0).astype(np.float32))
    assert g["info"].wide_scaled == 0


@pytest.mark.parametrize("wipe", [1, 3, 50])
def test_wide_bid_words_wiped_every_few_rounds(wipe):
    # the per-column bid words carry a 12-bit round tag RELATIVE to their last wipe (a later round's bid beats whatever earlier
    # rounds left, so the words are never reset between rounds); cyto_lap_opts.wide_wipe moves the wipes of a word buffer from every
    # 1 024 of its launches to every few: the tags start over again and again in the middle of phases -- same rounds, same answer
    rng = np.random.default_rng(60 + wipe)
    for n in (700, 2600):
        _check_wide(rng.random((n, n)).astype(np.float32), opts=dict(wide_wipe=wipe))
    _check_wide(np.repeat(rng.random((200, 1000)), 5, axis=0).astype(np.float32), opts=dict(wide_wipe=wipe))


@pytest.mark.parametrize("groups", [0, 4])
def test_wide_more_tight_hops_than_a_label_counts(groups):
    # a band matrix: row i is cheap on columns i and i + 1 only.  The column reduction gives column j to row j - 1 and leaves row
    # n - 1 free and column 0 unassigned; with no row-reduction rounds the one search runs n - 1 TIGHT edges in a row (every
    # reduced cost on the path is exactly 0 past the root's) -- more than the 4095 a label's hop field counts: there the distance is
    # stepped to the next representable value (edge_lv in lap_wide.hip, JV_WIDE_KMAX in the oracle) and the count starts over
    n = 6000
    rng = np.random.default_rng(6)
    c = (1.0 + rng.random((n, n))).astype(np.float32)
    i = np.arange(n)
    c[i, i] = 0.5
    c[i[:-1], i[:-1] + 1] = 0.5
    c[n - 1, n - 1] = 0.5 + 2.0 ** -10                     # the root's edge: a distance > 0 that the tight edges keep
    g, o = _check_wide(c, opts=dict(wide_groups=groups), rounds=-1)
    assert o["stats"].path_hops == n and g["info"].path_hops == n           # one path through every row
    assert o["stats"].scans_aug_relax >= 4096
    u, v = g["u"].astype(np.float64), g["v"].astype(np.float64)
    red = c.astype(np.float64) - u[:, None] - v[None, :]
    assert red.min() >= -1e-6 and np.abs(red[i, g["rowsol"]]).max() <= 1e-6      # the stepped distances still give feasible duals


def test_one_process_two_devices():
    # kernels that need more than 64 KB of dynamic LDS (prices and owners in LDS) get their per-device attribute on EVERY device a
    # process uses: device 0 first, then device 1 (needs two GPUs)
    from cytospace_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two devices in one process")
    n = 12000
    c = np.random.default_rng(n).random((n, n)).astype(np.float32)
    for opts in (None, CHAIN):
        a = lap_solve(c, np.float32, device_id=0, opts=opts)
        b = lap_solve(c, np.float32, device_id=1, opts=opts)
        for k in ("rowsol", "colsol", "u", "v"):
            assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(lap_solve(c, np.float32, device_id=1)["colsol"], jv_oracle(c, np.float32)["colsol"])


@pytest.mark.parametrize("rounds", [1, 3, 8, 9, 60])
def test_wide_first_rounds_on_the_whole_chip(rounds):
    # the rounds with a long active list are two full-chip launches each (the phase machine, wide_sc_*), a short list's rounds run in
    # the one-workgroup kernel: the same rounds, so any budget -- inside the first eight, at their end, beyond -- gives the oracle's state
    n = 4300
    c = np.random.default_rng(n + rounds).random((n, n)).astype(np.float32)
    _check_wide(c, rounds=rounds)
    base = -(np.random.default_rng(rounds).random((n // 10, n)) ** 3).astype(np.float32)
    _check_wide(np.repeat(base, 10, axis=0), rounds=rounds)           # duplicated rows: most bids lose their round


@pytest.mark.parametrize("groups", [0, 3])
def test_wide_plateaus_every_distance_equal(groups):
    # small integer costs: after the reductions every reduced cost in play is 0, so every label of a search has the SAME distance
    # and only the tight-hop counts order them (a breadth-first search in the tight subgraph).  Found by tools/stress_lap.py: the
    # certificate pass must cover every settled column -- also those at the end's distance with fewer hops --, an uncached tight
    # edge out of one of them changes which unassigned column is reached first.
    for seed, n in ((1110, 2270), (7, 900), (8, 1500)):
        rng = np.random.default_rng(seed)
        k = int(rng.integers(3, 50)) if seed == 1110 else 4
        if seed == 1110:
            n = int(np.random.default_rng(1110).integers(300, 3000))
            rng = np.random.default_rng(1110); rng.integers(300, 3000)
            c = rng.integers(0, int(rng.integers(3, 50)), (n, n)).astype(np.float32)
        else:
            c = rng.integers(0, k, (n, n)).astype(np.float32)
        _check_wide(c, opts=dict(wide_groups=groups))


def test_float64_certificate_of_a_float32_solve():
    # cyto_lap_opts.certify: one more pass over the matrix proves in float64 how far from optimal the float32 result can be --
    # gap = sum_i (u_i - min_j (c_ij - v_j)) >= total - optimum, every difference exact in float64.  Against numpy on the same duals;
    # and the bound holds: an independent exact solver's total lies within it.
    from scipy.optimize import linear_sum_assignment
    from tools import instances
    for c in (np.random.default_rng(41).random((1500, 1500)).astype(np.float32), instances.typed_unique_cost(1203, 1203, 7)[0],
              np.repeat(np.random.default_rng(42).random((301, 1204)).astype(np.float32), 4, axis=0)):
        n = len(c)
        for opts in (dict(certify=1), dict(certify=1, mode=1)):
            g = lap_solve(c, np.float32, return_info=True, opts=opts)
            i = g["info"]
            red = c.astype(np.float64) - g["v"].astype(np.float64)[None, :]
            viol = np.maximum(red[np.arange(n), g["rowsol"]] - red.min(1), 0.0)
            assert i.certified == 1 and i.gap_rows == int((viol > 0).sum()) and i.gap_max_f64 == viol.max()
            assert abs(i.gap_f64 - viol.sum()) <= 1e-12 * max(1.0, viol.sum())
            r, cc = linear_sum_assignment(c.astype(np.float64))
            opt = float(c.astype(np.float64)[r, cc].sum())
            mine = float(c.astype(np.float64)[np.arange(n), g["rowsol"]].sum())
            assert -1e-9 <= mine - opt <= i.gap_f64 + 1e-9 and i.gap_f64 <= 1e-5 * max(1.0, abs(opt))
    g = lap_solve(np.random.default_rng(43).random((300, 300)).astype(np.float32), np.float32, return_info=True)
    assert g["info"].certified == 0                                   # (off by default: one pass less)
    # a row map (duplicated spot rows stored once) certifies like the materialised matrix
    rows = np.random.default_rng(44).random((250, 1000)).astype(np.float32)
    rowmap = np.repeat(np.arange(250), 4).astype(np.int32)
    a = lap_solve_rows(rows, rowmap, return_info=True, opts=dict(certify=1))["info"]
    b = lap_solve(rows[rowmap], np.float32, return_info=True, opts=dict(certify=1))["info"]
    assert a.certified == b.certified == 1 and a.gap_f64 == b.gap_f64 and a.gap_rows == b.gap_rows


def test_default_solver_indices_on_certified_unique_instances():
    # the wide solver against the CLASSIC oracle's == scipy's indices on instances whose optimum is certified
    # unique (tests/golden/cross_unique.npz, made on the CPU by tools/cross_unique.py --make: uniform and few-cell-type, n up to 8 200) --
    # the first 60 here, all of them in tools/cross_unique.py
    import hashlib
    from tools import cross_unique
    d = np.load(cross_unique.OUT)
    assert len(d["n"]) >= 300
    for k in range(60):
        c = cross_unique.instance(str(d["kind"][k]), int(d["n"][k]), int(d["seed"][k]), int(d["K"][k]))
        g = lap_solve(c, np.float32, return_info=True, opts=dict(certify=1))
        assert hashlib.sha256(np.ascontiguousarray(g["colsol"], dtype=np.int32).tobytes()).hexdigest() == str(d["colsol_sha256"][k]), k
        assert g["info"].certified == 1 and g["info"].gap_f64 <= 1e-5 * max(1.0, abs(g["total"]))


def test_float64_polish_of_a_float32_solve():
    # cyto_lap_opts.polish: where the float64 certificate leaves a gap (the rule: rounding of the float32 duals), the
    # solve is finished in float64 from the float32 prices -- the matrix widened exactly, every row free, the float64 augmenting row
    # reduction and augmentation.  That IS the warm-started float64 restatement (oracle/jv_oracle.c: jv_oracle_warm_f64) applied to the
    # widened matrix: indices bit for bit, duals after narrowing; and an independent exact solver finds no better total in float64.
    from scipy.optimize import linear_sum_assignment
    from tools import instances
    for c in (instances.typed_unique_cost(2003, 2003, 9)[0], np.random.default_rng(51).random((1500, 1500)).astype(np.float32),
              instances.typed_unique_cost(4700, 4700, 10, K=4)[0]):
        n = len(c)
        c64 = c.astype(np.float64)
        g = lap_solve(c, np.float32, return_info=True, opts=dict(polish=1))
        i = g["info"]
        assert i.certified == 1 and i.gap_f64 >= 0.0 and (i.polished == 1) == (i.gap_f64 > 0.0)
        o = jv_oracle(c64, np.float64, warm=True)
        assert np.array_equal(g["rowsol"], o["rowsol"]) and np.array_equal(g["colsol"], o["colsol"])
        if i.polished:
            assert np.array_equal(g["v"], o["v"].astype(np.float32)) and np.array_equal(g["u"], o["u"].astype(np.float32))
        r, cc = linear_sum_assignment(c64)
        opt = float(c64[r, cc].sum())
        assert abs(float(c64[np.arange(n), g["rowsol"]].sum()) - opt) <= 1e-12 * max(1.0, abs(opt)) and abs(g["total"] - opt) <= 1e-9
    # repeated spot rows through the row map: same spots, the float64 optimum's total
    rows = instances.typed_unique_cost(300, 1200, 11)[0]
    rowmap = np.repeat(np.arange(300), 4).astype(np.int32)
    g = lap_solve_rows(rows, rowmap, return_info=True, opts=dict(polish=1))
    c64 = rows[rowmap].astype(np.float64)
    r, cc = linear_sum_assignment(c64)
    assert abs(float(c64[np.arange(1200), g["rowsol"]].sum()) - float(c64[r, cc].sum())) <= 1e-12 * abs(float(c64[r, cc].sum()))
    assert np.array_equal(np.sort(g["colsol"]), np.arange(1200))


def test_polish_recovers_the_unique_optimum_on_a_near_tie():
    # Found by tools/cross_unique.py in round 6 (instance 283 of tests/golden/cross_unique.npz: few-cell-type, n = 2 973, four cell types):
    # certified unique -- classic oracle == scipy, unchanged by the one-ulp re-solve -- and yet the default float32 solver, kernel and
    # wide restatement alike, ends 4e-9 above the optimum with 75 other indices: the runner-up assignment is closer than the rounding of
    # float32 duals resolves.  What holds for the default solve: the certificate bounds the distance (total - optimum <= gap_f64 <<
    # 1e-5).  What the polish adds: the optimum itself, index for index.
    from tools import cross_unique
    c = cross_unique.instance("typed", 2973, 5283, 4)
    n = len(c)
    c64 = c.astype(np.float64)
    o = jv_oracle(c, np.float32)
    opt = float(c64[np.arange(n), o["rowsol"]].sum())
    g = lap_solve(c, np.float32, return_info=True, opts=dict(certify=1))
    mine = float(c64[np.arange(n), g["rowsol"]].sum())
    assert g["info"].certified == 1 and -1e-12 <= mine - opt <= g["info"].gap_f64 <= 1e-5
    ow = jv_oracle_wide(c, np.float32)
    assert np.array_equal(g["colsol"], ow["colsol"])                       # (bit-exact against ITS restatement as ever)
    p = lap_solve(c, np.float32, return_info=True, opts=dict(polish=1))
    assert p["info"].polished == 1 and np.array_equal(p["colsol"], o["colsol"]) and np.array_equal(p["rowsol"], o["rowsol"])
    assert abs(p["total"] - opt) <= 1e-12 * max(1.0, abs(opt))
