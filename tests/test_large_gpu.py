"""GPU parity at the TRUE sizes of BASELINE.json's single-GPU configs (c1, c2, c3 and one c4 chunk pair).

The large LAPs are compared with tests/golden/large_*.npz: what the CPU JV oracle returned in the build container
for the same seeded instance (tests/golden/make_golden_large.py; certified there by its duals, by scipy and by a
one-ulp perturbation re-solve).  No oracle run at these sizes happens here (it would need minutes), and nothing of
/root/reference is read.  Every call goes through the C ABI (ctypes).
"""
import hashlib
import os

import numpy as np
import pytest

from cytospace_amd import _lib, common
from cytospace_amd.cytospace import ExpressionContext, assign_pearson, solve_linear_assignment_problem
from cytospace_amd.lap import lap_solve, lap_solve_rows
from cytospace_amd.linear_assignment_solvers import calculate_cost, call_solver, import_solver
from oracle import cost as ocost
from oracle.jv import jv_oracle
from tools import instances

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STAT_KEYS = ["scans_colred", "scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax",
             "augmentations", "path_hops", "free_after_colred", "free_after_arr1", "free_after_arr2"]


CHAIN = dict(mode=1)     # the chain solver: what the goldens' duals and counters were made with (oracle classic mode)
WIDE = dict(mode=2)      # the wide solver (the default for one problem): same optimum, its own duals


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _golden(tag):
    path = os.path.join(GOLD, f"large_{tag}.npz")
    assert os.path.exists(path), f"{path} missing: run tests/golden/make_golden_large.py {tag}"
    return np.load(path)


def _compare_with_golden(tag, buf, n, solve=None):
    """HIP solve of the device-resident n x n matrix vs the oracle's stored answer: colsol element for element,
    rowsol / u / v by sha256 (bit-exact without shipping 3 more arrays), the work counters, the total."""
    d = _golden(tag)
    assert int(d["n"]) == n
    g = solve() if solve else lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n, opts=CHAIN)
    assert np.array_equal(g["colsol"], d["colsol"]), f"{(g['colsol'] != d['colsol']).sum()} of {n} columns differ"
    assert sha(g["rowsol"]) == str(d["rowsol_sha256"])
    assert sha(g["v"]) == str(d["v_sha256"]), "dual prices v differ from the oracle's"
    assert sha(g["u"]) == str(d["u_sha256"]), "duals u differ from the oracle's"
    assert abs(g["total"] - float(d["total"])) <= 1e-5 * max(1.0, abs(float(d["total"])))   # BASELINE.json's tolerance
    want = dict(zip([str(k) for k in d["stats_keys"]], d["stats_vals"].tolist()))
    got = g["info"].as_dict()
    for k in STAT_KEYS:
        assert got[k] == want[k], (k, got[k], want[k])
    assert bool(d["unique"]), "golden is not uniqueness-certified"
    return g


@pytest.mark.parametrize("n", [20000, 24000, 30000, 33000, 50000])
def test_uniform_true_size(n):
    """c2 (20 000) and the north star's 50 000, and one size inside every chain variant's own range: <10,true> (20 000),
    <13,true> (24 000), <16,false> with u16 colsol in LDS and the prices in L2 (30 000), <0,false> with the streaming dense
    refresh and build_row_caches_stream (33 000, 50 000)."""
    buf = instances.blocks_to_device(instances.uniform_cost_blocks(n), n)
    try:
        _compare_with_golden(f"u{n}", buf, n)
    finally:
        buf.free()


def _wide_vs_golden(tag, n, solve, loc=None):
    """The wide solver at true size: the golden's indices (any exact solver must return them: the goldens are
    uniqueness-certified; spot level where spot rows are duplicated), a permutation, and its OWN duals certify the optimum --
    feasible on sampled rows, tight on the assignment, strong duality."""
    d = _golden(tag)
    g = solve()
    assert g["info"].wide == 1
    colsol, rowsol = g["colsol"], g["rowsol"]
    assert np.array_equal(np.sort(colsol), np.arange(n)) and np.array_equal(rowsol[colsol], np.arange(n))
    if loc is None:
        assert np.array_equal(colsol, d["colsol"]), f"{(colsol != d['colsol']).sum()} of {n} columns differ"
    else:
        assert np.array_equal(loc[colsol], loc[d["colsol"]])
    assert abs(g["total"] - float(d["total"])) <= 1e-5 * max(1.0, abs(float(d["total"])))
    u, v = g["u"].astype(np.float64), g["v"].astype(np.float64)
    assert abs(g["total"] - (u.sum() + v.sum())) <= 1e-5 * max(1.0, abs(g["total"]))
    wpath = os.path.join(GOLD, f"large_{tag}_wide.npz")
    if os.path.exists(wpath):
        # the wide restatement's own answer for this instance (make_golden_large.py --wide): the slots, the duals and the semantic
        # counters bit for bit -- although the kernel settles columns speculatively, in no fixed order
        dw = np.load(wpath)
        assert np.array_equal(colsol, dw["colsol"])
        assert sha(rowsol) == str(dw["rowsol_sha256"]) and sha(g["u"]) == str(dw["u_sha256"]) and sha(g["v"]) == str(dw["v_sha256"])
        want = dict(zip([str(k) for k in dw["stats_keys"]], dw["stats_vals"].tolist()))
        got = g["info"].as_dict()
        for kg, ko in (("scans_redtransfer", "scans_redtransfer"), ("scans_arr", "scans_arr"), ("scans_aug_relax", "scans_aug_relax"),
                       ("augmentations", "augmentations"), ("path_hops", "path_hops"), ("free_after_arr2", "free_after_arr"),
                       ("wide_rounds", "arr_rounds"), ("wide_retired", "arr_retired"), ("wide_scaled", "arr_scaled"), ("wide_phases", "arr_phases")):
            assert got[kg] == want[ko], (kg, got[kg], want[ko])
    return g


@pytest.mark.parametrize("n", [20000, 33000, 50000, 70000])
def test_wide_uniform_true_size(n):
    if not os.path.exists(os.path.join(GOLD, f"large_u{n}.npz")):
        pytest.fail(f"tests/golden/large_u{n}[_wide].npz is missing (a lost fixture must not silently drop this parity test)")
    buf = instances.blocks_to_device(instances.uniform_cost_blocks(n), n)
    try:
        g = _wide_vs_golden(f"u{n}", n, lambda: lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n))
        assert g["info"].wide_dense_aug < 0.05 * g["info"].scans_aug_relax
    finally:
        buf.free()


@pytest.mark.parametrize("tag", ["t20000", "t10000", "k5t20000", "t30000"])
def test_wide_cytolike_true_size(tag):
    """SURVEY 8(d)'s "cytospace-like" solver-only instances at true size: spots x cells of few cell types, every slot count 1 -- the
    deep-search class CytoSPACE's chunks belong to (the classic oracle needs 40x the row scans of the uniform c2), and the class on
    which float32 near-ties are the rule (pinned at two instances until round 6).  Ten types at 10 000 / 20 000 /
    30 000, five types at 20 000: the default solver's indices == the certified classic golden (== scipy; unique by the one-ulp
    re-solve), its rowsol / duals / counters == the wide restatement's golden, and the float64 certificate of the result holds."""
    import sys
    sys.path.insert(0, GOLD)
    import make_golden_large as mg
    if not (os.path.exists(os.path.join(GOLD, f"large_{tag}.npz")) and os.path.exists(os.path.join(GOLD, f"large_{tag}_wide.npz"))):
        pytest.fail(f"tests/golden/large_{tag}[_wide].npz is missing (a lost fixture must not silently drop this parity test)")
    n, cost, _ = mg.instance(tag)
    buf = _lib.DeviceBuffer.from_numpy(cost)
    del cost
    try:
        g = _wide_vs_golden(tag, n, lambda: lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n, opts=dict(certify=1)))
        i = g["info"]
        assert i.wide_scaled == 1
        assert i.certified == 1 and 0.0 <= i.gap_f64 <= 1e-5 * max(1.0, abs(g["total"]))
    finally:
        buf.free()


def test_wide_c3_shaped_and_c4_chunk_true_size():
    n = 50000
    uniq, loc = instances.c3_shaped_unique(n)
    buf = _lib.DeviceBuffer.from_numpy(uniq)
    try:
        g = _wide_vs_golden(f"c3s{n}", n, lambda: lap_solve_rows(None, loc, return_info=True, device_ptr=buf.ptr, nu=len(uniq), ld=n), loc)
        assert np.array_equal(np.bincount(loc[g["colsol"]], minlength=n // 10), np.full(n // 10, 10))
    finally:
        buf.free()
    n = 10000
    cost, loc = instances.c4_chunk_cost(n)
    buf = _lib.DeviceBuffer.from_numpy(cost)
    try:
        _wide_vs_golden(f"c4s{n}", n, lambda: lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n), loc)
    finally:
        buf.free()


def test_wide_chunk_of_16384_cells_whatever_the_schedule():
    """A 16 384-cell sub-spot chunk (the few-cell-type class at the size from which a search's full-row relaxations go to the whole
    workgroup): the spot of every cell == the certified classic golden (== scipy at spot level; unique by the one-ulp re-solve), rowsol
    / u / v and the semantic counters == the wide restatement's golden -- with the default schedule, with row caches that are never
    rebuilt during the searches, and one search at a time."""
    tag, n = "c4s16384", 16384
    for f in (f"large_{tag}.npz", f"large_{tag}_wide.npz"):
        if not os.path.exists(os.path.join(GOLD, f)):
            pytest.fail(f"tests/golden/{f} is missing (make_golden_large.py [--wide] {tag}): a lost fixture must not silently drop this parity test")
    cost, loc = instances.c4_chunk_cost(n)
    buf = _lib.DeviceBuffer.from_numpy(cost)
    del cost
    try:
        for opts in (dict(), dict(wide_rebuild=-1), dict(wide_rebuild=-1, wide_par=-1)):
            g = _wide_vs_golden(tag, n, lambda: lap_solve(None, np.float32, return_info=True, device_ptr=buf.ptr, n=n, ld=n, opts=opts), loc)
            assert g["info"].wide_scaled == 1
    finally:
        buf.free()


def test_wide_full_row_relaxations_by_the_whole_workgroup():
    """Rows of 16 384 columns and more that a search relaxes in full are swept by the WHOLE workgroup behind the round's barrier
    (lap_wide.hip: coop_dense_rows; shorter rows by the wave that settled the column).  Integer costs with a few dozen distinct
    values tie everywhere: no row cache certifies anything, every column a search settles sends its owner's whole row through that path.
    n = 16 400, against the wide restatement run here (seconds: the instance never scales and most of its searches are one edge): every
    output bit for bit, one search at a time and several at once."""
    from oracle.jv import jv_oracle_wide
    from tools.stress_lap import make
    n = 16400
    c = make("ints", n, np.random.default_rng(777))
    o = jv_oracle_wide(c, np.float32)
    st = o["stats"]
    dense = []
    for opts in (dict(), dict(wide_par=-1), dict(wide_par=5)):
        g = lap_solve(c, np.float32, return_info=True, opts=opts)
        i = g["info"]
        assert i.wide == 1
        for k in ("rowsol", "colsol", "u", "v"):
            assert np.array_equal(g[k], o[k]), (opts, k)
        assert (i.scans_arr, i.scans_aug_relax, i.wide_rounds, i.wide_retired, i.path_hops, i.wide_scaled, i.wide_phases) == \
               (st.scans_arr, st.scans_aug_relax, st.arr_rounds, st.arr_retired, st.path_hops, st.arr_scaled, st.arr_phases), opts
        dense.append(int(i.wide_dense_aug))
    assert min(dense) > 0, dense                                   # the cooperative path really ran


def test_chain_uniform_70000_colsol_in_global_memory():
    """n > 65 535: the chain kernels with colsol in global memory too (chain_variant 3's range) at a true size, 19.6 GB of cost."""
    n = 70000
    if not os.path.exists(os.path.join(GOLD, f"large_u{n}.npz")):
        pytest.fail(f"tests/golden/large_u{n}[_wide].npz is missing (a lost fixture must not silently drop this parity test)")
    buf = instances.blocks_to_device(instances.uniform_cost_blocks(n), n)
    try:
        _compare_with_golden(f"u{n}", buf, n)
    finally:
        buf.free()


def test_float64_uniform_17000_prices_in_l2():
    """float64 beyond ~15 800 columns: the streaming chain with the prices in L2 (not LDS) during row reduction, at a true size."""
    n = 17000
    d = _golden(f"u{n}_f64")
    c = instances.uniform_cost(n).astype(np.float64)
    g = lap_solve(c, np.float64, return_info=True, opts=CHAIN)          # the cold classic chain (the default is warm-started)
    assert np.array_equal(g["colsol"], d["colsol"])
    assert sha(g["rowsol"]) == str(d["rowsol_sha256"]) and sha(g["u"]) == str(d["u_sha256"]) and sha(g["v"]) == str(d["v_sha256"])
    assert abs(g["total"] - float(d["total"])) <= 1e-9 * max(1.0, abs(float(d["total"])))
    want = dict(zip([str(k) for k in d["stats_keys"]], d["stats_vals"].tolist()))
    got = g["info"].as_dict()
    for k in STAT_KEYS:
        assert got[k] == want[k], (k, got[k], want[k])


def test_float64_uniform_17000_warm_start():
    """The float64 default at a true size: warm-started from the float32 wide solve of the narrowed matrix (the matrix carries a
    term below float32's resolution, so the float64 problem is not the float32 one)."""
    n = 17000
    path = os.path.join(GOLD, f"large_u{n}_f64_warm.npz")
    if not os.path.exists(path):
        pytest.fail("tests/golden/large_u17000_f64_warm.npz is missing (a lost fixture must not silently drop this parity test)")
    d = np.load(path)
    c = instances.uniform_cost(n).astype(np.float64)
    c += np.random.default_rng(n).random((n, n)) * 2.0 ** -30
    g = lap_solve(c, np.float64, return_info=True)
    assert g["info"].f64_warm == 1
    assert np.array_equal(g["colsol"], d["colsol"])
    assert sha(g["rowsol"]) == str(d["rowsol_sha256"]) and sha(g["u"]) == str(d["u_sha256"]) and sha(g["v"]) == str(d["v_sha256"])
    assert abs(g["total"] - float(d["total"])) <= 1e-9 * max(1.0, abs(float(d["total"])))
    want = dict(zip([str(k) for k in d["stats_keys"]], d["stats_vals"].tolist()))
    got = g["info"].as_dict()
    for k in STAT_KEYS:
        assert got[k] == want[k], (k, got[k], want[k])
    assert g["info"].ms_total < 1000.0


def test_c3_shaped_lap_50000():
    """c3's LAP shape: 5 000 spot rows x 10 slots against 50 000 cells (duplicate-row elision, sparse inits)."""
    n = 50000
    uniq, loc = instances.c3_shaped_unique(n)
    buf = instances.blocks_to_device(instances.repeated_row_blocks(uniq, loc), n)
    try:
        g = _compare_with_golden(f"c3s{n}", buf, n)
        assert np.array_equal(np.bincount(loc[g["colsol"]], minlength=n // 10), np.full(n // 10, 10))
    finally:
        buf.free()


def test_c3_shaped_lap_50000_through_the_row_map():
    """The same LAP with the cost stored as its 5 000 distinct rows (1 GB instead of 10 GB) + location_repeat as the row map
    (cyto_lap_f32_rowmap): the oracle's answer for the materialised matrix, bit for bit."""
    n = 50000
    uniq, loc = instances.c3_shaped_unique(n)
    buf = _lib.DeviceBuffer.from_numpy(uniq)
    try:
        _compare_with_golden(f"c3s{n}", None, n, solve=lambda: lap_solve_rows(None, loc, return_info=True, device_ptr=buf.ptr,
                                                                              nu=len(uniq), ld=n, opts=CHAIN))
    finally:
        buf.free()


def test_c4_chunk_lap_10000():
    """One --sampling-sub-spots chunk of c4: 10 000 cells of 10 types, deep searches (hand-over to the dense kernel)."""
    n = 10000
    cost, loc = instances.c4_chunk_cost(n)
    buf = _lib.DeviceBuffer.from_numpy(cost)
    try:
        _compare_with_golden(f"c4s{n}", buf, n)
    finally:
        buf.free()


# ---- c1: 1k cells x 1k spots x 2k genes through the plug-in surface (linear_assignment_solvers.py:11-69) ----

def test_c1_plugin_surface():
    G, C, S = 2000, 1000, 1000
    sc, st, slots = instances.synth_expression(G, C, S, seed=11, dtype=np.float64)
    scn, stn = common.normalize_data(sc), common.normalize_data(st)
    np.testing.assert_allclose(scn, ocost.normalize_data(sc), rtol=1e-12, atol=1e-12)
    solver = import_solver("lapjv_hip")
    dist, loc = calculate_cost(scn, stn, slots, "lapjv_hip", "Pearson_correlation")
    ref, ref_loc = ocost.calculate_cost(ocost.normalize_data(sc), ocost.normalize_data(st), slots)
    assert np.array_equal(loc, ref_loc)
    np.testing.assert_allclose(dist, ref, rtol=0, atol=2e-6)
    y = call_solver(solver, "lapjv_hip", dist)
    o = jv_oracle(dist, np.float32)
    assert np.array_equal(y, o["colsol"])                       # same cost matrix: bit-exact vs the CPU oracle
    mapped, idx = solve_linear_assignment_problem(scn, stn, slots, "lapjv_hip", solver, 1, "Pearson_correlation", process_idx=7)
    assert idx == 7 and len(mapped) == C
    assert np.array_equal(np.bincount(mapped, minlength=S), slots)
    # the fused path and the split path agree; the total on the float64 reference cost is the optimum within 1e-5
    cols = np.arange(C)
    ro = jv_oracle(ref.astype(np.float32), np.float32)
    best = ref[ro["colsol"], cols].sum()
    mine = ref[np.asarray(mapped), cols].sum()      # slots == 1: the spot index is the row index
    assert abs(mine - best) <= 1e-5 * max(1.0, abs(best))
    assert np.array_equal(np.asarray(mapped), loc[y])


# ---- c3 at full size: 50k cells x 5k spots x 20k genes, MFMA cost GEMM + LAP ----

def test_c3_full_size_pipeline():
    G, C, S = 20000, 50000, 5000
    sc, st, slots = instances.synth_expression(G, C, S, seed=1)          # float32 counts (exact)
    cost, N, ld, gemm_ms = common.pearson_cost_device(sc, st, slots, already_normalized=False)
    try:
        assert N == C and ld >= C
        g = lap_solve(None, np.float32, return_info=True, device_ptr=cost.ptr, n=N, ld=ld)
        # (1) a permutation, every spot filled exactly
        assert np.array_equal(np.sort(g["colsol"]), np.arange(N))
        assert np.array_equal(g["rowsol"][g["colsol"]], np.arange(N))
        loc = np.repeat(np.arange(S), slots)
        mapped = loc[g["colsol"]]
        assert np.array_equal(np.bincount(mapped, minlength=S), slots)
        # (2) on a sample of spots: the device cost rows against the float64 reference formula
        #     (common.py:142-147, 190-199 restated in oracle/cost.py), and the optimality conditions ON THE REFERENCE COST
        rs = np.random.default_rng(0).choice(S, 96, replace=False)
        stn = ocost.normalize_data(st[:, rs].astype(np.float64))
        ref = np.empty((len(rs), C))
        for lo in range(0, C, 5000):            # cells are independent columns: block-wise keeps the float64 copy small
            ref[:, lo:lo + 5000] = -ocost.matrix_correlation_pearson(ocost.normalize_data(sc[:, lo:lo + 5000].astype(np.float64)), stn)
        first = np.concatenate([[0], np.cumsum(slots)[:-1]])
        rows = np.empty((len(rs), ld), np.float32)
        for k, s_ in enumerate(rs):
            _lib.check(_lib.lib().cyto_memcpy_d2h(rows[k].ctypes.data, cost.ptr + int(first[s_]) * ld * 4, ld * 4, 0))
        np.testing.assert_allclose(rows[:, :C], ref, rtol=0, atol=2e-6)
        u, v = g["u"].astype(np.float64), g["v"].astype(np.float64)
        for k, s_ in enumerate(rs):
            for i in range(first[s_], first[s_] + slots[s_]):
                red = ref[k] - u[i] - v
                assert red.min() > -1e-5                          # dual feasible on the float64 reference cost
                assert abs(red[g["rowsol"][i]]) < 1e-5            # complementary slackness
        # (3) strong duality: the primal total equals the dual objective (within float32 rounding of 50 000 duals)
        assert abs(g["total"] - (u.sum() + v.sum())) <= 1e-5 * max(1.0, abs(g["total"]))
    finally:
        cost.free()
    # (4) the fused entry point (cyto_assign_metric_typed) gives the same spots and the same total
    m2, tot2, info = assign_pearson(sc, st, slots, already_normalized=False, return_info=True)
    assert np.array_equal(m2, mapped)
    assert abs(tot2 - g["total"]) <= 1e-5 * max(1.0, abs(g["total"]))
    assert info.gemm_flops == 2.0 * G * S * C


# ---- c4-shaped: two --sampling-sub-spots chunks of 10 000 cells against 50 000 spots on one GPU ----

def test_c4_two_chunks_of_10000():
    G, C, S, chunk = 2000, 200000, 50000, 10000
    sc, st, slots = instances.synth_expression(G, C, S, seed=2)
    rng = np.random.default_rng(0)
    slot_ids = rng.permutation(np.repeat(np.arange(S), slots))          # which spot every cell's slot belongs to
    idx = [np.arange(k * chunk, (k + 1) * chunk) for k in range(2)]
    sub = [np.bincount(slot_ids[ix], minlength=S) for ix in idx]       # per-chunk slot counts (cytospace.py:436-439)
    used = np.unique(np.concatenate(idx))
    sc_used = np.ascontiguousarray(sc[:, used])
    from concurrent.futures import ThreadPoolExecutor
    with ExpressionContext(sc_used, st, already_normalized=False) as ctx:
        with ThreadPoolExecutor(2) as ex:
            res = list(ex.map(lambda k: ctx.assign_chunk(np.searchsorted(used, idx[k]), sub[k], return_info=True), range(2)))
    scn = ocost.normalize_data(sc_used.astype(np.float64))
    stn = ocost.normalize_data(st.astype(np.float64))
    for k, (mapped, total, info) in enumerate(res):
        assert np.array_equal(np.bincount(mapped, minlength=S), sub[k])
        spots = np.flatnonzero(sub[k])
        ref = -ocost.matrix_correlation_pearson(scn[:, np.searchsorted(used, idx[k])], stn[:, spots])    # S_u x 10000 float64
        pos = np.searchsorted(spots, mapped)
        mine = ref[pos, np.arange(chunk)].sum()
        o = jv_oracle(ref[np.repeat(np.arange(len(spots)), sub[k][spots])].astype(np.float32), np.float32)
        best = float(o["total"])
        assert abs(mine - best) <= 1e-5 * max(1.0, abs(best)), (k, mine, best)
        assert abs(total - best) <= 1e-5 * max(1.0, abs(best))


# ---- c5-shaped: --single-cell chunks (slots == 1, every chunk against its own subset of the spots; cytospace.py:598-640) ----

def test_c5_shaped_single_cell_chunks():
    G, C, S, chunk = 1500, 15000, 15000, 5000
    sc, st, slots = instances.synth_expression(G, C, S, seed=5)          # slots == 1 everywhere: one cell per spot
    assert (slots == 1).all()
    rng = np.random.default_rng(1)
    cells, spots = rng.permutation(C), rng.permutation(S)
    idx_sc = [np.sort(cells[k * chunk:(k + 1) * chunk]) for k in range(3)]
    idx_st = [np.sort(spots[k * chunk:(k + 1) * chunk]) for k in range(3)]
    with ExpressionContext(sc, st, already_normalized=False) as ctx:
        res = ctx.assign_chunks([(idx_sc[k], np.ones(chunk, np.int64), idx_st[k]) for k in range(3)], return_info=True)
    stn = ocost.normalize_data(st.astype(np.float64))
    scn = ocost.normalize_data(sc.astype(np.float64))
    for k, (mapped, total, info) in enumerate(res):
        assert np.array_equal(np.sort(mapped), np.arange(chunk))          # positions in the chunk's spot list: a permutation
        ref = -ocost.matrix_correlation_pearson(scn[:, idx_sc[k]], stn[:, idx_st[k]])     # chunk spots x chunk cells, float64
        o = jv_oracle(ref.astype(np.float32), np.float32)
        best = float(ref[o["colsol"], np.arange(chunk)].sum())
        mine = float(ref[mapped, np.arange(chunk)].sum())
        assert abs(mine - best) <= 1e-5 * max(1.0, abs(best)), (k, mine, best)
        assert info.lap.row_groups == chunk                                # no duplicated rows in single-cell mode


# ---- the fused path's block pipeline (more than 16 384 cells: upload + transform of block b + 1 next to the contraction of block b) ----

@pytest.mark.parametrize("metric", ["Pearson_correlation", "Spearman_correlation"])
def test_fused_block_pipeline_equals_the_split_path(metric):
    G, C, S = 240, 17200, 1720                       # three blocks of cells (8192, 8192, 816)
    sc, st, slots = instances.synth_expression(G, C, S, seed=11)
    mapped, total, info = assign_pearson(sc, st, slots, already_normalized=False, return_info=True, distance_metric=metric)
    assert np.array_equal(np.bincount(mapped, minlength=S), slots)
    cost, N, ld, _ = common.pearson_cost_device(sc, st, np.ones(S, np.int64), already_normalized=False, metric=metric)
    try:
        rows = cost.to_numpy((S, ld), np.float32)[:, :C]
    finally:
        cost.free()
    loc = np.repeat(np.arange(S), slots)
    g = lap_solve_rows(rows, loc)                                        # the same unique rows, built in one piece
    assert np.array_equal(loc[g["colsol"]], mapped)
    assert abs(g["total"] - total) <= 1e-5 * max(1.0, abs(total))
    # the same counts as uint16 / uint8 (CYTO_DTYPE_*): the block pipeline uploads half / a quarter of the bytes, same mapping, same total
    assert sc.max() < 256 and st.max() < 65536
    m2, t2, _ = assign_pearson(sc.astype(np.uint8), st.astype(np.uint16), slots, already_normalized=False, return_info=True, distance_metric=metric)
    assert np.array_equal(m2, mapped) and t2 == total


def test_fused_block_pipeline_failure_then_another_thread_solves():
    """Fault injection into the block pipeline of the fused path: a cell without counts in the SECOND of three blocks (zero
    variance: its standardised column is NaN, cytospace/common/common.py:190-199 divides by the zero std just the same) makes
    the call fail while contractions of other blocks were queued on the pipeline's second stream.  Every stream of the call has
    drained before its buffers go back to the block cache, so solves on other threads -- which are handed those blocks at
    once -- stay correct, during the failures and after them."""
    import threading
    G, C, S = 240, 17200, 1720
    sc, st, slots = instances.synth_expression(G, C, S, seed=12)
    want, want_total, _ = assign_pearson(sc, st, slots, already_normalized=False, return_info=True)
    bad = sc.copy()
    bad[:, 8192 + 77] = 0                                # (block 1 of 0..2)
    errors, results = [], []

    def failing():
        for _ in range(4):
            try:
                assign_pearson(bad, st, slots, already_normalized=False)
                errors.append("no exception")
            except ValueError:
                pass
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))

    def solving():
        for _ in range(4):
            m, t, _ = assign_pearson(sc, st, slots, already_normalized=False, return_info=True)
            results.append(bool(np.array_equal(m, want) and abs(t - want_total) <= 1e-9 * max(1.0, abs(want_total))))

    th = [threading.Thread(target=failing), threading.Thread(target=solving), threading.Thread(target=solving)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(results) == 8 and all(results)
    m, t, _ = assign_pearson(sc, st, slots, already_normalized=False, return_info=True)        # ... and afterwards
    assert np.array_equal(m, want)


# ---- configs[3] as configured: 200 000 cells -> 20 sub-spot chunks of 10 000 cells against 50 000 spots, ONE batched call ----

def test_c4_all_20_chunks_in_one_call():
    G, C, S, chunk = 2000, 200000, 50000, 10000
    sc, st, slots = instances.synth_expression(G, C, S, seed=2)
    rng = np.random.default_rng(0)
    slot_ids = rng.permutation(np.repeat(np.arange(S), slots))          # which spot every cell's slot belongs to
    nchunks = C // chunk
    idx = [np.arange(k * chunk, (k + 1) * chunk) for k in range(nchunks)]
    sub = [np.bincount(slot_ids[ix], minlength=S) for ix in idx]       # per-chunk slot counts (cytospace.py:436-439)
    assert sum(s_.sum() for s_ in sub) == C and np.array_equal(sum(sub), slots)
    with ExpressionContext(sc, st, already_normalized=False) as ctx:
        res = ctx.assign_chunks([(idx[k], sub[k]) for k in range(nchunks)], max_concurrent=nchunks, return_info=True)
    assert len(res) == nchunks
    for k, (mapped, total, info) in enumerate(res):
        assert np.array_equal(np.bincount(mapped, minlength=S), sub[k]), k       # every chunk fills exactly its sub-spot slots
        assert info.lap.wide == 1
    # all chunks together: every spot receives exactly its cells (apply_linear_assignment concatenates: cytospace.py:453-467)
    assert np.array_equal(np.bincount(np.concatenate([m for m, _, _ in res]), minlength=S), slots)
    # the optimum, on the float64 reference cost, for two of the chunks
    scn_all = None
    stn = ocost.normalize_data(st.astype(np.float64))
    for k in (3, 17):
        mapped, total, info = res[k]
        scn = ocost.normalize_data(sc[:, idx[k]].astype(np.float64))
        spots = np.flatnonzero(sub[k])
        ref = -ocost.matrix_correlation_pearson(scn, stn[:, spots])
        pos = np.searchsorted(spots, mapped)
        mine = ref[pos, np.arange(chunk)].sum()
        o = jv_oracle(ref[np.repeat(np.arange(len(spots)), sub[k][spots])].astype(np.float32), np.float32)
        best = float(o["total"])
        assert abs(mine - best) <= 1e-5 * max(1.0, abs(best)), (k, mine, best)
        assert abs(total - best) <= 1e-5 * max(1.0, abs(best))
    del scn_all


# ---- configs[4]'s LAP work at scale: 500 000 cells = 50 single-cell-mode chunks of 10 000 cells x 10 000 spots, ONE batched call ----

def test_c5_50_chunks_of_10000_in_one_call():
    G, chunk, sets, K = 500, 10000, 8, 50
    sc, st = instances.single_cell_expression(G, sets * chunk, sets * chunk, seed=5)
    ones = np.ones(chunk, np.int64)
    work = [(np.arange((k % sets) * chunk, (k % sets + 1) * chunk), ones,
             np.arange(((k // sets) % sets) * chunk, ((k // sets) % sets + 1) * chunk)) for k in range(K)]
    with ExpressionContext(sc, st, already_normalized=False) as ctx:
        res = ctx.assign_chunks(work, max_concurrent=K, return_info=True)
    assert len(res) == K
    for k, (mapped, total, info) in enumerate(res):
        assert np.array_equal(np.sort(mapped), np.arange(chunk)), k       # positions in the chunk's spot list: a permutation
        assert info.lap.wide == 1 and info.lap.row_groups == chunk
    scn = ocost.normalize_data(sc.astype(np.float64))
    stn = ocost.normalize_data(st.astype(np.float64))
    for k in (7, 42):
        mapped, total, info = res[k]
        ref = -ocost.matrix_correlation_pearson(scn[:, work[k][0]], stn[:, work[k][2]])     # chunk spots x chunk cells, float64
        o = jv_oracle(ref.astype(np.float32), np.float32)
        best = float(ref[o["colsol"], np.arange(chunk)].sum())
        mine = float(ref[mapped, np.arange(chunk)].sum())
        assert abs(mine - best) <= 1e-5 * max(1.0, abs(best)), (k, mine, best)
