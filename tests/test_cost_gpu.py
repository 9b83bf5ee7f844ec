"""GPU parity of the cost build and the fused chunk solve against the golden vectors captured from
the reference and against the numpy/C oracle."""
import os
import time

import numpy as np
import pytest

from cytospace_amd import common as gcommon
from cytospace_amd import cytospace as gcyto
from cytospace_amd import linear_assignment_solvers as gsolvers
from oracle import cost as ocost
from oracle.jv import jv_oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return np.load(os.path.join(GOLD, name))


def test_gv1_normalize_data():
    for f in ("gv1_normalize.npz", "gv1b_normalize_int.npz"):
        d = load(f)
        out = gcommon.normalize_data(d["counts"])
        assert out.dtype == np.float64 and out.shape == d["out"].shape
        np.testing.assert_allclose(out, d["out"], rtol=1e-12, atol=1e-12)     # float64 on the device
    d = load("gv1_normalize.npz")
    out = gcommon.normalize_data(d["counts"])
    assert np.all(out[:, 4] == 0.0) and np.all(np.isfinite(out))


def test_gv2_pearson_correlation():
    d = load("gv2_pearson.npz")
    corr = gcommon.matrix_correlation_pearson(d["sc_norm"], d["st_norm"])
    assert corr.shape == d["corr"].shape
    np.testing.assert_allclose(corr, d["corr"], rtol=0, atol=2e-6)            # fp32 contraction tolerance
    with pytest.raises(ValueError):
        gcommon.matrix_correlation_pearson(d["sc_norm"][:-1], d["st_norm"])


def test_gv3_calculate_cost_rows_and_slots():
    d = load("gv3_calculate_cost.npz")
    dist, loc = gsolvers.calculate_cost(d["sc_norm"], d["st_norm"], d["slots"], "lapjv_hip", "Pearson_correlation")
    assert np.array_equal(loc, d["location_repeat"])
    assert dist.shape == d["distance_repeat"].shape
    np.testing.assert_allclose(dist, d["distance_repeat"], rtol=0, atol=2e-6)
    # repeated rows are bit-identical copies of their spot row
    for r in range(1, len(loc)):
        if loc[r] == loc[r - 1]:
            assert np.array_equal(dist[r], dist[r - 1])


@pytest.mark.parametrize("G,S,C", [(20, 3, 4), (33, 5, 7), (90, 129, 131), (200, 130, 260), (1000, 257, 513)])      # 1, 2, 3, 7, 32 k-tiles
def test_cost_vs_numpy_odd_shapes(G, S, C):
    rng = np.random.default_rng(G + S + C)
    sc = rng.poisson(3.0, (G, C)).astype(np.float64)
    st = rng.poisson(9.0, (G, S)).astype(np.float64)
    scn, stn = ocost.normalize_data(sc), ocost.normalize_data(st)
    ref = ocost.matrix_correlation_pearson(scn, stn)
    got = gcommon.matrix_correlation_pearson(scn, stn)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)


def test_standardize_from_raw_counts_matches_two_step():
    rng = np.random.default_rng(4)
    x = rng.poisson(2.0, (150, 70)).astype(np.float64)
    z1 = gcommon.StandardizedMatrix(x, already_normalized=False).to_numpy()
    y = ocost.normalize_data(x)
    z2 = (y - y.mean(0)) / (y.std(0) * np.sqrt(x.shape[0]))
    np.testing.assert_allclose(z1, z2, rtol=0, atol=1e-6)


@pytest.mark.parametrize("key", ["visium_s1", "visium_s7", "single_s1", "single_s7"])
def test_gv5_fused_chunk_solve_spot_level(key):
    d = load("gv5_solve_lap.npz")
    slots = d[key + "_slots"]
    mapped, pidx = gcyto.solve_linear_assignment_problem(d[key + "_sc_norm"], d[key + "_st_norm"], slots, "lapjv_hip",
                                                         gsolvers.import_solver("lapjv_hip"), 1, "Pearson_correlation",
                                                         process_idx=3)
    assert pidx == 3 and len(mapped) == int(slots.sum())
    assert np.array_equal(np.bincount(mapped, minlength=len(slots)), slots)
    assert np.array_equal(np.asarray(mapped), d[key + "_mapped"])            # spot level == reference + exact solver


def test_fused_total_cost_on_reference_cost_matrix():
    # total cost of the GPU assignment evaluated on the float64 reference cost within 1e-5 of the optimum
    d = load("gv5_solve_lap.npz")
    key = "single_s1"
    dist, loc = ocost.calculate_cost(d[key + "_sc_norm"], d[key + "_st_norm"], d[key + "_slots"])
    mapped, total, info = gcyto.assign_pearson(d[key + "_sc_norm"], d[key + "_st_norm"], d[key + "_slots"], return_info=True)
    o = jv_oracle(dist, np.float64)
    # slots are all 1 here: row == spot
    mine = dist[mapped, np.arange(len(mapped))].sum()
    assert abs(mine - o["total"]) <= 1e-5 * max(1.0, abs(o["total"]))
    assert abs(total - o["total"]) <= 1e-4
    assert info.gemm_flops > 0 and info.lap.scans_colred == len(mapped)


def test_zero_variance_column_is_reported():
    rng = np.random.default_rng(1)
    sc = rng.random((50, 20))
    st = rng.random((50, 20))
    sc[:, 3] = 0.25                         # zero variance -> reference divides by zero (common.py:197)
    with pytest.raises(ValueError):
        gcyto.assign_pearson(sc, st, np.ones(20, np.int64))


def test_non_square_chunk_rejected():
    rng = np.random.default_rng(2)
    with pytest.raises(ValueError):
        gcyto.assign_pearson(rng.random((30, 10)), rng.random((30, 4)), np.array([1, 2, 3, 3]))


def test_plugin_surface_roundtrip():
    # cost built on the device, perturbed and solved through import_solver/call_solver like the reference does
    d = load("gv5_solve_lap.npz")
    key = "visium_s7"
    solver = gsolvers.import_solver("lapjv_hip")
    dist, loc = gsolvers.calculate_cost(d[key + "_sc_norm"], d[key + "_st_norm"], d[key + "_slots"], "lapjv_hip",
                                        "Pearson_correlation")
    np.random.seed(7)
    cost_scaled = dist + 1e-16 * np.random.rand(*dist.shape)
    y = gsolvers.call_solver(solver, "lapjv_hip", cost_scaled)
    assert np.array_equal(loc[y], d[key + "_mapped"])
    with pytest.raises(NotImplementedError):
        gsolvers.import_solver("nonsense")


def test_assign_chunks_concurrent_matches_sequential():
    # sub-spot sampling style chunks (cytospace.py:650-660): every chunk sees the full ST matrix
    rng = np.random.default_rng(3)
    G, S, k = 120, 30, 4
    sc = ocost.normalize_data(rng.poisson(3.0, (G, S * k)).astype(np.float64))
    st = ocost.normalize_data(rng.poisson(9.0, (G, S)).astype(np.float64))
    index_sc = gcyto.partition_indices(np.arange(S * k), split_by_interval_int=40, shuffle=False)
    slot_ids = np.repeat(np.arange(S), k)
    slots_list = [np.bincount(slot_ids[ix], minlength=S) for ix in index_sc]
    res = gcyto.assign_chunks(sc, st, None, index_sc, subsampled_slots_list=slots_list, max_concurrent=3)
    assert sorted(res) == list(range(len(index_sc)))
    for idx, ix in enumerate(index_sc):
        ref = gcyto.assign_pearson(sc[:, ix], st, slots_list[idx])
        assert np.array_equal(res[idx], ref)
        assert np.array_equal(np.bincount(res[idx], minlength=S), slots_list[idx])
    # two ranks: every chunk is solved exactly once
    r0 = gcyto.assign_chunks(sc, st, None, index_sc, subsampled_slots_list=slots_list, rank=0, world_size=2)
    r1 = gcyto.assign_chunks(sc, st, None, index_sc, subsampled_slots_list=slots_list, rank=1, world_size=2)
    assert sorted(list(r0) + list(r1)) == list(range(len(index_sc)))


# ---- Spearman_correlation and Euclidean through the same contraction ----

def test_gv2b_spearman_correlation():
    d = load("gv2b_spearman.npz")
    corr = gcommon.matrix_correlation_spearman(d["sc_norm"], d["st_norm"])
    np.testing.assert_allclose(corr, d["corr"], rtol=0, atol=2e-6)     # ranks are exact; fp32 contraction tolerance


@pytest.mark.parametrize("tag,metric,rtol,atol", [("spearman", "Spearman_correlation", 0.0, 2e-6),
                                                  ("euclidean", "Euclidean", 2e-6, 1e-5)])
def test_gv9_calculate_cost_other_metrics(tag, metric, rtol, atol):
    d = load("gv9_metrics_cost.npz")
    dist, loc = gsolvers.calculate_cost(d["sc_norm"], d["st_norm"], d["slots"], "lapjv_hip", metric)
    assert np.array_equal(loc, d[tag + "_location_repeat"]) and dist.shape == d[tag + "_distance_repeat"].shape
    np.testing.assert_allclose(dist, d[tag + "_distance_repeat"], rtol=rtol, atol=atol)
    # repeated spot rows are bit-identical copies
    for s in np.flatnonzero(d["slots"] > 1):
        rows = np.flatnonzero(loc == s)
        assert all(np.array_equal(dist[rows[0]], dist[r]) for r in rows[1:])
    with pytest.raises(ValueError):
        gsolvers.calculate_cost(d["sc_norm"], d["st_norm"], d["slots"], "lapjv_hip", "Manhattan")


@pytest.mark.parametrize("G", [5, 1000, 1025, 4100])
def test_spearman_ranks_exact_with_heavy_ties(G):
    # count data: most entries tie at 0; the device ranks (average ties, float64 compares) must equal the oracle's
    rng = np.random.default_rng(G)
    sc = rng.poisson(0.7, (G, 7)).astype(np.float64)
    st = rng.poisson(2.0, (G, 3)).astype(np.float64)
    got = gcommon.matrix_correlation_spearman(sc, st)
    want = ocost.matrix_correlation_spearman(sc, st)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


@pytest.mark.parametrize("tag,metric", [("spearman", "Spearman_correlation"), ("euclidean", "Euclidean")])
def test_gv10_fused_solve_other_metrics_spot_level(tag, metric):
    d = load("gv10_metrics_solve.npz")
    mapped, pidx = gcyto.solve_linear_assignment_problem(d["sc_norm"], d["st_norm"], d["slots"], "lapjv_hip", None, 1,
                                                         metric, process_idx=5)
    assert pidx == 5
    mapped = np.asarray(mapped)
    assert np.array_equal(np.bincount(mapped, minlength=len(d["slots"])), d["slots"])
    # spot level == reference + exact solver, or (degenerate optimum) the same total on the reference's float64 cost
    if not np.array_equal(mapped, d[tag + "_mapped"]):
        cost64, _ = ocost.calculate_cost(d["sc_norm"], d["st_norm"], np.ones(len(d["slots"]), np.int64), "lapjv", metric)
        cells = np.arange(len(mapped))
        assert abs(cost64[mapped, cells].sum() - cost64[d[tag + "_mapped"], cells].sum()) <= 1e-5 * max(1.0, abs(cost64[mapped, cells].sum()))


@pytest.mark.parametrize("mode", ["sc", "ss"])
def test_gv11_apply_linear_assignment_chunked_modes(mode):
    # A8: the reference's apply_linear_assignment (chunk fan-out) on the same DataFrames; the reference concatenates in
    # completion order, so compare as a set of (cell, spot coordinates) pairs
    import pandas as pd
    d = load("gv11_apply_linear_assignment.npz")
    sc, st, slots = d[mode + "_counts"], d[mode + "_st_counts"], d[mode + "_slots"]
    G, C = sc.shape
    S = st.shape[1]
    sc_df = pd.DataFrame(sc, index=[f"g{i}" for i in range(G)], columns=[f"c{i}" for i in range(C)])
    st_df = pd.DataFrame(st, index=sc_df.index, columns=[f"s{i}" for i in range(S)])
    ncol = 10 if mode == "sc" else 4
    coords = pd.DataFrame({"row": np.arange(S) // ncol, "col": np.arange(S) % ncol}, index=st_df.columns)
    idx_sc = np.split(d[mode + "_idx_sc"], np.cumsum(d[mode + "_idx_sc_lens"])[:-1])
    kw = {}
    if mode == "sc":
        kw["index_st_list"] = np.split(d["sc_idx_st"], np.cumsum(d["sc_idx_st_lens"])[:-1])
    else:
        kw["subsampled_cell_number_to_node_assignment_list"] = list(d["ss_sub"])
    loc, ids = gcyto.apply_linear_assignment(sc_df, st_df, coords, slots, "lapjv_hip", None, 1, "Pearson_correlation", 2,
                                             idx_sc, **kw)
    got = {(int(c[1:]), int(r), int(k)) for c, (r, k) in zip(ids, loc.to_numpy())}
    want = {(int(c), int(r), int(k)) for c, (r, k) in zip(d[mode + "_out_cell"], d[mode + "_out_rowcol"])}
    assert len(got) == len(ids) == len(want) and got == want
    # two ranks split the chunks and together give the same pairs
    parts = [gcyto.apply_linear_assignment(sc_df, st_df, coords, slots, "lapjv_hip", None, 1, "Pearson_correlation", 2,
                                           idx_sc, rank=r, world_size=2, **kw) for r in range(2)]
    got2 = {(int(c[1:]), int(r), int(k)) for loc2, ids2 in parts for c, (r, k) in zip(ids2, loc2.to_numpy())}
    assert got2 == want


@pytest.mark.parametrize("G", [9000, 20000, 36601])
def test_spearman_ranks_many_genes(G):
    # unfiltered 10x gene sets have 33 538 / 36 601 genes: several sorted chunks per column (G > 8 192), every MOWN variant,
    # and two launches beyond 32 768 genes.  Against pandas' rank() (average ties, 1-based), heavy ties included.
    import pandas as pd
    rng = np.random.default_rng(G)
    x = rng.poisson(0.7, (G, 5)).astype(np.float64)             # mostly 0 / 1 / 2: long tie runs
    x[:, 3] = rng.random(G)                                      # one column without ties
    x[:, 4] = np.round(rng.normal(size=G), 2)
    z = gcommon.StandardizedMatrix(x, True, 0, "Spearman_correlation").to_numpy()
    want = pd.DataFrame(x).rank().to_numpy()
    want = (want - want.mean(0)) / (want.std(0) * np.sqrt(G))    # the operand holds the standardised ranks
    np.testing.assert_allclose(z, want, rtol=0, atol=2e-7)
    np.testing.assert_array_equal(ocost.rank_columns(x), pd.DataFrame(x).rank().to_numpy())


def test_sparse_counts_are_expanded_on_the_device():
    # sparse inputs (scipy.sparse, what scipy.io.mmread of a 10x matrix holds) go up as non-zeros and are
    # expanded by cyto_csc_to_dense_f32; same assignments as the dense upload (raw counts, normalised on the device)
    import scipy.sparse as sp
    from cytospace_amd import _lib
    d, idx_sc = _gv11_ss()
    sc, st = d["ss_counts"].astype(np.float32), d["ss_st_counts"].astype(np.float32)
    subs = list(d["ss_sub"])
    chunks = [(ix, subs[k]) for k, ix in enumerate(idx_sc)]
    with gcyto.ExpressionContext(sc, st, False, 0) as dense:
        want = dense.assign_chunks(chunks)
    for a, b in ((sp.coo_matrix(sc), sp.csr_matrix(st)), (sp.csc_matrix(sc), st)):
        with gcyto.ExpressionContext(a, b, False, 0) as ctx:
            got = ctx.assign_chunks(chunks)
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
    buf, G, C, ld = gcommon.sparse_to_device(sp.csr_matrix(sc))
    assert np.array_equal(buf.to_numpy((G, ld), np.float32)[:, :C], sc)
    buf.free()
    bad = sp.csc_matrix(sc)
    bad.indices = bad.indices.copy()
    bad.indices[0] = G + 5                                     # a row index outside the matrix: ValueError, not a wild write
    with pytest.raises(ValueError):
        gcommon.sparse_to_device(bad)


def test_lap_cspr_branch_integerised_cost():
    # cytospace.py:334-347 + linear_assignment_solvers.py:72-96: cost 10^6 d + 10 rand + 1 truncated to integers, workers =
    # cells; OR-tools is not available, the exact integer optimum comes from scipy on the same integer matrix
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(21)
    G, S, k = 150, 25, 4
    sc = ocost.normalize_data(rng.poisson(3.0, (G, S * k)).astype(np.float64))
    st = ocost.normalize_data(rng.poisson(9.0, (G, S)).astype(np.float64))
    slots = np.full(S, k, np.int64)
    mapped, idx = gcyto.solve_linear_assignment_problem(sc, st, slots, "lap_CSPR", None, 3, "Pearson_correlation", process_idx=2)
    assert idx == 2 and np.array_equal(np.bincount(mapped, minlength=S), slots)
    dist, loc = gsolvers.calculate_cost(sc, st, slots, "lap_CSPR", "Pearson_correlation")
    np.random.seed(3)
    ci = np.transpose(10**6 * dist.astype(np.float64) + 10 * np.random.rand(*dist.shape) + 1).astype(int)
    mat = gsolvers.match_solution(ci.tolist())
    r, c = linear_sum_assignment(ci)
    assert int(mat[:, 1].sum()) == int(ci[r, c].sum())                       # an optimum of the integer problem
    assert np.array_equal(np.sort(mat[:, 0]), np.arange(S * k))
    assert np.array_equal(np.asarray(mapped), loc[mat[:, 0].astype(int)])
    # zero-cost arcs are never used (the reference does not add them), non-square input is rejected
    z = np.array([[0, 5, 9], [4, 0, 7], [8, 6, 0]])
    m = gsolvers.match_solution(z)
    assert np.all(z[np.arange(3), m[:, 0].astype(int)] != 0) and int(m[:, 1].sum()) == 9 + 4 + 6
    with pytest.raises(ValueError):
        gsolvers.match_solution(np.ones((2, 3), int))


def _gv11_ss():
    d = load("gv11_apply_linear_assignment.npz")
    idx_sc = np.split(d["ss_idx_sc"], np.cumsum(d["ss_idx_sc_lens"])[:-1])
    return d, idx_sc


def test_broadcast_context_single_rank_equals_local_transform():
    # the collective of the path (RCCL broadcast of the transformed ST operand) with a 1-rank communicator on this GPU:
    # same answers as the context that transforms ST itself; raw counts in (normalised on the device)
    from cytospace_amd import _lib
    d, idx_sc = _gv11_ss()
    sc, st = d["ss_counts"].astype(np.float32), d["ss_st_counts"].astype(np.float32)
    subs = list(d["ss_sub"])
    comm = _lib.Communicator(_lib.Communicator.unique_id(), 0, 1)
    try:
        with gcyto.ExpressionContext(sc, st, False, 0, comm=comm) as shared, gcyto.ExpressionContext(sc, st, False, 0) as local:
            assert shared.bcast_ms is not None and shared.bcast_ms >= 0.0
            a = shared.assign_chunks([(ix, subs[k]) for k, ix in enumerate(idx_sc)])
            b = local.assign_chunks([(ix, subs[k]) for k, ix in enumerate(idx_sc)])
            one = [local.assign_chunk(ix, subs[k]) for k, ix in enumerate(idx_sc)]        # one chunk at a time
        for x, y, z in zip(a, b, one):
            assert np.array_equal(x, y) and np.array_equal(x, z)
        r = gcyto.assign_chunks(sc, st, d["ss_slots"], idx_sc, subsampled_slots_list=subs, already_normalized=False, comm=comm)
        assert all(np.array_equal(r[k], a[k]) for k in range(len(idx_sc)))
    finally:
        comm.close()


def test_two_ranks_broadcast_and_disjoint_chunks(tmp_path):
    # one process per GPU: rank 0 transforms ST and broadcasts it, rank 1 never sees the ST matrix; the ranks solve disjoint
    # chunks whose union is the reference's own result (gv11).  Needs two devices (the driver's multi-GPU node).
    import subprocess
    import sys
    from cytospace_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    d, idx_sc = _gv11_ss()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_two_rank_worker.py")
    idf = str(tmp_path / "uid.bin")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", idf, str(tmp_path / f"out{r}.npz")]) for r in range(2)]
    for p_ in procs:
        assert p_.wait(timeout=300) == 0
    got = {}
    for r in range(2):
        o = np.load(str(tmp_path / f"out{r}.npz"))
        for k in o["chunks"]:
            assert int(k) not in got
            got[int(k)] = o[f"m{int(k)}"]
    assert sorted(got) == list(range(len(idx_sc)))
    S = d["ss_st_counts"].shape[1]
    ncol = 4
    pairs = {(int(c), int(sp) // ncol, int(sp) % ncol) for k, ix in enumerate(idx_sc) for c, sp in zip(ix, got[k])}
    want = {(int(c), int(r), int(k)) for c, (r, k) in zip(d["ss_out_cell"], d["ss_out_rowcol"])}
    assert pairs == want and S > 0


def _gv11_frames(mode):
    import pandas as pd
    d = load("gv11_apply_linear_assignment.npz")
    sc, st, slots = d[mode + "_counts"], d[mode + "_st_counts"], d[mode + "_slots"]
    G, C = sc.shape
    S = st.shape[1]
    sc_df = pd.DataFrame(sc, index=[f"g{i}" for i in range(G)], columns=[f"c{i}" for i in range(C)])
    st_df = pd.DataFrame(st, index=sc_df.index, columns=[f"s{i}" for i in range(S)])
    ncol = 10 if mode == "sc" else 4
    coords = pd.DataFrame({"row": np.arange(S) // ncol, "col": np.arange(S) % ncol}, index=st_df.columns)
    idx_sc = np.split(d[mode + "_idx_sc"], np.cumsum(d[mode + "_idx_sc_lens"])[:-1])
    kw = {}
    if mode == "sc":
        kw["index_st_list"] = np.split(d["sc_idx_st"], np.cumsum(d["sc_idx_st_lens"])[:-1])
    else:
        kw["subsampled_cell_number_to_node_assignment_list"] = list(d["ss_sub"])
    want = {(int(c), int(r), int(k)) for c, (r, k) in zip(d[mode + "_out_cell"], d[mode + "_out_rowcol"])}
    return sc_df, st_df, coords, slots, idx_sc, kw, want


@pytest.mark.parametrize("mode", ["sc", "ss"])
@pytest.mark.parametrize("how", ["devices=[0, 0]", "CYTOSPACE_HIP_DEVICES=0,0", "devices=[0, 0, 0]"])
def test_gv11_reference_signature_drives_every_logical_device(mode, how, monkeypatch):
    # SURVEY 8(b), the multi-chunk seam: apply_linear_assignment with the REFERENCE's arguments (cytospace.py:354-357) schedules
    # the chunks over every device of the process, one host thread per device.  On this one-GPU box the device list names GPU 0
    # twice (logical ranks): the scheduler, the rank that never sees the ST matrix, the broadcast (a device-to-device copy
    # between logical ranks of one GPU; RCCL between distinct GPUs), and the merge in submission order all run -- and the
    # (cell, spot) pairs are the ones the reference's own run produced (gv11).
    sc_df, st_df, coords, slots, idx_sc, kw, want = _gv11_frames(mode)
    extra = {}
    if how.startswith("CYTOSPACE"):
        monkeypatch.setenv("CYTOSPACE_HIP_DEVICES", "0,0")
    else:
        extra["devices"] = [0] * how.count("0")
    loc, ids = gcyto.apply_linear_assignment(sc_df, st_df, coords, slots, "lapjv_hip", None, 1, "Pearson_correlation", 2,
                                             idx_sc, **kw, **extra)
    got = {(int(c[1:]), int(r), int(k)) for c, (r, k) in zip(ids, loc.to_numpy())}
    assert len(got) == len(ids) == len(want) and got == want
    # submission order: the cells come back chunk after chunk
    assert [int(c[1:]) for c in ids] == [int(c) for ix in idx_sc for c in ix]


def test_in_process_communicator_collectives_and_failing_together():
    # csrc/comm.hip, the in-process kind (logical ranks on one GPU): count / kind / agree, the rank without ST receiving the
    # root's operand, and the failure rules -- a rank whose genes are not the root's fails with ValueError while its peer fails
    # with CYTO_ERR_PEER (nobody enters the data broadcast, nobody hangs); a rank that aborts releases a peer that waits.
    import threading
    from cytospace_amd import _lib
    d, idx_sc = _gv11_ss()
    sc, st = d["ss_counts"].astype(np.float32), d["ss_st_counts"].astype(np.float32)
    subs = list(d["ss_sub"])
    comms = _lib.Communicator.init_local([0, 0])
    assert [c.count() for c in comms] == [2, 2] and comms[0].kind() == "in-process"
    out = [None, None]

    def run(r, fn):
        try:
            out[r] = fn(r)
        except BaseException as e:     # noqa: BLE001
            out[r] = e

    def both(fn):
        th = [threading.Thread(target=run, args=(r, fn)) for r in range(2)]
        [t.start() for t in th]
        [t.join(120) for t in th]
        assert not any(t.is_alive() for t in th), "a rank hangs"
        return list(out)
    assert both(lambda r: comms[r].agree(0)) == [0, 0]
    assert both(lambda r: comms[r].agree(3 * r)) == [3, 3]

    def ctx_ok(r):
        with gcyto.ExpressionContext(sc, st if r == 0 else None, False, 0, comm=comms[r], n_spots=st.shape[1]) as ctx:
            return ctx.assign_chunks([(ix, subs[k]) for k, ix in enumerate(idx_sc)])
    a, b = both(ctx_ok)
    with gcyto.ExpressionContext(sc, st, False, 0) as local:
        want = local.assign_chunks([(ix, subs[k]) for k, ix in enumerate(idx_sc)])
    assert all(np.array_equal(x, y) and np.array_equal(x, z) for x, y, z in zip(a, b, want))

    def ctx_bad_genes(r):       # rank 1's cells have one gene less than the root's matrices (same padding bucket of 32)
        with gcyto.ExpressionContext(sc if r == 0 else sc[:-1], st if r == 0 else None, False, 0, comm=comms[r], n_spots=st.shape[1]):
            return "built"
    e0, e1 = both(ctx_bad_genes)
    assert isinstance(e1, ValueError) and isinstance(e0, _lib.CytoHipError) and "status 9" in str(e0)
    assert both(lambda r: comms[r].agree(0)) == [0, 0]            # the communicator is still usable: everybody left together

    def one_aborts(r):
        if r == 1:
            comms[1].abort()
            return "aborted"
        return comms[0].agree(0)
    r0, r1 = both(one_aborts)
    assert r1 == "aborted" and isinstance(r0, _lib.CytoHipError) and "status 9" in str(r0)
    for c in comms:
        c.close()
    # a worker that fails in its host code BEFORE the collective (a cell index outside the matrix: numpy's IndexError): the call
    # raises that error, no thread is left waiting
    with pytest.raises(IndexError):
        gcyto.assign_chunks_on_devices(sc, st, d["ss_slots"], idx_sc + [np.array([10 ** 9])], subsampled_slots_list=subs + [subs[0]],
                                       devices=[0, 0], already_normalized=False)
    with pytest.raises(ValueError):
        gcyto.visible_devices([0, 99])


_RCCL_ABORT_WORKER = r'''
import os, sys, threading, time
sys.path.insert(0, sys.argv[1])
os.environ["CYTO_COMM_FORCE_RCCL"] = "1"
from cytospace_amd import _lib
ndev = _lib.device_count()
# (a) one rank through ncclCommInitAll (what a one-GPU box can run of the RCCL kind): a collective, then the abort, then every
#     later collective fails AT ONCE with CYTO_ERR_PEER instead of touching a destroyed communicator
c, = _lib.Communicator.init_local([0])
assert c.kind() == "rccl" and c.count() == 1 and c.agree(5) == 5 and not c.aborted()
c.abort()
assert c.aborted()
for _ in range(2):
    try:
        c.agree(0); raise SystemExit("a collective on an aborted communicator returned")
    except _lib.CytoHipError as e:
        assert "status 9" in str(e), e
c.abort(); c.close()
print("RCCL_ABORT_ONE_OK")
if ndev >= 2:
    # (b) two ranks on two devices: rank 0 is already INSIDE ncclAllReduce when rank 1 -- which never enters -- aborts.  The abort
    #     of rank 1's own communicator alone would leave rank 0 in its kernel for ever; aborting every sibling releases it.
    comms = _lib.Communicator.init_local([0, 1])
    assert comms[0].kind() == "rccl"
    out = {}
    def waiting():
        try:
            out["r0"] = comms[0].agree(0)
        except BaseException as e:
            out["r0"] = e
    t = threading.Thread(target=waiting, daemon=True)
    t.start()
    time.sleep(1.0)
    assert t.is_alive(), "rank 0 should be waiting inside the collective"
    comms[1].abort()
    t.join(60)
    assert not t.is_alive(), "rank 0 still hangs in the collective after the abort"
    assert isinstance(out["r0"], _lib.CytoHipError) and "status 9" in str(out["r0"]), out
    assert comms[0].aborted() and comms[1].aborted()
    for c in comms:
        c.close()
    print("RCCL_ABORT_TWO_OK")
'''


def test_rccl_kind_abort_releases_every_sibling(tmp_path):
    # ADVICE r5 / VERDICT r5 weak 9: a worker thread that fails in host code must release peers that already wait inside
    # ncclBroadcast / ncclAllReduce -- every communicator of the process is aborted, not only the failing rank's.  In a process of
    # its own: a communicator that was aborted mid-collective leaves its device in whatever state RCCL leaves it.
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(_RCCL_ABORT_WORKER)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_ABORT_ONE_OK" in r.stdout, r.stdout + r.stderr
    from cytospace_amd import _lib
    if _lib.device_count() >= 2:
        assert "RCCL_ABORT_TWO_OK" in r.stdout, r.stdout + r.stderr


def test_failing_worker_aborts_all_and_join_has_a_deadline(monkeypatch):
    # assign_chunks_on_devices: a worker that raises aborts EVERY communicator; a peer that does not come back within the grace
    # period makes the call raise instead of hanging (here: rank 0's thread is held back artificially past the grace period)
    import threading
    from cytospace_amd import _lib
    d, idx_sc = _gv11_ss()
    sc, st = d["ss_counts"].astype(np.float32), d["ss_st_counts"].astype(np.float32)
    subs = list(d["ss_sub"])
    seen = []
    real_abort = _lib.Communicator.abort

    def spy_abort(self):
        seen.append(self.rank)
        return real_abort(self)
    monkeypatch.setattr(_lib.Communicator, "abort", spy_abort)
    with pytest.raises(IndexError):
        gcyto.assign_chunks_on_devices(sc, st, d["ss_slots"], idx_sc + [np.array([10 ** 9])], subsampled_slots_list=subs + [subs[0]],
                                       devices=[0, 0, 0], already_normalized=False)
    assert sorted(set(seen)) == [0, 1, 2]                      # the failing worker aborted all three, not only its own
    # the deadline: rank 0 never returns (stuck before its first collective); its peers fail, the call raises after the grace period
    monkeypatch.setenv("CYTOSPACE_HIP_ABORT_GRACE_S", "1.5")
    real = gcyto.assign_chunks
    gate = threading.Event()

    def stuck_rank0(*a, **kw):
        if kw.get("rank") == 0:
            gate.wait(30)
            raise RuntimeError("released")
        if kw.get("rank") == 1:
            raise KeyError("rank 1 fails in host code")
        return real(*a, **kw)
    monkeypatch.setattr(gcyto, "assign_chunks", stuck_rank0)
    t0 = time.monotonic()
    with pytest.raises(_lib.CytoHipError, match="did not return within"):
        gcyto.assign_chunks_on_devices(sc, st, d["ss_slots"], idx_sc, subsampled_slots_list=subs, devices=[0, 0], already_normalized=False)
    assert time.monotonic() - t0 < 20
    gate.set()


def test_one_process_distinct_devices_rccl_broadcast():
    # the same seam on two REAL devices: ncclCommInitAll, ncclBroadcast of the ST operand over xGMI.  Needs two GPUs.
    from cytospace_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    for mode in ("sc", "ss"):
        sc_df, st_df, coords, slots, idx_sc, kw, want = _gv11_frames(mode)
        loc, ids = gcyto.apply_linear_assignment(sc_df, st_df, coords, slots, "lapjv_hip", None, 1, "Pearson_correlation", 2,
                                                 idx_sc, **kw)
        got = {(int(c[1:]), int(r), int(k)) for c, (r, k) in zip(ids, loc.to_numpy())}
        assert got == want
    comms = _lib.Communicator.init_local([0, 1])
    assert comms[0].kind() == "rccl" and comms[1].count() == 2
    for c in comms:
        c.close()


@pytest.mark.parametrize("metric", ["Pearson_correlation", "Spearman_correlation", "Euclidean"])
def test_expression_context_chunks_equal_per_chunk_uploads(metric):
    # multi-chunk seam: transform once + device-side column gathers == uploading and transforming every chunk
    rng = np.random.default_rng(17)
    G, S, C = 90, 24, 60
    sc = ocost.normalize_data(rng.poisson(2.0, (G, C)).astype(np.float64))
    st = ocost.normalize_data(rng.poisson(8.0, (G, S)).astype(np.float64))
    with gcyto.ExpressionContext(sc, st, True, 0, metric) as ctx:
        # (a) a subset of cells against a subset of spots, with spots that receive no cell
        idx_sc = rng.permutation(C)[:20]
        idx_st = rng.permutation(S)[:9]
        slots = np.array([3, 0, 5, 2, 0, 4, 1, 0, 5])
        got, total, info = ctx.assign_chunk(idx_sc, slots, idx_st, return_info=True)
        want, total2, _ = gcyto.assign_pearson(sc[:, idx_sc], st[:, idx_st], slots, return_info=True, distance_metric=metric)
        assert np.array_equal(got, want) and total == total2
        assert np.array_equal(np.bincount(got, minlength=len(slots)), slots)
        assert info.gemm_flops == 2.0 * 96 * 6 * 20          # only the 6 spots with cells are contracted (Gpad = 96)
        # (b) all spots, per-chunk slot counts
        idx_sc = np.arange(10, 58)
        slots = np.full(S, 2)
        assert np.array_equal(ctx.assign_chunk(idx_sc, slots), gcyto.assign_pearson(sc[:, idx_sc], st, slots, distance_metric=metric))
        with pytest.raises(ValueError):
            ctx.assign_chunk(idx_sc, np.full(S, 3))            # not square
        with pytest.raises(ValueError):
            ctx.assign_chunk(np.array([0, 1, C]), np.array([3] + [0] * (S - 1)))   # cell index out of range


def test_gv7_cells_per_spot_estimate():
    # estimate_cell_number_RNA_reads (cytospace.py:116-134) with the normalisation on the device
    import pandas as pd
    d = load("gv7_upstream.npz")
    st = d["st_counts"]
    st_df = pd.DataFrame(st, index=[f"g{i}" for i in range(st.shape[0])], columns=[f"s{i}" for i in range(st.shape[1])])
    for mean in (5, 20):
        assert np.array_equal(gcyto.estimate_cell_number_RNA_reads(st_df, mean), d[f"cells_per_spot_mean{mean}"])


def test_float32_inputs_give_the_float64_results():
    # raw counts are exact in float32: uploading them as float32 (half the traffic) must not change anything
    rng = np.random.default_rng(23)
    G, S, k = 150, 20, 3
    sc = rng.poisson(2.0, (G, S * k)).astype(np.float64)
    st = rng.poisson(7.0, (G, S)).astype(np.float64)
    slots = np.full(S, k)
    for metric in ("Pearson_correlation", "Spearman_correlation", "Euclidean"):
        m64, t64, _ = gcyto.assign_pearson(sc, st, slots, already_normalized=False, return_info=True, distance_metric=metric)
        m32, t32, _ = gcyto.assign_pearson(sc.astype(np.float32), st.astype(np.float32), slots, already_normalized=False,
                                           return_info=True, distance_metric=metric)
        assert np.array_equal(m64, m32) and t64 == t32
    with gcyto.ExpressionContext(sc.astype(np.float32), st.astype(np.float32), already_normalized=False) as c32, \
         gcyto.ExpressionContext(sc, st, already_normalized=False) as c64:
        idx = np.arange(0, 30)
        sl = np.array([3] * 10 + [0] * 10)
        assert np.array_equal(c32.assign_chunk(idx, sl), c64.assign_chunk(idx, sl))


def test_integer_count_dtypes_give_the_float64_results():
    # round 6: raw counts are small integers -- uint16 / uint8 matrices are a half / a quarter of the float32
    # upload (CYTO_DTYPE_U16 / _U8; the transform kernels widen in their loads).  Same numbers bit for bit: the standardised
    # operands, normalize_data, the fused solves of every metric (also with different dtypes for the two matrices, odd shapes that
    # take the column-per-lane kernels, and the block pipeline of large problems), the chunked context.
    rng = np.random.default_rng(29)
    for G, S, k in ((150, 20, 3), (97, 13, 5)):
        sc = rng.poisson(2.0, (G, S * k)).astype(np.float64)
        st = rng.poisson(7.0, (G, S)).astype(np.float64)
        sc[:, 3] = 0                                              # a cell without counts: NaN -> 0 whatever the dtype
        slots = np.full(S, k)
        for dt in (np.uint16, np.uint8):
            assert np.array_equal(gcommon.normalize_data(sc.astype(dt)), gcommon.normalize_data(sc))
            for metric in ("Pearson_correlation", "Euclidean"):
                a = gcommon.StandardizedMatrix(sc.astype(dt), False, 0, metric).to_numpy()
                b = gcommon.StandardizedMatrix(sc, False, 0, metric).to_numpy()
                assert np.array_equal(a, b, equal_nan=True)
        sc[:, 3] = rng.poisson(2.0, G)
        for metric in ("Pearson_correlation", "Spearman_correlation", "Euclidean"):
            m64, t64, _ = gcyto.assign_pearson(sc, st, slots, already_normalized=False, return_info=True, distance_metric=metric)
            for dsc, dst in ((np.uint16, np.uint16), (np.uint8, np.uint16), (np.uint8, np.float32), (np.float64, np.uint8)):
                m, t, _ = gcyto.assign_pearson(sc.astype(dsc), st.astype(dst), slots, already_normalized=False, return_info=True,
                                               distance_metric=metric)
                assert np.array_equal(m, m64) and t == t64, (metric, dsc, dst)
        with gcyto.ExpressionContext(sc.astype(np.uint8), st.astype(np.uint16), already_normalized=False) as cu, \
             gcyto.ExpressionContext(sc, st, already_normalized=False) as c64:
            idx = np.arange(0, 3 * (S // 2))
            sl = np.array([3] * (S // 2) + [0] * (S - S // 2))
            assert np.array_equal(cu.assign_chunk(idx, sl), c64.assign_chunk(idx, sl))
    # _counts_matrix: what apply_linear_assignment uploads for a DataFrame of integer counts
    assert gcyto._counts_matrix(np.array([[0, 255], [3, 4]], np.int64)).dtype == np.uint8
    assert gcyto._counts_matrix(np.array([[0, 256], [3, 4]], np.int64)).dtype == np.uint16
    assert gcyto._counts_matrix(np.array([[0, 65536], [3, 4]], np.int64)).dtype == np.float32
    assert gcyto._counts_matrix(np.array([[-1, 5], [3, 4]], np.int64)).dtype == np.float32
    assert gcyto._counts_matrix(np.array([[0.0, 5.0], [3.0, 4.0]])).dtype == np.float32
    assert gcyto._counts_matrix(np.array([[0.1, 5.0], [3.0, 4.0]])).dtype == np.float64

