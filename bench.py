#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CytoSPACE linear-assignment hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: `python bench.py --gpus N` starts its N ranks ITSELF (one child process per GPU, LOCAL_RANK = device), or runs as one
rank of an external one-process-per-GPU launcher (`python -m <launcher> --nproc-per-node N ... bench.py --gpus N ...`: RANK /
LOCAL_RANK / WORLD_SIZE from the environment).  Either way the ranks meet through cytospace_amd.rendezvous.FileStore (a directory of small
files: the 128-byte RCCL id, the barriers around the timed region, the max over ranks) and through RCCL itself (the communicator of
the chunk legs; `n_gpus` is checked against ncclCommCount).  No tensor framework is imported anywhere.

A "step" is one full solve of the north star's problem: a 50 000 x 50 000 dense synthetic float32 cost matrix (SURVEY 8d
"uniform") already resident in HBM -> assignment (column reduction, row-cache build, the row-reduction phase -- reduction transfer,
then the eps-scaled Jacobi rounds on the whole chip --, the augmentation kernel).  It fits one GPU (10 GB), so by the measurement
contract it is the N = 1 workload (BASELINE.json's target sentence names it); configs[1] (20 000 x 20 000) is the extra leg "c2".
With N GPUs every rank solves its own copy of the instance (the reference shards independent sub-LAPs across workers:
cytospace.py:430-451; inside ONE LAP the path does not shard: replicas only), so scaling is "weak" and value = N * n / max-over-
ranks time.  Parity of the headline: indices == the committed uniqueness-certified golden (tests/golden/large_u50000.npz: classic
oracle == scipy), duals / rowsol == the wide restatement's golden, bit for bit, on every rank and every step's last result; plus the
in-run n = 3 000 gate against both oracles.

`roofline` is FLOOR-based: every cost entry must be read once, 4 n^2 bytes; achieved = 4 n^2 / ms_per_step, frac = achieved / 8 TB/s
(cannot exceed 1).  `traffic` = HBM bytes of the whole solve from the committed rocprofv3 --pmc passes; `latency_model` says what
really bounds the solve (dependent rounds x microseconds per round); `dominant_kernel` is the longest single launch with its
HIP-event time (compare with profiles/<tag>_kernel_stats*.csv).

Besides the headline the same JSON line carries, at N = 1, the other single-GPU workloads the north star names (--no-extras skips):
  "c2"        BASELINE configs[1]: the 20 000 x 20 000 uniform LAP against its goldens
  "c2_batch"  32 copies of that instance solved together
  "c2_cytolike" a 20 000 x 20 000 cost of the few-cell-type ("cytospace-like") generator, slots == 1, against its
              committed certified golden -- the instance class CytoSPACE's chunks belong to
  "c3"        configs[2] end to end: 20 000 genes x 50 000 cells x 5 000 spots, normalise + standardise, fp32-MFMA cost
              GEMM (its own "roofline" against the 157.3 TFLOP/s f32 matrix peak), LAP; wall time includes the H2D copies
  "c4_chunks" K concurrent 10 000-cell --sampling-sub-spots chunk LAPs (configs[3]'s unit of work) on one GPU, with the
              CPU oracle run on all host cores beside it (core count stated)
  "c5_chunks" configs[4]'s 50 single-cell-mode chunks (10 000 cells x 10 000 single-cell spots) in one batched call on one GPU
(every batched / chunked leg is timed on its SECOND pass -- a process that solves more than one batch takes its work buffers from the device
block cache; the first pass, which allocates them, is reported beside it as first_call_wall_s / first_pass_seconds_rank0)
and, at every N, "c4_strong" / "c4_sharded": configs[3]'s structure with the path's one collective -- rank 0 transforms the ST
matrix and broadcasts the operand over xGMI (RCCL), every rank uploads its own cells and solves its chunks in one batched call.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # same guide: dense fp32 matrix (v_mfma_f32_32x32x2_f32) peak


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def floor_roofline(problems, n, seconds, what):
    """The headline's floor-based roofline for a leg that solves `problems` LAPs of n rows: every entry of every cost matrix must be
    read once -- problems x 4 n^2 bytes -- over the leg's wall time, against the HBM peak (cannot exceed 1)."""
    b = 4.0 * n * n * problems
    gbs = b / seconds / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
            "floor_bytes": b, "definition": f"floor: {problems} x 4 n^2 bytes (n = {n}: {what}) read once, divided by the leg's wall time"}


def check_uniform_golden(n, r, wide, required):
    """A uniform-instance result against the committed goldens of that size: indices (tests/golden/large_u<n>.npz is
    uniqueness-certified: ANY exact solver's indices) and -- the wide solver -- rowsol and both duals bit for bit against the wide
    restatement's golden.  Returns (indices_ok, duals_ok) -- None where there is no golden; a missing golden of a `required` size
    and any mismatch end the run."""
    gpath = os.path.join(ROOT, "tests", "golden", f"large_u{n}.npz")
    if not os.path.exists(gpath):
        if required:
            raise SystemExit(f"bench.py: {gpath} is missing -- the parity of the headline size cannot be shown")
        return None, None
    d = np.load(gpath)
    ok = bool(np.array_equal(r["colsol"], d["colsol"]))
    duals = None
    wpath = os.path.join(ROOT, "tests", "golden", f"large_u{n}_wide.npz")
    if ok and wide:
        if os.path.exists(wpath):
            dw = np.load(wpath)
            duals = bool(_sha(r["u"]) == str(dw["u_sha256"]) and _sha(r["v"]) == str(dw["v_sha256"]) and _sha(r["rowsol"]) == str(dw["rowsol_sha256"]))
        elif required:
            raise SystemExit(f"bench.py: {wpath} is missing")
    if not ok or duals is False:
        raise SystemExit(f"uniform {n} x {n}: HIP result differs from tests/golden/large_u{n}[_wide].npz")
    return ok, duals


def extra_uniform(dev, n):
    """A uniform n x n LAP beside the headline (configs[1]'s 20 000 when the headline is the north star's 50 000, and the other way
    round), resident in HBM; the answer is compared with the committed goldens."""
    from cytospace_amd.lap import lap_solve
    from tools import instances
    t = time.perf_counter()
    buf = instances.blocks_to_device(instances.uniform_cost_blocks(n), n, dev)
    t_gen = time.perf_counter() - t
    lap_solve(None, np.float32, device_id=dev, device_ptr=buf.ptr, n=n, ld=n)          # warm-up
    walls, r = [], None
    for _ in range(3):
        t = time.perf_counter()
        r = lap_solve(None, np.float32, device_id=dev, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
        walls.append(time.perf_counter() - t)
    wall = float(np.median(walls))
    i = r["info"]
    out = {"workload": f"{n} x {n} dense uniform float32 cost resident in HBM" + (" (BASELINE.json configs[1])" if n == 20000 else ""),
           "n": n, "ms_per_solve": round(wall * 1e3, 2), "assignments_per_s": round(n / wall, 1),
           "kernel_ms": {"colred": round(i.ms_colred, 2), "row_caches": round(i.ms_cache, 2), "row_reduction": round(i.ms_arr, 2),
                         "wide_aug": round(i.ms_aug, 2)},
           "rounds": int(i.wide_rounds), "bids": int(i.scans_arr), "full_row_bids": int(i.wide_dense_arr), "searches": int(i.augmentations),
           "columns_settled": int(i.scans_aug_relax), "floor_4n2_frac": round(4.0 * n * n / wall / 1e9 / HBM_PEAK_GBS, 6),
           "instance_seconds": round(t_gen, 1)}
    ok, duals = check_uniform_golden(n, r, bool(i.wide), required=False)
    out["bit_exact_vs_oracle_golden"] = ok
    out["duals_bit_exact_vs_wide_oracle_golden"] = duals
    return out, buf, r


def extra_c2_cytolike(dev):
    """SURVEY 8(d)'s "cytospace-like" solver-only instance at configs[1]'s size: 20 000 spots x 20 000 cells of ten cell types
    (tools/instances.typed_unique_cost, every slot count 1), resident in HBM.  Indices against the committed golden (classic
    oracle, certified: duals on all n^2 entries, scipy, one-ulp perturbation), duals against the wide restatement's golden."""
    from cytospace_amd.lap import lap_solve
    from tools import instances
    n = 20000
    t = time.perf_counter()
    cost, _ = instances.typed_unique_cost(n, n, 20)
    t_gen = time.perf_counter() - t
    from cytospace_amd import _lib
    buf = _lib.DeviceBuffer.from_numpy(cost, dev)
    lap_solve(None, np.float32, device_id=dev, device_ptr=buf.ptr, n=n, ld=n)          # warm-up
    walls, r = [], None
    for _ in range(3):
        t = time.perf_counter()
        r = lap_solve(None, np.float32, device_id=dev, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
        walls.append(time.perf_counter() - t)
    buf.free()
    wall = float(np.median(walls))
    i = r["info"]
    out = {"workload": f"{n} x {n} few-cell-type cost (typed_unique_cost seed 20, slots == 1) resident in HBM",
           "ms_per_solve": round(wall * 1e3, 2), "assignments_per_s": round(n / wall, 1),
           "kernel_ms": {"colred": round(i.ms_colred, 2), "row_caches": round(i.ms_cache, 2), "row_reduction": round(i.ms_arr, 2),
                         "wide_aug": round(i.ms_aug, 2)},
           "rounds": int(i.wide_rounds), "phases": int(i.wide_phases), "bids": int(i.scans_arr), "full_row_bids": int(i.wide_dense_arr),
           "searches": int(i.augmentations), "columns_settled": int(i.scans_aug_relax),
           "row_scans_counted": int(i.row_scans), "hbm_rows_actually_read": int(i.hbm_row_reads),
           "floor_4n2_frac": round(4.0 * n * n / wall / 1e9 / HBM_PEAK_GBS, 6), "roofline": floor_roofline(1, n, wall, "the cost matrix"),
           "instance_seconds": round(t_gen, 1)}
    gpath = os.path.join(ROOT, "tests", "golden", "large_t20000.npz")
    if os.path.exists(gpath):
        d = np.load(gpath)
        ok = bool(np.array_equal(r["colsol"], d["colsol"]))
        out["cpu_oracle_seconds_in_the_build_container"] = round(float(d["oracle_seconds"]), 1)
        wpath = os.path.join(ROOT, "tests", "golden", "large_t20000_wide.npz")
        if ok and os.path.exists(wpath):
            dw = np.load(wpath)
            ok = bool(_sha(r["u"]) == str(dw["u_sha256"]) and _sha(r["v"]) == str(dw["v_sha256"]) and _sha(r["rowsol"]) == str(dw["rowsol_sha256"]))
            out["duals_bit_exact_vs_wide_oracle_golden"] = ok
        out["bit_exact_vs_oracle_golden"] = ok
        if not ok:
            raise SystemExit("c2_cytolike: HIP result differs from tests/golden/large_t20000[_wide].npz")
    else:
        raise SystemExit(f"c2_cytolike: {gpath} is missing")
    return out


def extra_c3(dev):
    """BASELINE configs[2]: 50k cells x 5k spots x 20k genes, fused on one GPU (raw float32 counts in, spots out)."""
    from cytospace_amd.cytospace import assign_pearson
    from tools import instances
    G, C, S = 20000, 50000, 5000
    t = time.perf_counter()
    sc, st, slots = instances.synth_expression(G, C, S, seed=1)
    t_gen = time.perf_counter() - t
    # Raw counts are small integers: they cross PCIe as the narrowest unsigned integer type that holds them (uint16 here: the largest
    # count of the instance is a few thousand), widened on the device -- what cytospace_amd.cytospace._counts_matrix hands over for a
    # DataFrame of integer counts.  The conversion belongs to the reader (untimed, like the generator); the float32 form is timed beside it.
    cmax = int(max(sc.max(), st.max()))
    cdt = np.uint8 if cmax < 256 else np.uint16 if cmax < 65536 else np.float32
    sc_n, st_n = np.ascontiguousarray(sc, dtype=cdt), np.ascontiguousarray(st, dtype=cdt)
    assign_pearson(sc, st, slots, already_normalized=False, device_id=dev)                # warm-up
    t = time.perf_counter()
    mapped32, total32, info32 = assign_pearson(sc, st, slots, already_normalized=False, device_id=dev, return_info=True)
    wall32 = time.perf_counter() - t
    assign_pearson(sc_n, st_n, slots, already_normalized=False, device_id=dev)            # warm-up
    t = time.perf_counter()
    mapped, total, info = assign_pearson(sc_n, st_n, slots, already_normalized=False, device_id=dev, return_info=True)
    wall = time.perf_counter() - t
    ok = bool(np.array_equal(np.bincount(mapped, minlength=S), slots))
    if not ok:
        raise SystemExit("c3: bincount(mapped) != slots")
    if not (np.array_equal(mapped, mapped32) and total == total32):
        raise SystemExit("c3: integer counts and float32 counts gave different results")
    tf = info.gemm_flops / (info.ms_gemm * 1e-3) / 1e12
    # K1 alone (normalise + standardise: colsum, moments, write) on counts already RESIDENT in HBM -- in the fused call above it
    # hides behind the PCIe upload.  SURVEY 8(d): bytes = (3 reads + 1 write) x 4 x G x columns
    from cytospace_amd import _lib
    k1 = None
    try:
        Cb = 16384
        xb = _lib.DeviceBuffer.from_numpy(np.ascontiguousarray(sc[:, :Cb]), dev)
        Gpad, ldz = -(-G // 32) * 32, -(-Cb // 128) * 128
        zb = _lib.DeviceBuffer(Gpad * ldz * 4, dev)
        L = _lib.lib()
        for rep in range(3):
            _lib.check(L.cyto_device_synchronize(dev))
            t = time.perf_counter()
            _lib.check(L.cyto_transform(0, G, Cb, xb.ptr, Cb, 0, 1, 0, zb.ptr, ldz, Gpad, dev, None))
            _lib.check(L.cyto_device_synchronize(dev))
            dt = time.perf_counter() - t
        xb.free(); zb.free()
        k1_bytes = 4.0 * 4 * G * Cb
        k1 = {"bound": "hbm", "kernel": "K1: colsum_partial_v + colmoments_partial_v + transform_write_v (+ finishers)", "columns": Cb,
              "ms": round(dt * 1e3, 3), "algorithmic_bytes": k1_bytes, "achieved": round(k1_bytes / dt / 1e9, 1), "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": round(k1_bytes / dt / 1e9 / HBM_PEAK_GBS, 4),
              "note": "wall time of cyto_transform on device-resident float32 counts between two device synchronisations"}
    except Exception as e:   # noqa: BLE001 (a diagnostic leg must not cost the line)
        k1 = {"error": f"{type(e).__name__}: {e}"}
    # the whole of c3 with the counts already RESIDENT in HBM (device matrices into the context, one chunk = the whole problem):
    # what the device does when the 4.4 GB upload at 57 GB/s does not hide it -- transforms, gathers, contraction, LAP
    resident = None
    try:
        import ctypes
        from cytospace_amd.cytospace import ExpressionContext
        dsc = _lib.DeviceBuffer.from_numpy(np.ascontiguousarray(sc), dev)
        dst = _lib.DeviceBuffer.from_numpy(np.ascontiguousarray(st), dev)
        L = _lib.lib()
        walls = []
        for rep in range(2):
            msc, mst = _lib.Matrix(), _lib.Matrix()
            msc.data, msc.ld, msc.is_f64, msc.on_device = dsc.ptr, C, 0, 1
            mst.data, mst.ld, mst.is_f64, mst.on_device = dst.ptr, S, 0, 1
            _lib.check(L.cyto_device_synchronize(dev))
            t = time.perf_counter()
            ctx = ExpressionContext.__new__(ExpressionContext)
            ctx.bcast_ms, ctx._h, ctx.G, ctx.C, ctx.S = None, ctypes.c_void_p(), G, C, S
            ms = ctypes.c_double()
            _lib.check(L.cyto_ctx_create_ex(0, G, ctypes.byref(msc), C, ctypes.byref(mst), S, 0, None, 0, 0, dev, ctypes.byref(ctx._h), ctypes.byref(ms)))
            t_ctx = time.perf_counter() - t
            m2, tot2, inf2 = ctx.assign_chunk(np.arange(C), slots, return_info=True)
            walls.append((time.perf_counter() - t, t_ctx, inf2))
            ctx.close()
        dsc.free(); dst.free()
        w, t_ctx, inf2 = min(walls, key=lambda x: x[0])
        resident = {"wall_ms": round(w * 1e3, 1), "assignments_per_s": round(C / w, 1), "transforms_ms": round(t_ctx * 1e3, 1),
                    "gather_ms": round(inf2.ms_standardize, 2), "gemm_ms": round(inf2.ms_gemm, 2), "lap_ms": round(inf2.lap.ms_total, 2),
                    "same_mapping_as_the_fused_call": bool(np.array_equal(m2, mapped)),
                    "note": "float32 counts uploaded beforehand (untimed); cyto_ctx_create_ex on device matrices + one chunk = the whole problem"}
    except Exception as e:   # noqa: BLE001
        resident = {"error": f"{type(e).__name__}: {e}"}
    # CPU beside it: the numpy float64 restatement of normalize_data + matrix_correlation_pearson
    # + the row gather (oracle/cost.py) on a bounded sample -- every spot against the first 2 000 cells
    cpu_cost = None
    try:
        from oracle import cost as ocost
        Cs = 2000
        try:
            from threadpoolctl import threadpool_info
            blas = [f"{p.get('internal_api')}:{p.get('num_threads')}" for p in threadpool_info()]
        except Exception:   # noqa: BLE001
            blas = ["unknown"]
        t = time.perf_counter()
        a = ocost.normalize_data(sc[:, :Cs].astype(np.float64))
        b = ocost.normalize_data(st.astype(np.float64))
        dr, _ = ocost.calculate_cost(a, b, slots, "lapjv", "Pearson_correlation")
        dt_cpu = time.perf_counter() - t
        flops = 2.0 * G * S * Cs
        cpu_cost = {"kind": "port", "sample": f"oracle/cost.py (numpy float64 restatement of common.py:142-147, 190-199 and the slot gather of "
                                              f"linear_assignment_solvers.py:63-66) on all {S} spots x the first {Cs} of the {C} cells",
                    "seconds": round(dt_cpu, 2), "cells_per_s": round(Cs / dt_cpu, 1), "GFLOPs_of_the_contraction_over_the_whole_time": round(flops / dt_cpu / 1e9, 1),
                    "blas_threads": blas, "usable_cores": _usable_cores(), "cost_rows_x_cols": list(dr.shape)}
        del a, b, dr
    except Exception as e:   # noqa: BLE001
        cpu_cost = {"error": f"{type(e).__name__}: {e}"}
    # the cells go up in blocks of 8192 and block b's contraction runs while block b + 1 is copied and transformed, so the
    # upload + transform time already contains all contractions but the last block's: the parts do not add up to the wall time
    return {"workload": f"{G} genes x {C} cells x {S} spots (10 slots each), raw counts as {np.dtype(cdt).name} -> spots",
            "wall_ms_incl_h2d": round(wall * 1e3, 1), "assignments_per_s_wall": round(C / wall, 1),
            "counts_dtype": np.dtype(cdt).name, "largest_count": cmax,
            "float32_counts": {"wall_ms_incl_h2d": round(wall32 * 1e3, 1), "upload_and_transform_ms": round(info32.ms_standardize, 1),
                               "same_mapping_and_total": True, "note": "the same call with the counts as float32 (rounds 1-5): twice the upload"},
            "kernel_ms": {"upload_and_transform_with_the_gemm_blocks_inside": round(info.ms_standardize, 1),
                          "pearson_gemm_blocks_sum": round(info.ms_gemm, 2), "lap": round(info.lap.ms_total, 1)},
            "roofline": {"bound": "mfma", "kernel": "pearson_gemm", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF,
                         "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4), "flops_per_launch": info.gemm_flops},
            "k1_roofline": k1, "counts_resident_in_hbm": resident, "cpu_baseline_cost_build": cpu_cost,
            "counts_density": round(float(np.count_nonzero(sc[:, :2000])) / (G * 2000), 3),
            "lap": {"ms": round(info.lap.ms_total, 2), "rounds": int(info.lap.wide_rounds), "scaled": bool(info.lap.wide_scaled),
                    "one_edge_searches": int(info.lap.wide_trivial)},
            "lap_row_scans": int(info.lap.row_scans), "bincount_equals_slots": ok, "instance_seconds": round(t_gen, 1)}


def extra_c2_batch(dev, cost_buf, n, B=32):
    """B independent instances of the headline workload solved together (one launch per chain phase, a workgroup per problem):
    what the chip delivers on configs[1]-sized problems when there is more than one of them (CytoSPACE's chunked modes)."""
    from cytospace_amd.lap import lap_solve_batch_device
    bufs = [cost_buf] + [cost_buf.clone() for _ in range(B - 1)]       # every chain reads its own copy
    lap_solve_batch_device([bufs[0].ptr], [n], device_id=dev, max_concurrent=1)
    walls = []
    for _ in range(2):          # the first call also allocates the work buffers of B problems (the device block cache is empty); the second is the steady state
        t = time.perf_counter()
        res = lap_solve_batch_device([b.ptr for b in bufs], [n] * B, device_id=dev, max_concurrent=B, return_info=True)
        walls.append(time.perf_counter() - t)
    wall = walls[-1]
    for b in bufs[1:]:
        b.free()
    same = all(np.array_equal(r["colsol"], res[0]["colsol"]) and np.array_equal(r["v"], res[0]["v"]) for r in res)
    if not same:
        raise SystemExit("c2_batch: copies of one instance solved together gave different answers")
    i = res[0]["info"]
    return {"workload": f"{B} copies of the headline instance ({n} x {n}) solved together, each on its own copy of the matrix",
            "wall_s": round(wall, 3), "first_call_wall_s": round(walls[0], 3), "assignments_per_s": round(B * n / wall, 1), "batch_kernel_ms": round(i.ms_total, 1),
            "row_reduction_ms": round(i.ms_arr, 1), "augmentation_ms": round(i.ms_aug, 1), "wide_solver": bool(i.wide), "copies_identical": same,
            "roofline": floor_roofline(B, n, wall, "the batch's cost matrices")}, res[0]


def _usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def extra_c4_chunks(dev, K, distinct=4, cpu_n=5000, cpu_threads=64):
    """K concurrent sub-spot chunk LAPs (10 000 cells each) on one GPU; beside it the CPU oracle on the host's cores
    (BASELINE.md section 3 item 2), on a BOUNDED sample: chunks of cpu_n cells (a 10 000-cell chunk takes the classic oracle
    155 s on one core -- tests/golden/large_c4s10000.npz records it --, so the default run times 5 000-cell chunks and puts the GPU's
    rate ON THE SAME 5 000-cell chunks beside it: `gpu_on_the_same_sample`; assignments/s falls with n on both sides)."""
    from concurrent.futures import ThreadPoolExecutor
    from cytospace_amd import _lib
    from cytospace_amd.lap import lap_solve_batch_device
    from oracle.jv import jv_oracle, jv_oracle_wide
    from tools import instances
    n = 10000
    t = time.perf_counter()
    costs = [instances.c4_chunk_cost(n, seed=4 + k)[0] for k in range(distinct)]
    small = [instances.c4_chunk_cost(cpu_n, seed=40 + k)[0] for k in range(distinct)]
    t_gen = time.perf_counter() - t
    bufs = [_lib.DeviceBuffer.from_numpy(costs[k], dev) for k in range(distinct)]
    bufs += [bufs[k % distinct].clone() for k in range(distinct, K)]          # every chain reads its OWN copy (no shared cache lines)
    lap_solve_batch_device([b.ptr for b in bufs[:1]], [n], device_id=dev, max_concurrent=1)     # warm-up (block cache, code objects)
    t = time.perf_counter()
    one = lap_solve_batch_device([bufs[0].ptr], [n], device_id=dev, max_concurrent=1, return_info=True)[0]
    wall1 = time.perf_counter() - t
    walls = []
    for _ in range(2):          # (first call: the work buffers of K problems are allocated -- the block cache is empty; second: the steady state)
        t = time.perf_counter()
        res = lap_solve_batch_device([b.ptr for b in bufs], [n] * K, device_id=dev, max_concurrent=K, return_info=True)
        walls.append(time.perf_counter() - t)
    wall = walls[-1]
    for b in bufs:
        b.free()
    exact = None
    gpath = os.path.join(ROOT, "tests", "golden", "large_c4s10000.npz")
    if os.path.exists(gpath):            # instance 0 (seed 4) is the golden instance: every copy of it must match the oracle's answer
        # spot rows are duplicated: the spot of every cell is what any exact solver must return (the golden is certified for that);
        # the wide restatement's own golden pins the slots and the duals bit for bit
        d = np.load(gpath)
        loc0 = instances.c4_chunk_cost(n, seed=4)[1]
        exact = all(np.array_equal(loc0[res[k]["colsol"]], loc0[d["colsol"]]) for k in range(0, K, distinct))
        wpath = os.path.join(ROOT, "tests", "golden", "large_c4s10000_wide.npz")
        if exact and res[0]["info"].wide and os.path.exists(wpath):
            dw = np.load(wpath)
            exact = all(np.array_equal(res[k]["colsol"], dw["colsol"]) and _sha(res[k]["v"]) == str(dw["v_sha256"])
                        and _sha(res[k]["u"]) == str(dw["u_sha256"]) for k in range(0, K, distinct))
        if not exact:
            raise SystemExit("c4_chunks: HIP result differs from tests/golden/large_c4s10000[_wide].npz")
    same = all(np.array_equal(res[k]["colsol"], res[k % distinct]["colsol"]) for k in range(K))
    if not same:
        raise SystemExit("c4_chunks: copies of one instance solved concurrently gave different answers")
    # ---- CPU: the oracle, one thread alone, then T threads at once (ctypes releases the GIL), one chunk each ----
    cores = _usable_cores()
    T = max(1, min(cores, cpu_threads))
    t = time.perf_counter()
    o1 = jv_oracle(small[0], np.float32)
    cpu1 = time.perf_counter() - t
    t = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        ora = list(ex.map(lambda k: jv_oracle(small[k % distinct], np.float32), range(T)))
    cpu_wall = time.perf_counter() - t
    bs = [_lib.DeviceBuffer.from_numpy(small[k], dev) for k in range(distinct)]
    rs = lap_solve_batch_device([b.ptr for b in bs], [cpu_n] * distinct, device_id=dev, max_concurrent=distinct, return_info=True)
    # the GPU on the CPU's own sample (like for like): T chunks of cpu_n cells in one batched call, and one alone
    bs += [bs[k % distinct].clone() for k in range(distinct, T)]
    t = time.perf_counter()
    lap_solve_batch_device([b.ptr for b in bs[:T]], [cpu_n] * T, device_id=dev, max_concurrent=T)
    gpu_same_wall = time.perf_counter() - t
    t = time.perf_counter()
    lap_solve_batch_device([bs[0].ptr], [cpu_n], device_id=dev, max_concurrent=1)
    gpu_same_one = time.perf_counter() - t
    for b in bs:
        b.free()
    with ThreadPoolExecutor(T) as ex:      # the batch runs the wide solver: its restatement on the same instances (not timed)
        oraw = list(ex.map(lambda k: jv_oracle_wide(small[k], np.float32), range(min(distinct, T))))
    small_exact = all(np.array_equal(rs[k]["colsol"], oraw[k]["colsol"]) and np.array_equal(rs[k]["v"], oraw[k]["v"])
                      and abs(oraw[k]["total"] - ora[k]["total"]) <= 1e-5 * max(1.0, abs(ora[k]["total"]))
                      for k in range(min(distinct, T)))
    if not small_exact or not np.array_equal(o1["colsol"], ora[0]["colsol"]):
        raise SystemExit("c4_chunks: HIP result differs from the CPU oracle on the CPU sample")
    i0 = one["info"]
    return {"workload": f"{K} concurrent {n} x {n} sub-spot chunk LAPs ({distinct} distinct seeded instances, every chain on its own copy), "
                        "cost resident in HBM, ONE launch per solver phase with a workgroup per chunk",
            "chunks": K, "wall_s": round(wall, 3), "first_call_wall_s": round(walls[0], 3), "assignments_per_s": round(K * n / wall, 1),
            "roofline": floor_roofline(K, n, wall, "every chunk's cost matrix, resident in HBM"),
            "full_row_bids_chunk0": int(res[0]["info"].wide_dense_arr), "bids_chunk0": int(res[0]["info"].scans_arr),
            "rounds_chunk0": int(res[0]["info"].wide_rounds),
            # what the 256 chains together draw from HBM: every full-row scan reads its 4 n bytes (the cached scans read none)
            "hbm_rows_read": int(sum(r["info"].hbm_row_reads for r in res)),
            "hbm_GBs_rows_read": round(sum(r["info"].hbm_row_reads for r in res) * 4.0 * n / wall / 1e9, 1),
            "one_chunk_alone": {"wall_s": round(wall1, 4), "assignments_per_s": round(n / wall1, 1), "kernel_ms": round(i0.ms_total, 1),
                                "roofline": floor_roofline(1, n, wall1, "the chunk's cost matrix"),
                                "row_reduction_ms": round(i0.ms_arr, 1), "augmentation_ms": round(i0.ms_aug, 1), "wide_solver": bool(i0.wide),
                                "row_scans": int(i0.row_scans), "aug_scans": int(i0.scans_aug_relax),
                                "aug_full_row_scans": int(i0.wide_dense_aug if i0.wide else i0.aug_dense_scans),
                                "us_per_aug_scan": round(i0.ms_aug * 1e3 / max(1, i0.scans_aug_relax), 3)},
            "bit_exact_vs_oracle_golden": exact, "copies_identical": same,
            "cpu_baseline": {"value": round(T * cpu_n / cpu_wall, 1), "unit": "assignments/s", "cores": T, "kind": "port",
                             "usable_cores": cores, "host_cores": os.cpu_count(),
                             "one_thread_alone": round(cpu_n / cpu1, 1),
                             "gpu_on_the_same_sample": {"assignments_per_s": round(T * cpu_n / gpu_same_wall, 1), "chunks": T, "wall_s": round(gpu_same_wall, 3),
                                                        "one_chunk_alone_assignments_per_s": round(cpu_n / gpu_same_one, 1)},
                             "ten_thousand_cell_chunk_one_thread_seconds_in_the_build_container": 154.6,
                             "sample": f"oracle/jv_oracle.c on sub-spot chunks of {cpu_n} cells (same generator): one thread alone {cpu1:.1f} s; "
                                       f"{T} threads at once, one chunk each, {cpu_wall:.1f} s; the HIP path solves the same {distinct} "
                                       f"instances bit-identically ({rs[0]['info'].ms_total:.0f} ms of kernels each)",
                             "cpu_model": _cpu_model()},
            "instance_seconds": round(t_gen, 1)}


def extra_c5_chunks(dev, K=50, G=500, chunk=10000, sets=8, cpu_n=10000, cpu_threads=64):
    """BASELINE configs[4]'s LAP work on ONE GPU: --single-cell mode partitions 500 000 cells x 500 000 single-cell spots into
    50 chunks of <= 10 000 cells against <= 10 000 spots (cytospace.py:441-451; every slot count 1, no duplicated rows); here
    all K chunks go through ONE batched context call (gather + fp32-MFMA cost build per chunk, then every LAP together).
    (Synthetic, Xenium-like: a G-gene panel; `sets` x 10 000 cells and spots are drawn, chunk k pairs cell set k % sets with spot
    set k // sets.)  CPU beside it: the oracle's LAP on the cost matrices of cpu_n-cell chunks (cost built by the GPU: numpy
    needs 8 s per chunk for it), one chunk per thread."""
    from concurrent.futures import ThreadPoolExecutor
    from cytospace_amd import common
    from cytospace_amd.cytospace import ExpressionContext
    from cytospace_amd.lap import lap_solve
    from oracle.jv import jv_oracle, jv_oracle_wide
    from tools import instances
    t = time.perf_counter()
    sc, st = instances.single_cell_expression(G, sets * chunk, sets * chunk, seed=5)
    t_gen = time.perf_counter() - t
    from cytospace_amd.cytospace import _counts_matrix
    if float(max(sc.max(), st.max())) < 65536:                     # (integer counts as the reader hands them over: see extra_c4_sharded)
        sc, st = _counts_matrix(sc.astype(np.int64)), _counts_matrix(st.astype(np.int64))
    ones = np.ones(chunk, np.int64)
    work = [(np.arange((k % sets) * chunk, (k % sets + 1) * chunk), ones,
             np.arange(((k // sets) % sets) * chunk, ((k // sets) % sets + 1) * chunk)) for k in range(K)]
    t0 = time.perf_counter()
    with ExpressionContext(sc, st, already_normalized=False, device_id=dev) as ctx:
        ctx.assign_chunks(work[:1], max_concurrent=1)                                      # warm-up
        t1 = time.perf_counter()
        walls = []
        for _ in range(2):      # (first call: cold block cache; second: the steady state)
            t2 = time.perf_counter()
            res = ctx.assign_chunks(work, max_concurrent=K, return_info=True)
            walls.append(time.perf_counter() - t2)
        wall = walls[-1]
    if not all(np.array_equal(np.sort(m), np.arange(chunk)) for m, _, _ in res):
        raise SystemExit("c5_chunks: a chunk's mapping is not a permutation of its spots")
    # ---- CPU sample: cpu_n-cell chunks, LAP only, same cost matrices on both sides (bit-exact comparison) ----
    distinct = 2
    costs = []
    for k in range(distinct):
        cb, N, ld, _ = common.pearson_cost_device(sc[:, k * cpu_n:(k + 1) * cpu_n], st[:, k * cpu_n:(k + 1) * cpu_n],
                                                  np.ones(cpu_n, np.int64), dev, already_normalized=False)
        costs.append(np.ascontiguousarray(cb.to_numpy((N, ld), np.float32)[:, :cpu_n]))
        cb.free()
    cores = _usable_cores()
    T = max(1, min(cores, cpu_threads))
    t = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        ora = list(ex.map(lambda k: jv_oracle(costs[k % distinct], np.float32), range(T)))
    cpu_wall = time.perf_counter() - t
    gp = [lap_solve(costs[k], np.float32, device_id=dev, return_info=True, opts=dict(mode=1)) for k in range(distinct)]   # the chain solver
    gw = [lap_solve(costs[k], np.float32, device_id=dev, return_info=True) for k in range(distinct)]                      # the wide solver (the batch's)
    oraw = [jv_oracle_wide(costs[k], np.float32) for k in range(distinct)]
    if not all(np.array_equal(gp[k]["colsol"], ora[k]["colsol"]) and np.array_equal(gp[k]["v"], ora[k]["v"])
               and np.array_equal(gw[k]["colsol"], oraw[k]["colsol"]) and np.array_equal(gw[k]["v"], oraw[k]["v"])
               and abs(oraw[k]["total"] - ora[k]["total"]) <= 1e-5 * max(1.0, abs(ora[k]["total"])) for k in range(min(distinct, T))):
        raise SystemExit("c5_chunks: HIP result differs from the CPU oracle on the CPU sample")
    i0 = res[0][2]
    return {"workload": f"{K} single-cell-mode chunks ({chunk} cells x {chunk} single-cell spots, {G}-gene panel; configs[4] = 50 such chunks) "
                        "through one batched context call on one GPU: per-chunk gather + MFMA cost build, then every LAP together",
            "chunks": K, "wall_s": round(wall, 3), "first_call_wall_s": round(walls[0], 3), "assignments_per_s": round(K * chunk / wall, 1),
            "roofline": floor_roofline(K, chunk, wall, "every chunk's cost matrix as the cost build leaves it in HBM; the wall time includes the cost builds"),
            "context_s": round(t1 - t0, 2), "counts_dtype": sc.dtype.name, "cost_build_ms_total": round(sum(r[2].ms_standardize + r[2].ms_gemm for r in res), 1),
            "chunk0": {"lap_batch_kernel_ms": round(i0.lap.ms_total, 1), "row_reduction_ms": round(i0.lap.ms_arr, 1), "wide_solver": bool(i0.lap.wide),
                       "augmentation_ms": round(i0.lap.ms_aug, 1), "row_scans": int(i0.lap.row_scans),
                       "aug_scans": int(i0.lap.scans_aug_relax), "searches": int(i0.lap.augmentations)},
            "cpu_baseline": {"value": round(T * cpu_n / cpu_wall, 1), "unit": "assignments/s", "cores": T, "kind": "port",
                             "usable_cores": cores,
                             "sample": f"oracle/jv_oracle.c (LAP only) on the cost matrices of {cpu_n}-cell single-cell chunks, {T} threads at once, "
                                       f"one chunk each, {cpu_wall:.1f} s; the HIP solver gives bit-identical results on the same matrices "
                                       f"({gp[0]['info'].ms_total:.0f} ms of kernels each)"},
            "instance_seconds": round(t_gen, 1)}


def extra_c4_sharded(dev, rank, world, store, comm, chunks_per_rank, G=5000, S=50000, chunk=10000, cell_sets=8, total_chunks=0):
    """BASELINE configs[3]'s structure on N GPUs (weak scaling: chunks_per_rank chunks of 10 000 cells per GPU against
    50 000 spots): rank 0 transforms the ST matrix and broadcasts the float32 operand over xGMI (RCCL; the one collective of
    the path), every rank uploads only its own cells as raw counts and solves its chunks in one batched call.
    (Synthetic: a rank draws cell_sets x 10 000 cells; chunk k uses cell set k % cell_sets with its own sub-spot slot draw,
    so 64 chunks need the expression of 80 000 cells, not 640 000.)"""
    from cytospace_amd import _lib
    from cytospace_amd.cytospace import ExpressionContext
    # total_chunks > 0: STRONG scaling -- the fixed problem of configs[3] (200 000 cells = 20 chunks of 10 000), chunk k on rank
    # k % world (equal sizes: what the LPT schedule gives), every chunk with its own seeded cells whatever the world size
    strong = total_chunks > 0
    mine = list(range(rank, total_chunks, world)) if strong else list(range(chunks_per_rank))
    if strong:
        chunks_per_rank, cell_sets = len(mine), max(1, len(mine))
    K = 10
    g = np.random.default_rng(5)                                   # shared by all ranks: gene means, type multipliers
    m = g.lognormal(0.0, 1.5, G).astype(np.float32)
    mult = g.lognormal(0.0, 0.75, (K, G)).astype(np.float32)
    t = time.perf_counter()
    r = np.random.default_rng(1000 + rank)
    cell_sets = max(1, min(cell_sets, chunks_per_rank))
    C = cell_sets * chunk
    sc = np.empty((G, C), np.float32)
    for lo in range(0, C, 5000):
        if strong and mine:                                        # cell set q of this rank = the cells of global chunk mine[q]
            r = np.random.default_rng(7000 + 2 * mine[lo // chunk] + (lo % chunk) // 5000)
        ty = r.integers(0, K, 5000)
        sc[:, lo:lo + 5000] = r.poisson(0.3 * m[:, None] * mult[ty].T)
    st = None
    if rank == 0:                                                  # a spot: four cells' worth of counts (only the root holds ST)
        st = np.empty((G, S), np.float32)
        for lo in range(0, S, 5000):
            ty = g.integers(0, K, (4, 5000))
            st[:, lo:lo + 5000] = g.poisson(0.3 * m[:, None] * (mult[ty[0]] + mult[ty[1]] + mult[ty[2]] + mult[ty[3]]).T)
    if strong:
        subs = [np.bincount(np.random.default_rng(9000 + k).integers(0, S, chunk), minlength=S) for k in mine]
    else:
        subs = [np.bincount(r.integers(0, S, chunk), minlength=S) for _ in range(chunks_per_rank)]    # sub-spot slot counts
    # raw counts cross PCIe as the narrowest unsigned integer type that holds them (what cytospace_amd.cytospace._counts_matrix hands
    # apply_linear_assignment for a table of integer counts; untimed like the generator: it is the reader's job)
    from cytospace_amd.cytospace import _counts_matrix
    sc = _counts_matrix(sc.astype(np.int64) if sc.dtype.kind == "f" and float(sc.max()) < 65536 else sc)
    if st is not None:
        st = _counts_matrix(st.astype(np.int64) if st.dtype.kind == "f" and float(st.max()) < 65536 else st)
    t_gen = time.perf_counter() - t
    def sync():
        _lib.check(_lib.lib().cyto_device_synchronize(dev))
        if store is not None:
            store.barrier()

    els = []
    for _ in range(2):          # the whole leg twice: the first pass allocates every work buffer (cold block cache), the second is the steady state
        sync()
        t0 = time.perf_counter()
        with ExpressionContext(sc, st, False, dev, comm=comm, n_spots=S) as ctx:
            t1 = time.perf_counter()
            res = ctx.assign_chunks([(np.arange((k % cell_sets) * chunk, (k % cell_sets + 1) * chunk), subs[k])
                                     for k in range(chunks_per_rank)], max_concurrent=chunks_per_rank, return_info=True)
            bcast_ms = ctx.bcast_ms
        sync()
        els.append(time.perf_counter() - t0)
    el = els[-1]
    ok = all(np.array_equal(np.bincount(mp, minlength=S), subs[k]) for k, (mp, _, _) in enumerate(res))
    if not ok:
        raise SystemExit("c4_sharded: bincount(mapped) != the chunk's slot counts")
    if store is not None:
        el = float(store.allreduce_max(el))
    first_el = els[0]
    i0 = res[0][2] if res else None
    if strong:
        return {"workload": f"configs[3] as configured: {total_chunks * chunk} cells = {total_chunks} sub-spot chunks of {chunk} cells against {S} spots, "
                            f"{G} genes, on {world} GPU(s): chunk k on rank k % {world}; ST transformed on rank 0 + RCCL broadcast",
                "assignments_per_s": round(total_chunks * chunk / el, 1), "seconds": round(el, 3), "first_pass_seconds_rank0": round(first_el, 3), "scaling": "strong",
                "roofline": floor_roofline(total_chunks, chunk, el, "every chunk's LAP as the reference materialises it; the time includes upload, transforms, broadcast and cost builds"),
                "chunks_on_rank0": len(mine), "counts_dtype": sc.dtype.name, "bcast_ms_rank0": None if bcast_ms is None else round(bcast_ms, 3),
                "rank0_longest_lap_kernel_ms": round(max(r_[2].lap.ms_total for r_ in res), 1) if res else None,
                "bincount_equals_slots": ok, "instance_seconds_rank0": round(t_gen, 1)}
    return {"workload": f"{world} GPU(s) x {chunks_per_rank} sub-spot chunks of {chunk} cells against {S} spots, {G} genes: "
                        "ST transformed on rank 0 + RCCL broadcast, per-rank raw-count upload, batched chunk solves",
            "assignments_per_s": round(world * chunks_per_rank * chunk / el, 1), "seconds": round(el, 3), "first_pass_seconds_rank0": round(first_el, 3), "scaling": "weak",
            "roofline": floor_roofline(world * chunks_per_rank, chunk, el, "every chunk's LAP as the reference materialises it; the time includes upload, transforms, broadcast and cost builds; peak = ONE GPU's: divide by the GPUs for a per-device fraction"),
            "cost_build_ms_per_rank": round(sum(r[2].ms_standardize + r[2].ms_gemm for r in res), 1),
            "context_s_rank0": round(t1 - t0, 3), "counts_dtype": sc.dtype.name, "bcast_ms_rank0": None if bcast_ms is None else round(bcast_ms, 3),
            "bcast_bytes": int(-(-G // 32) * 32) * int(-(-S // 128) * 128) * 4,
            "chunk0": {"gather_ms": round(i0.ms_standardize, 2), "gemm_ms": round(i0.ms_gemm, 2), "spots_with_cells": int((subs[0] > 0).sum()),
                       "lap_batch_kernel_ms": round(i0.lap.ms_total, 1), "lap_row_scans": int(i0.lap.row_scans)},
            "bincount_equals_slots": ok, "instance_seconds_rank0": round(t_gen, 1)}


class ThreadStore:
    """barrier / allgather / max for LOGICAL ranks that are threads of one process (--oversubscribe: more ranks than devices -- RCCL
    refuses duplicate devices, so the multi-rank chunk legs run on the in-process communicator, cyto_comm_init_local, one thread per rank)"""

    def __init__(self, world):
        import threading
        self.world, self._bar, self._vals = world, threading.Barrier(world), [None] * world

    def view(self, rank):
        return _ThreadStoreView(self, rank)


class _ThreadStoreView:
    def __init__(self, s, rank):
        self.s, self.rank, self.world = s, rank, s.world

    def barrier(self):
        self.s._bar.wait()

    def allgather(self, x):
        self.s._vals[self.rank] = x
        self.s._bar.wait()
        out = list(self.s._vals)
        self.s._bar.wait()
        return out

    def allreduce_max(self, x):
        return max(self.allgather(x))


def make_cost(n, seed):
    # draw in float64, then cast (float32 draws live on a 2^-24 grid)
    return np.random.default_rng(seed).random((n, n)).astype(np.float32)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one child process per GPU, LOCAL_RANK = device;
    rank 0 inherits stdout and prints the line), hand them a rendezvous directory, wait.  A failed or stuck rank ends the job."""
    import shutil
    import subprocess
    import tempfile
    from cytospace_amd import _lib
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.gpus > ndev and not args.oversubscribe:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {ndev} HIP device(s) "
                         "(--oversubscribe places several ranks on one device: a harness self-test, not a measurement)")
    rdv = tempfile.mkdtemp(prefix="cytohip_rdv_")
    procs = []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), CYTO_RDV_DIR=rdv, CYTO_BENCH_SPAWNED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=None if r == 0 else sys.stderr.fileno()))
        deadline = time.monotonic() + args.rank_timeout
        rcs = [None] * len(procs)
        while any(rc is None for rc in rcs):
            for k, pr in enumerate(procs):
                if rcs[k] is None:
                    rcs[k] = pr.poll()
            if any(rc not in (None, 0) for rc in rcs) or time.monotonic() > deadline:
                time.sleep(2.0)                             # (the others may be on their way out for the same reason)
                for k, pr in enumerate(procs):
                    if pr.poll() is None:
                        pr.kill()                           # exactly the children started above, by handle
                    rcs[k] = pr.wait()
                why = "timed out" if all(rc in (0, -9) for rc in rcs) else "a rank failed"
                raise SystemExit(f"bench.py --gpus {args.gpus}: {why} (exit codes {rcs})")
            time.sleep(0.05)
    finally:
        shutil.rmtree(rdv, ignore_errors=True)
    return 0


def whole_solve_traffic(tag, n):
    """HBM bytes of ONE whole solve from the committed rocprofv3 --pmc passes (tools/prof_round.sh -> tools/pmc_to_json.py:
    per-kernel per-dispatch averages and dispatch counts of a run of `solves` solves; read = 2 x FETCH_SIZE, the gfx950 correction,
    calibrated on colred_partial which reads the matrix exactly once)."""
    src = f"profiles/{tag}_pmc_traffic_n{n}.json"
    try:
        pm = json.load(open(os.path.join(ROOT, src)))
        if pm.get("n") != n:
            return None, None, None
        ks = pm["kernels"]
        solves = max(1, int(ks.get("colred_partial<float>", {}).get("dispatches", 1)))
        tot = sum((v.get("hbm_read_bytes", 0) + v.get("hbm_write_bytes_uncalibrated", 0)) * v.get("dispatches", 0) for v in ks.values()) / solves
        return int(tot), src, ks
    except (OSError, ValueError, KeyError):
        return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=50000, help="LAP size (default: the north star's 50 000; BASELINE.json configs[1] is 20000)")
    ap.add_argument("--cpu-n", type=int, default=0,
                    help="size of the bounded CPU-baseline sample (0: min(n, 20000): ~16 s of one core; when it equals n the very same "
                         "instance is used and the GPU result is compared bit for bit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (no c2 / c3 / c4_chunks / c5_chunks ...)")
    ap.add_argument("--c4-chunks", type=int, default=256, help="concurrent chunk LAPs in the c4_chunks leg (256 = one chain per CU; 400 MB each)")
    ap.add_argument("--c4-rank-chunks", type=int, default=64, help="chunks per rank in the c4_sharded leg")
    ap.add_argument("--sharded-timeout", type=float, default=600.0, help="watchdog of the c4_sharded leg, seconds")
    ap.add_argument("--rank-timeout", type=float, default=3000.0, help="self-spawned ranks (--gpus N without a launcher) are ended after this many seconds")
    ap.add_argument("--oversubscribe", action="store_true", help="allow more ranks than devices (rank r on device r %% devices; no RCCL "
                                                                 "communicator: a self-test of the harness on a one-GPU box)")
    ap.add_argument("--pmc-tag", default="r06", help="profiles/<tag>_pmc_traffic_n<n>.json supplies roofline.traffic")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    # ONE JSON line on stdout: everything else this process (or a library under it: RCCL prints a version banner through C
    # stdio) writes to file descriptor 1 goes to stderr instead; the line itself is written to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    from cytospace_amd import _lib
    from cytospace_amd.lap import lap_solve
    from cytospace_amd.rendezvous import FileStore
    from oracle.jv import jv_oracle, jv_oracle_wide   # checker + cpu_baseline leg only

    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if world > ndev and not args.oversubscribe:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} HIP device(s) visible")
    dev = local_rank % ndev
    n = args.n
    store = FileStore.from_env() if world > 1 else None

    # ---- the ranks' communicator (RCCL): made FIRST, under a watchdog, so that n_gpus is what RCCL itself counted ----
    comm, rccl = None, None
    if world > 1 and world <= ndev:
        import threading
        uid = store.bcast(_lib.Communicator.unique_id() if rank == 0 else None)
        cbox = {}

        def _mk():
            try:
                c = _lib.Communicator(uid, rank, world, device_id=dev)
                cbox["count"] = c.count()
                cbox["comm"] = c
            except BaseException as e:   # noqa: BLE001
                cbox["error"] = f"{type(e).__name__}: {e}"
        th = threading.Thread(target=_mk, daemon=True)
        th.start()
        th.join(300.0)
        states = store.allgather(cbox.get("count", cbox.get("error", "stuck")))
        if all(x == world for x in states):
            comm, rccl = cbox["comm"], {"ranks_counted_by_rccl": world, "kind": cbox["comm"].kind()}
        else:
            raise SystemExit(f"bench.py: the RCCL communicator of {world} ranks could not be made: {states}")
    elif world > 1:
        rccl = {"ranks_counted_by_rccl": None, "note": f"oversubscribed: {world} ranks on {ndev} device(s), RCCL refuses duplicate devices"}

    def barrier():
        _lib.check(_lib.lib().cyto_device_synchronize(dev))
        if store is not None:
            store.barrier()

    # ---- parity gate on a size the oracles finish in a second (bit-exact, incl. duals): the wide solver (what every float32 leg
    # of this file runs: single solves, batches, chunks) against its restatement, the chain solver (cyto_lap_opts.mode = 1) against
    # the classic restatement, and -- the instance has a unique optimum -- the same indices from both ----
    pn = 3000
    pc = make_cost(pn, 1234 + rank)
    g = lap_solve(pc, np.float32, device_id=dev, return_info=True)
    ow = jv_oracle_wide(pc, np.float32)
    gc = lap_solve(pc, np.float32, device_id=dev, return_info=True, opts=dict(mode=1))
    o = jv_oracle(pc, np.float32)
    parity_small = bool(g["info"].wide == 1 and all(np.array_equal(g[k], ow[k]) for k in ("rowsol", "colsol", "u", "v"))
                        and g["info"].scans_arr == ow["stats"].scans_arr and g["info"].scans_aug_relax == ow["stats"].scans_aug_relax
                        and all(np.array_equal(gc[k], o[k]) for k in ("rowsol", "colsol", "u", "v"))
                        and gc["info"].row_scans == o["stats"].row_scans
                        and np.array_equal(g["colsol"], o["colsol"]))
    if not parity_small:
        raise SystemExit("parity gate failed: HIP solver differs from the CPU oracle")

    # ---- the workload, resident in HBM before the timed region (every rank: the same seeded instance, so every rank's result
    # is checked against the committed golden of that size) ----
    from tools import instances
    buf = instances.blocks_to_device(instances.uniform_cost_blocks(n), n, dev)
    res = None
    for _ in range(args.warmup):
        res = lap_solve(None, np.float32, device_id=dev, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
    barrier()
    t0 = time.perf_counter()
    infos = []
    for _ in range(args.steps):
        res = lap_solve(None, np.float32, device_id=dev, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
        infos.append(res["info"])
    barrier()
    elapsed = time.perf_counter() - t0
    if store is not None:
        elapsed = float(store.allreduce_max(elapsed))

    # ---- parity of the headline itself: the committed goldens (required at the default sizes), and size-independent properties
    # on rows drawn again from the generator (the 10 GB matrix is not kept on the host) ----
    info = res["info"]
    colsol, rowsol = res["colsol"], res["rowsol"]
    gold_idx, gold_duals = check_uniform_golden(n, res, bool(info.wide), required=n in (20000, 50000))
    perm_ok = bool(np.array_equal(np.sort(colsol), np.arange(n)) and np.array_equal(rowsol[colsol], np.arange(n)))
    tot64, dual_ok = 0.0, True
    u64, v64 = res["u"].astype(np.float64), res["v"].astype(np.float64)
    for lo, blk in instances.uniform_cost_blocks(n):
        rows = np.arange(lo, lo + len(blk))
        tot64 += float(blk[np.arange(len(blk)), rowsol[rows]].astype(np.float64).sum())
        if lo % (8 * 2048) == 0:             # dual feasibility (u_i + v_j <= c_ij) and complementary slackness on a sample of row blocks
            red = blk[:256].astype(np.float64) - u64[rows[:256], None] - v64[None, :]
            dual_ok = dual_ok and bool(red.min() > -1e-5 and np.abs(red[np.arange(len(red)), rowsol[rows[:256]]]).max() < 1e-5)
    total_ok = abs(tot64 - res["total"]) <= 1e-5 * max(1.0, abs(tot64))
    if not (perm_ok and total_ok and dual_ok):
        raise SystemExit(f"full-size property check failed: perm={perm_ok} total={total_ok} dual={dual_ok}")
    if store is not None and not all(store.allgather(bool(perm_ok and total_ok and dual_ok and gold_idx is not False))):
        raise SystemExit("a rank's headline result failed its checks")
    buf.free()

    extras = {}
    c2_res = None
    if world == 1 and not args.no_extras:
        other = 20000 if n != 20000 else 50000
        extras["c2" if other == 20000 else "n50000"], obuf, c2_res = extra_uniform(dev, other)
        if other == 20000:
            extras["c2_batch"], r0 = extra_c2_batch(dev, obuf, other)
            if not (np.array_equal(r0["colsol"], c2_res["colsol"]) and np.array_equal(r0["v"], c2_res["v"])):
                raise SystemExit("c2_batch: the batched solve differs from the single solve")
        obuf.free()

    sharded, sharded_stuck, box = None, False, {}
    logical = world > 1 and comm is None and args.oversubscribe
    if not args.no_extras and logical and rank == 0:
        # more ranks than devices (a harness self-test): the multi-rank chunk legs -- rank r's chunks, the rank that never sees the ST
        # matrix, the status agreement, the operand broadcast -- run as `world` THREADS of this process on the in-process communicator
        # (logical ranks of one device: comm.hip's LOCAL kind); the other processes of the job have nothing to do here
        import threading
        lcomms = _lib.Communicator.init_local([dev] * world)
        tstore = ThreadStore(world)
        lbox = [dict() for _ in range(world)]

        def _lrank(r):
            try:
                lbox[r]["strong"] = extra_c4_sharded(dev, r, world, tstore.view(r), lcomms[r], 0, G=2000, total_chunks=20)
                lbox[r]["r"] = extra_c4_sharded(dev, r, world, tstore.view(r), lcomms[r], max(1, args.c4_rank_chunks // world))
            except BaseException as e:          # noqa: BLE001
                lbox[r]["r"] = {"error": f"{type(e).__name__}: {e}"}
                for c in lcomms:
                    c.abort()
                tstore._bar.abort()
        ths = [threading.Thread(target=_lrank, args=(r,), daemon=True) for r in range(world)]
        for t_ in ths:
            t_.start()
        for t_ in ths:
            t_.join(args.sharded_timeout)
        sharded_stuck = any(t_.is_alive() for t_ in ths)
        errs = [b["r"] for b in lbox if isinstance(b.get("r"), dict) and "error" in b["r"]]
        if sharded_stuck:
            sharded = {"error": f"no result within {args.sharded_timeout} s"}
        elif errs:
            sharded = errs[0]
        else:
            sharded = dict(lbox[0]["r"], logical_ranks=f"{world} ranks as threads of rank 0's process on device {dev} (in-process communicator)")
            box["strong"] = dict(lbox[0]["strong"], logical_ranks=sharded["logical_ranks"])
            for c in lcomms:
                c.close()
    if not args.no_extras and (world == 1 or comm is not None):
        # every rank takes part.  The headline above is already measured: a failure or a hang of this additional multi-rank leg
        # must not cost the line, so it runs under a watchdog and reports what happened instead.
        import threading
        if comm is None:                        # N = 1: a one-rank RCCL communicator (the broadcast still goes through RCCL)
            comm = _lib.Communicator(_lib.Communicator.unique_id(), 0, 1, device_id=dev)

        def _leg():
            try:
                box["strong"] = extra_c4_sharded(dev, rank, world, store, comm, 0, G=2000, total_chunks=20)
                box["r"] = extra_c4_sharded(dev, rank, world, store, comm, args.c4_rank_chunks)
            except BaseException as e:          # noqa: BLE001 (SystemExit from the leg's own gates included)
                box["r"] = {"error": f"{type(e).__name__}: {e}"}

        th = threading.Thread(target=_leg, daemon=True)
        th.start()
        th.join(args.sharded_timeout)
        sharded_stuck = th.is_alive()
        sharded = {"error": f"no result within {args.sharded_timeout} s"} if sharded_stuck else box.get("r")
    failed = sharded_stuck or (isinstance(sharded, dict) and "error" in sharded)
    if comm is not None and not failed:
        comm.close()
    if rank != 0:
        if failed:
            os._exit(0)                         # (a stuck collective would keep the interpreter from exiting)
        return
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * n * args.steps / elapsed

    # ---- roofline: the floor.  Every cost entry is read at least once: 4 n^2 bytes per solve.  achieved = floor bytes / step time
    # (<= peak by construction).  What the solve really waits for is latency: dependent rounds, listed beside it. ----
    def avg(f):
        return float(np.mean([getattr(i_, f) for i_ in infos]))
    arr_ms, aug_ms, total_ms = avg("ms_arr"), avg("ms_aug"), avg("ms_total")
    floor_bytes = 4.0 * n * n
    achieved = floor_bytes / (ms_per_step * 1e-3) / 1e9
    traffic, traffic_source, pmk = whole_solve_traffic(args.pmc_tag, n)
    dom_name = "wide_aug" if aug_ms >= max(info.ms_colred, info.ms_cache) else "colred_partial"
    dom_ms = aug_ms if dom_name == "wide_aug" else float(info.ms_colred)
    dom_pm = None
    if pmk:
        key = [k for k in pmk if k.startswith(dom_name)]
        if key:
            dom_pm = pmk[key[0]].get("hbm_read_bytes", 0) + pmk[key[0]].get("hbm_write_bytes_uncalibrated", 0)
    rounds = max(1, int(info.wide_rounds))
    aug_rounds = max(1, int(info.wide_aug_rounds))
    roofline = {
        "bound": "hbm", "kernel": "whole solve (colred_* + build_row_caches_wave + wide_rt + wide_sc_* + wide_arr + wide_aug)",
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        "definition": "floor: every cost entry read once = 4 n^2 bytes per solve, divided by ms_per_step (wall, max over ranks)",
        "floor_bytes_per_solve": floor_bytes,
        "traffic": traffic,
        "traffic_source": (traffic_source + " (separate rocprofv3 --pmc passes of tools/quick_lap_bench.py at this size, summed over the "
                                            "kernels of one solve; read = 2 x FETCH_SIZE, calibrated on colred_partial; not re-measured in this run)")
        if traffic is not None else None,
        "latency_model": {
            "note": "what bounds the solve: dependent rounds, not bytes",
            "row_reduction": {"ms": round(arr_ms, 3), "rounds": int(info.wide_rounds), "phases": int(info.wide_phases),
                              "us_per_round": round(arr_ms * 1e3 / rounds, 2), "bids": int(info.scans_arr),
                              "full_row_bids": int(info.wide_dense_arr), "scaled": bool(info.wide_scaled)},
            "searches": {"ms": round(aug_ms, 3), "searches": int(info.augmentations), "batches_of_searches": int(info.wide_par_batches),
                         "discarded_and_rerun": int(info.wide_par_discarded), "rounds": int(info.wide_aug_rounds),
                         "us_per_round": round(aug_ms * 1e3 / aug_rounds, 2), "columns_settled": int(info.scans_aug_relax),
                         "columns_settled_speculatively": int(info.wide_aug_settled), "full_row_relaxations": int(info.wide_dense_aug)},
            "streaming": {"colred_ms": round(float(info.ms_colred), 3), "colred_GBs": round(floor_bytes / (info.ms_colred * 1e-3) / 1e9, 1),
                          "row_caches_ms": round(float(info.ms_cache), 3), "row_caches_GBs": round(floor_bytes / (info.ms_cache * 1e-3) / 1e9, 1),
                          "note": "the two kernels that really stream the matrix: bytes = 4 n^2 each"}},
        "dominant_kernel": {"name": dom_name, "ms_per_launch_hip_events": round(dom_ms, 3), "hbm_bytes_per_launch_pmc": dom_pm,
                            "note": "the longest single launch of a solve; its HIP-event time is what profiles/<tag>_kernel_stats_*.csv must show"},
        "kernel_ms_sum": round(total_ms, 3), "hbm_rows_actually_read": int(info.hbm_row_reads),
        "row_scans_counted": int(info.row_scans),
    }

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cn = args.cpu_n if args.cpu_n > 0 else min(n, 20000)
        cc = make_cost(cn, cn)
        t1 = time.perf_counter()
        oc = jv_oracle(cc, np.float32)
        dt = time.perf_counter() - t1
        same_as_gpu = None
        if c2_res is not None and cn == 20000:            # the GPU solved this very instance in the c2 leg
            same_as_gpu = bool(np.array_equal(c2_res["colsol"], oc["colsol"]) and np.array_equal(c2_res["rowsol"], oc["rowsol"]))
            if not same_as_gpu:
                raise SystemExit("cpu_baseline: the HIP solver's indices differ from the CPU oracle's on the 20 000 x 20 000 sample")
        scipy_leg = None
        try:   # an independent second CPU number, on a bounded sample of the same generator
            from scipy.optimize import linear_sum_assignment
            sn = min(n, 6000)
            scost = make_cost(sn, sn)
            t3 = time.perf_counter()
            sr, scol = linear_sum_assignment(scost.astype(np.float64))
            dts = time.perf_counter() - t3
            sg = lap_solve(scost, np.float32, device_id=dev)
            scipy_leg = {"value": round(sn / dts, 1), "unit": "assignments/s", "cores": 1, "n": sn, "seconds": round(dts, 2),
                         "same_assignment_as_the_hip_solver": bool(np.array_equal(sg["rowsol"], scol)),
                         "note": "scipy.optimize.linear_sum_assignment (float64) on a smaller instance of the same generator; assignments/s falls with n"}
        except Exception as e:   # noqa: BLE001
            scipy_leg = {"error": f"{type(e).__name__}: {e}"}
        full = None
        try:
            gd = np.load(os.path.join(ROOT, "tests", "golden", f"large_u{n}.npz"))
            full = {"seconds_in_the_build_container": round(float(gd["oracle_seconds"]), 1),
                    "assignments_per_s": round(n / float(gd["oracle_seconds"]), 1),
                    "note": f"the same oracle on the FULL {n} x {n} instance, one core, when the golden was made (not timed in this run)"}
        except (OSError, KeyError):
            pass
        cpu = {"value": round(cn / dt, 1), "unit": "assignments/s", "cores": 1, "kind": "port", "scipy": scipy_leg,
               "sample_n": cn, "seconds": round(dt, 1), "same_indices_as_the_hip_solver_on_this_sample": same_as_gpu,
               "full_instance": full,
               "sample": f"oracle/jv_oracle.c (C port of JV, -O3 -mavx2, 1 thread; lapjv wheel unavailable) on "
                         + ("the bench instance itself" if cn == n else "a smaller instance of the same generator (bounded sample)")
                         + f" ({cn}x{cn} uniform): {dt:.1f} s, {oc['stats'].row_scans} row scans; assignments/s falls with n "
                           "(the golden file records the full-size time: cpu_baseline.full_instance)",
               "cpu_model": _cpu_model(), "host_cores": os.cpu_count()}

    out = {
        "metric": "cell-to-spot assignments/sec on NxN synthetic cost; bit-exact vs lapjv",
        "metric_note": "bit-exact vs the in-tree CPU JV oracle: indices == the committed uniqueness-certified golden of this instance (classic-order "
                       "oracle == scipy), rowsol and duals == the wide restatement's golden; the lapjv wheel is not in this image.  Result contract "
                       "of the default solver: an optimal assignment -- where the optimum is not unique (duplicated spot rows, integer costs) its "
                       "choice among the optimal assignments differs from the classic Jonker-Volgenant tie-break (same spots, same total); "
                       "cyto_lap_opts.mode = 1 gives the classic order",
        "value": round(value, 1), "unit": "assignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n}x{n} dense uniform float32 cost resident in HBM, JV HIP solver "
                               + ("(the north star's target size; fits one GPU: 10 GB)" if n == 50000 else
                                  "(BASELINE.json configs[1])" if n == 20000 else "(size given with --n)"),
                   "n": n, "lap_per_gpu": 1, "parallelism": f"independent LAPs x{world} (inside one LAP: replicas only)"},
        "rccl": rccl,
        "parity": {"bit_exact_vs_cpu_oracle_n3000": parity_small, "indices_equal_the_certified_golden": gold_idx,
                   "duals_equal_the_wide_oracle_golden": gold_duals, "full_size_permutation": perm_ok,
                   "full_size_total_1e-5": bool(total_ok), "full_size_dual_feasible": dual_ok,
                   "note": "oracle = C restatement of JV; the lapjv wheel is not available in this image"},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    out.update(extras)
    if box.get("strong") is not None:
        out["c4_strong"] = box["strong"]
    if sharded is not None:
        out["c4_sharded"] = sharded
    if world == 1 and not args.no_extras:
        out["c2_cytolike"] = extra_c2_cytolike(dev)
        out["c3"] = extra_c3(dev)
        out["c4_chunks"] = extra_c4_chunks(dev, args.c4_chunks)
        out["c5_chunks"] = extra_c5_chunks(dev)
    json_out.write(json.dumps(out) + "\n")
    json_out.flush()
    if failed:
        os._exit(0)                             # the line is out; a stuck collective must not keep the process alive


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
