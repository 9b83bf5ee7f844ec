#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CytoSPACE linear-assignment hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full solve of BASELINE.json's configs[1]: a 20000 x 20000 dense synthetic float32 cost
matrix already resident in HBM -> assignment (column reduction, row-cache build, the RT/ARR chain kernel,
the augmentation kernel).  With N GPUs every rank solves its own, differently seeded, instance of the same size (the
reference shards independent sub-LAPs across workers: cytospace.py:430-451; no data-path collective),
so scaling is "weak" and value = N * n / max-over-ranks time.

torch is used only for the rendezvous (barrier + max over ranks); the product never imports it.
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


def make_cost(n, seed):
    # draw in float64, then cast (float32 draws live on a 2^-24 grid)
    return np.random.default_rng(seed).random((n, n)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=20000, help="LAP size (default: BASELINE.json configs[1])")
    ap.add_argument("--cpu-n", type=int, default=0,
                    help="size of the bounded CPU-baseline sample (0: min(n, 20000); when it equals n the very same "
                         "instance is used and the GPU result is compared bit for bit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")

    from cytospace_amd import _lib
    from cytospace_amd.lap import lap_solve
    from oracle.jv import jv_oracle   # checker + cpu_baseline leg only

    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = local_rank % ndev
    n = args.n

    def barrier():
        _lib.check(_lib.lib().cyto_device_synchronize(dev))
        if dist is not None:
            import torch
            t = torch.zeros(1, device="cuda" if torch.cuda.is_available() else "cpu")
            dist.all_reduce(t)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        _lib.check(_lib.lib().cyto_device_synchronize(dev))

    # ---- parity gate on a size the oracle finishes in a second (bit-exact, incl. duals) ----
    pn = 3000
    pc = make_cost(pn, 1234 + rank)
    g = lap_solve(pc, np.float32, device_id=dev, return_info=True)
    o = jv_oracle(pc, np.float32)
    parity_small = bool(np.array_equal(g["colsol"], o["colsol"]) and np.array_equal(g["rowsol"], o["rowsol"])
                        and np.array_equal(g["u"], o["u"]) and np.array_equal(g["v"], o["v"])
                        and g["info"].row_scans == o["stats"].row_scans)
    if not parity_small:
        raise SystemExit("parity gate failed: HIP solver differs from the CPU oracle")

    # ---- the workload, resident in HBM before the timed region ----
    cost = make_cost(n, n + rank)
    buf = _lib.DeviceBuffer.from_numpy(cost, dev)
    res = None
    for _ in range(args.warmup):
        res = lap_solve(None, np.float32, device_id=dev, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
    barrier()
    t0 = time.perf_counter()
    arr_ms_l, aug_ms_l, total_ms = [], [], []
    for _ in range(args.steps):
        res = lap_solve(None, np.float32, device_id=dev, return_info=True, device_ptr=buf.ptr, n=n, ld=n)
        arr_ms_l.append(res["info"].ms_arr)
        aug_ms_l.append(res["info"].ms_aug)
        total_ms.append(res["info"].ms_total)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- size-independent properties at full size (the oracle would need ~1 min here) ----
    info = res["info"]
    colsol, rowsol = res["colsol"], res["rowsol"]
    perm_ok = bool(np.array_equal(np.sort(colsol), np.arange(n)) and np.array_equal(rowsol[colsol], np.arange(n)))
    tot64 = float(cost[np.arange(n), rowsol].astype(np.float64).sum())
    total_ok = abs(tot64 - res["total"]) <= 1e-5 * max(1.0, abs(tot64))
    # dual feasibility (u_i + v_j <= c_ij) and complementary slackness on a row sample
    rs = np.random.default_rng(0).choice(n, size=min(n, 512), replace=False)
    red = cost[rs].astype(np.float64) - res["u"][rs].astype(np.float64)[:, None] - res["v"].astype(np.float64)[None, :]
    dual_ok = bool(red.min() > -1e-5 and np.abs(red[np.arange(len(rs)), rowsol[rs]]).max() < 1e-5)
    if not (perm_ok and total_ok and dual_ok):
        raise SystemExit(f"full-size property check failed: perm={perm_ok} total={total_ok} dual={dual_ok}")
    buf.free()

    if rank != 0:
        return
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * n * args.steps / elapsed

    # roofline of the dominant kernel, per launch, timed with HIP events on the launch stream
    # (cyto_lap_info.ms_arr / ms_aug).  jv_chain2 = reduction transfer + augmenting row reduction (the
    # longest kernel), jv_aug_lazy (n > 5120; jv_aug2 below) = augmentation.  Algorithmic bytes =
    # 4 * n * row scans.
    arr_scans = info.scans_redtransfer + info.scans_arr
    aug_scans = info.scans_aug_init + info.scans_aug_relax
    arr_ms = float(np.mean(arr_ms_l))
    aug_ms = float(np.mean(aug_ms_l))
    aug_name = "jv_aug_lazy" if n > 5120 else "jv_aug2"
    dom = ("jv_chain2", arr_scans, arr_ms) if arr_ms >= aug_ms else (aug_name, aug_scans, aug_ms)
    dom_bytes = 4.0 * n * dom[1]
    achieved = dom_bytes / (dom[2] * 1e-3) / 1e9
    traffic = None
    try:   # HBM bytes from the separate rocprofv3 --pmc pass committed under profiles/ (same n, same instance)
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01e_pmc_traffic_n20000.json")))
        if pm.get("n") == n:
            key = [k for k in pm["kernels"] if k.startswith(dom[0] + "<")]
            if key:
                traffic = pm["kernels"][key[0]]["hbm_read_bytes"] + pm["kernels"][key[0]]["hbm_write_bytes_uncalibrated"]
    except (OSError, ValueError, KeyError):
        traffic = None
    total_avg_ms = float(np.mean(total_ms))
    roofline = {
        "bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "algorithmic_bytes_per_launch": dom_bytes, "row_scans_per_launch": int(dom[1]), "kernel_ms_avg": round(dom[2], 3),
        "other_kernels": {
            "jv_chain2": {"ms": round(arr_ms, 3), "row_scans": int(arr_scans), "algorithmic_GBs": round(4.0 * n * arr_scans / (arr_ms * 1e-3) / 1e9, 2)},
            aug_name: {"ms": round(aug_ms, 3), "row_scans": int(aug_scans), "algorithmic_GBs": round(4.0 * n * aug_scans / max(aug_ms, 1e-9) / 1e-3 / 1e9, 2),
                       "full_row_scans": int(info.aug_dense_scans + info.augmentations - info.aug_sparse_inits)},
            "colred(3 kernels)": {"ms": round(float(info.ms_colred), 3), "GBs": round(4.0 * n * n / (info.ms_colred * 1e-3) / 1e9, 1)},
            "build_row_caches": {"ms": round(float(info.ms_cache), 3), "GBs": round(4.0 * n * n / (info.ms_cache * 1e-3) / 1e9, 1)}},
        "whole_solve": {"row_scans": int(info.row_scans), "bytes": 4.0 * n * info.row_scans, "kernel_ms_avg": round(total_avg_ms, 3),
                        "achieved_GBs": round(4.0 * n * info.row_scans / (total_avg_ms * 1e-3) / 1e9, 2),
                        "floor_4n2_frac": round(4.0 * n * n / (total_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)},
        "hbm_rows_actually_read": int(info.hbm_row_reads),
    }

    cpu = None
    full_size_bit_exact = None
    if not args.no_cpu_baseline and world == 1:
        cn = args.cpu_n if args.cpu_n > 0 else min(n, 20000)
        same = cn == n
        cc = cost if same else make_cost(cn, cn)
        t1 = time.perf_counter()
        oc = jv_oracle(cc, np.float32)
        dt = time.perf_counter() - t1
        if same:   # the workload itself: the GPU result must be the oracle's, bit for bit
            full_size_bit_exact = bool(all(np.array_equal(res[k], oc[k]) for k in ("rowsol", "colsol", "u", "v"))
                                       and info.row_scans == oc["stats"].row_scans)
            if not full_size_bit_exact:
                raise SystemExit("full-size parity failed: HIP solver differs from the CPU oracle on the bench instance")
        cpu = {"value": round(cn / dt, 1), "unit": "assignments/s", "cores": 1, "kind": "port",
               "sample": f"oracle/jv_oracle.c (C port of JV, -O3 -mavx2, 1 thread; lapjv wheel unavailable) on "
                         + ("the bench instance itself" if same else "a smaller instance of the same generator")
                         + f" ({cn}x{cn} uniform): {dt:.1f} s, {oc['stats'].row_scans} row scans "
                         f"({4.0 * cn * oc['stats'].row_scans / dt / 1e9:.1f} GB/s algorithmic); assignments/s falls with n",
               "cpu_model": _cpu_model(), "host_cores": os.cpu_count()}

    out = {
        "metric": "cell-to-spot assignments/sec on NxN synthetic cost; bit-exact vs lapjv",
        "value": round(value, 1), "unit": "assignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n}x{n} dense uniform float32 cost resident in HBM, JV HIP solver "
                               + ("(BASELINE.json configs[1])" if n == 20000 else "(size given with --n; BASELINE.json configs[1] is 20000)"),
                   "n": n, "lap_per_gpu": 1, "parallelism": f"independent LAPs x{world}"},
        "parity": {"bit_exact_vs_cpu_oracle_n3000": parity_small, "full_size_bit_exact_vs_cpu_oracle": full_size_bit_exact,
                   "full_size_permutation": perm_ok,
                   "full_size_total_1e-5": bool(total_ok), "full_size_dual_feasible": dual_ok,
                   "note": "oracle = C restatement of JV; the lapjv wheel is not available in this image"},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))


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
