"""Throughput of K concurrent sub-spot chunk LAPs (10 000 cells each, c4/c5's unit of work) on ONE GPU, K = 1 ... 128.
usage: c4_concurrency.py [n K1 K2 ...] [--rebuild B]   (B: cyto_lap_opts.wide_rebuild; -1 = the row caches are never rebuilt during the searches)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd import _lib
from cytospace_amd.lap import lap_solve_batch_device
from tools import instances
rebuild = 0
if "--rebuild" in sys.argv:
    k = sys.argv.index("--rebuild"); rebuild = int(sys.argv[k + 1]); del sys.argv[k:k + 2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Ks = [int(x) for x in sys.argv[2:]] or [1, 8, 32, 64, 128]
distinct = 4
costs = [instances.c4_chunk_cost(n, seed=4 + k)[0] for k in range(distinct)]
bufs = [_lib.DeviceBuffer.from_numpy(costs[k % distinct]) for k in range(max(Ks))]
lap_solve_batch_device([bufs[0].ptr], [n], max_concurrent=1)
for k in range(min(distinct, len(bufs))):                     # every distinct instance alone: the slowest one bounds a batch from below
    t = time.perf_counter()
    r = lap_solve_batch_device([bufs[k].ptr], [n], max_concurrent=1, return_info=True, opts=dict(wide_rebuild=rebuild))[0]
    i = r["info"]
    print(f"instance {k} alone: {time.perf_counter() - t:.2f} s (arr {i.ms_arr:.0f} ms, aug {i.ms_aug:.0f} ms, launches {i.wide_arr_launches} + {i.wide_aug_launches}, "
          f"full-row bids {i.wide_dense_arr}, full-row relaxations {i.wide_dense_aug})", flush=True)
ref = None
for K in Ks:
    t = time.perf_counter()
    res = lap_solve_batch_device([b.ptr for b in bufs[:K]], [n] * K, max_concurrent=K, return_info=True, opts=dict(wide_rebuild=rebuild))
    wall = time.perf_counter() - t
    ref = res[0]["colsol"] if ref is None else ref
    ok = all(np.array_equal(res[k]["colsol"], res[k % distinct]["colsol"]) for k in range(K)) and np.array_equal(res[0]["colsol"], ref)
    kms = [r["info"].ms_total for r in res]
    i0 = res[0]["info"]
    print(f"      problem 0: arr {i0.ms_arr:.0f} ms in {i0.wide_arr_launches} launches, aug {i0.ms_aug:.0f} ms in {i0.wide_aug_launches}; full-row bids {i0.wide_dense_arr}, relaxations {i0.wide_dense_aug}")
    print(f"K={K:4d}: wall {wall:7.2f} s  {K * n / wall:9.0f} assignments/s  per-chunk kernel ms min/mean/max {min(kms):.0f}/{np.mean(kms):.0f}/{max(kms):.0f}  identical={ok}", flush=True)
