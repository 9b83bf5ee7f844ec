"""K concurrent sub-spot chunk LAPs (bench.py's c4_chunks leg without the CPU side): wall time and what problem 0 reports.
usage: batch_chunks_bench.py [K] [n]   (developer tool; CYTO_ARR_WASTE=<full-row bids per group that ask for a cache rebuild> to experiment)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cytospace_amd import _lib  # noqa: E402
from cytospace_amd.lap import lap_solve_batch_device  # noqa: E402
from tools import instances  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
distinct = 4
costs = [instances.c4_chunk_cost(n, seed=4 + k)[0] for k in range(distinct)]
bufs = [_lib.DeviceBuffer.from_numpy(costs[k]) for k in range(distinct)]
bufs += [bufs[k % distinct].clone() for k in range(distinct, K)]
lap_solve_batch_device([bufs[0].ptr], [n], max_concurrent=1)
for rep in range(2):
    t = time.perf_counter()
    res = lap_solve_batch_device([b.ptr for b in bufs], [n] * K, max_concurrent=K, return_info=True)
    wall = time.perf_counter() - t
    i = res[0]["info"]
    print(f"K={K} n={n} rep={rep}: wall {wall * 1e3:.0f} ms => {K * n / wall / 1e6:.2f} M assignments/s | problem 0: arr {i.ms_arr:.0f} ms aug {i.ms_aug:.0f} ms "
          f"cache {i.ms_cache:.1f} rounds {i.wide_rounds} dense bids {i.wide_dense_arr} aug launches {i.wide_aug_launches} ARR_WASTE={os.environ.get('CYTO_ARR_WASTE')}", flush=True)
same = all(np.array_equal(res[k]["colsol"], res[k % distinct]["colsol"]) for k in range(K))
print("copies identical:", same)
