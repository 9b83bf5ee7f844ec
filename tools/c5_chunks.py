"""Single-cell mode chunks (BASELINE configs[4]'s unit of work: <= 10 000 cells against <= 10 000 single-cell spots, every
slot count 1, no duplicated rows): K chunks through ONE batched context call on one GPU.
usage: python tools/c5_chunks.py [chunk] [genes] [K ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.cytospace import ExpressionContext
from tools import instances

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
Ks = [int(x) for x in sys.argv[3:]] or [1, 8, 50]
sets = 4
sc, st = instances.single_cell_expression(G, sets * chunk, sets * chunk, seed=5)
ones = np.ones(chunk, np.int64)
with ExpressionContext(sc, st, already_normalized=False) as ctx:
    ctx.assign_chunks([(np.arange(256), np.ones(256, np.int64), np.arange(256))])
    for K in Ks:
        work = [(np.arange((k % sets) * chunk, (k % sets + 1) * chunk), ones, np.arange(((k // sets) % sets) * chunk, ((k // sets) % sets + 1) * chunk))
                for k in range(K)]
        t = time.perf_counter()
        res = ctx.assign_chunks(work, max_concurrent=K, return_info=True)
        wall = time.perf_counter() - t
        ok = all(np.array_equal(np.sort(m), np.arange(chunk)) for m, _, _ in res)
        i0 = res[0][2]
        print(f"K={K:3d}: wall {wall:6.2f} s  {K * chunk / wall:9.0f} assignments/s  chunk0: gather {i0.ms_standardize:.1f} gemm {i0.ms_gemm:.1f} "
              f"lap {i0.lap.ms_total:.0f} ms (arr {i0.lap.ms_arr:.0f}, aug {i0.lap.ms_aug:.0f}), scans rt+arr {i0.lap.scans_redtransfer + i0.lap.scans_arr} "
              f"aug {i0.lap.scans_aug_relax} (dense {i0.lap.aug_dense_scans}, skipped {i0.lap.aug_scans_skipped}), searches {i0.lap.augmentations}, "
              f"launches {i0.lap.wide_arr_launches}+{i0.lap.wide_aug_launches} full-row bids {i0.lap.wide_dense_arr} relaxations {i0.lap.wide_dense_aug}  perm={ok}", flush=True)
        L = i0.lap
        print(f"        wide_arr: {L.wide_list_rounds} list rounds {L.wide_ms_list:.1f} ms, {L.wide_chain_rounds} chain rounds {L.wide_ms_chain:.1f} ms, retired {L.wide_retired}, free after {L.free_after_arr2} | "
              f"wide_aug: {L.wide_aug_rounds} rounds {L.wide_ms_aug_rounds:.1f} ms, settled {L.wide_aug_settled}, verify {L.wide_ms_aug_verify:.1f}, finish {L.wide_ms_aug_finish:.1f}, one-edge {L.wide_trivial} in {L.wide_ms_aug_trivial:.1f} ms", flush=True)
