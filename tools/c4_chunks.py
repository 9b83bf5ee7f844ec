"""Config c4 shape on ONE GPU: C cells x S spots in --sampling-sub-spots chunks of `chunk` cells; the
matrices are uploaded/transformed once (ExpressionContext), all chunks go through the solver together (a workgroup per chunk).
Usage: c4_chunks.py [G C S chunk concurrent]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from c3_pipeline import synth
from cytospace_amd.cytospace import ExpressionContext, partition_indices

G, C, S, chunk, conc = (int(x) for x in sys.argv[1:6]) if len(sys.argv) > 5 else (5000, 200000, 50000, 10000, 64)
t = time.time(); sc, st, slots = synth(G, C, S); print(f"synthetic G={G} C={C} S={S} in {time.time()-t:.1f}s", flush=True)
sc = sc.astype(np.float64); st = st.astype(np.float64)
index_sc = partition_indices(np.arange(C), split_by_interval_int=chunk, shuffle=False)
# per-chunk slot counts: the cells of a chunk are spread over the spots like the reference's sub-spot sampling
rng = np.random.default_rng(0)
slot_ids = rng.permutation(np.repeat(np.arange(S), slots))
sub = [np.bincount(slot_ids[ix], minlength=S) for ix in index_sc]
t0 = time.time()
with ExpressionContext(sc, st, already_normalized=False) as ctx:
    t1 = time.time()
    res = ctx.assign_chunks([(index_sc[k], sub[k]) for k in range(len(index_sc))], max_concurrent=conc, return_info=True)
    t2 = time.time()
ok = all(np.array_equal(np.bincount(m, minlength=S), s_) for (m, _, _), s_ in zip(res, sub))
lap_ms = [i.lap.ms_total for _, _, i in res]; gemm_ms = [i.ms_gemm for _, _, i in res]; gath = [i.ms_standardize for _, _, i in res]
print(f"context (H2D + normalise + standardise, once): {t1-t0:.2f}s; {len(index_sc)} chunks of {chunk}, {conc} at a time: {t2-t1:.2f}s "
      f"=> {C/(t2-t0):.0f} assignments/s end to end, {C/(t2-t1):.0f}/s for the chunk phase; bincount==slots: {ok}")
print(f"per chunk: gather {np.mean(gath):.1f} ms, GEMM {np.mean(gemm_ms):.1f} ms (spots with cells: ~{int(np.mean([(s_>0).sum() for s_ in sub]))}), "
      f"LAP {np.mean(lap_ms):.1f} ms")
li = res[0][2].lap
print("chunk 0 LAP: arr %.0f ms (%d RT + %d ARR scans, %d dense refreshes), aug %.0f ms (%d scans, %d skipped, %d full-row, %d searches, %d sparse inits), free rows %d/%d/%d, handover at search %d" % (
    li.ms_arr, li.scans_redtransfer, li.scans_arr, li.dense_refreshes, li.ms_aug, li.scans_aug_relax, li.aug_scans_skipped, li.aug_dense_scans,
    li.augmentations, li.aug_sparse_inits, li.free_after_colred, li.free_after_arr1, li.free_after_arr2, li.aug_handover))
