"""Config c3: Visium-scale fused pipeline on one GPU -- normalise + standardise + fp32 MFMA Pearson
cost build + slot expansion + JV solve.  Synthetic expression per SURVEY 8(d).  Usage: c3_pipeline.py [G C S]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cytospace_amd.cytospace import assign_pearson

def synth(G, C, S, seed=1, K=10):
    rng = np.random.default_rng(seed)
    m = rng.lognormal(0.0, 1.5, G).astype(np.float32)
    mult = rng.lognormal(0.0, 0.75, (K, G)).astype(np.float32)
    types = rng.integers(0, K, C)
    slots = np.full(S, C // S, np.int64)
    slots[: C - slots.sum()] += 1
    sc = np.empty((G, C), np.float32)
    for lo in range(0, C, 5000):       # chunks: keeps the rate matrix small
        hi = min(C, lo + 5000)
        rate = 0.3 * m[:, None] * mult[types[lo:hi]].T
        sc[:, lo:hi] = rng.poisson(rate)
    # spot s = sum of `slots[s]` random cells (expression already drawn: a spot is a mixture of real cells)
    st = np.zeros((G, S), np.float32)
    perm = rng.permutation(C)
    pos = 0
    for s_ in range(S):
        st[:, s_] = sc[:, perm[pos:pos + slots[s_]]].sum(1)
        pos += slots[s_]
    return sc, st, slots

if __name__ == "__main__":
    metric = "Pearson_correlation"
    use_f64 = "--f64" in sys.argv
    if use_f64:
        sys.argv.remove("--f64")
    for a_ in list(sys.argv):
        if a_.startswith("--metric="):
            metric = a_.split("=", 1)[1]; sys.argv.remove(a_)
    G, C, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20000, 50000, 5000)
    t = time.time(); sc, st, slots = synth(G, C, S); print(f"synthetic G={G} C={C} S={S} in {time.time()-t:.1f}s", flush=True)
    if use_f64:      # the reference's arrays are float64; raw counts are exact in float32 (half the upload)
        sc = sc.astype(np.float64); st = st.astype(np.float64)
    t = time.time()
    mapped, total, info = assign_pearson(sc, st, slots, already_normalized=False, return_info=True, distance_metric=metric)
    wall = time.time() - t
    d = {k: getattr(info, k) for k, _ in info._fields_ if not k.startswith("reserved") and k != "lap"}
    print("assign (%s) wall %.2fs (includes the H2D of the %s inputs)" % (metric, wall, sc.dtype), d, flush=True)
    li = info.lap
    print("LAP: n=%d ms_total=%.1f colred=%.1f cache=%.1f arr=%.1f aug=%.1f scans: arr=%d aug=%d skipped=%d dense=%d groups=%d augmentations=%d sparse_inits=%d rt=%d arr_dense_refresh=%d free_cr=%d free_a1=%d free_a2=%d" % (
        C, li.ms_total, li.ms_colred, li.ms_cache, li.ms_arr, li.ms_aug, li.scans_arr, li.scans_aug_relax, li.aug_scans_skipped,
        li.aug_dense_scans, li.row_groups, li.augmentations, li.aug_sparse_inits, li.scans_redtransfer, li.dense_refreshes, li.free_after_colred, li.free_after_arr1, li.free_after_arr2), flush=True)
    ok = np.array_equal(np.bincount(mapped, minlength=S), slots)
    print("bincount == slots:", ok, " total cost %.6f" % total, " assignments/s (kernels) %.0f" % (C / (li.ms_total * 1e-3)))
