"""Goldens for the LARGE LAP instances (BASELINE configs c2/c3/c4 at true size), made in the build container.

Run:  python tests/golden/make_golden_large.py [tag ...]      (tags: u20000 u24000 u30000 u33000 u50000 u70000 c3s50000 c4s10000 c4s16384 t10000 t20000 t30000 k5t20000)
      python tests/golden/make_golden_large.py --wide [tag ...]   the same instances through the oracle's WIDE mode -> large_<tag>_wide.npz
      python tests/golden/make_golden_large.py --f64 [tag ...]    float64 solves (uniform tags only) -> large_<tag>_f64.npz

For every instance (generators: tools/instances.py, reproducible bit for bit on any machine) this stores what the
CPU JV oracle (oracle/jv_oracle.c, float32) returns -- colsol (int32), the float64 re-summed total, sha256 of u and v,
the oracle's row-scan counters -- and CERTIFIES it against an independent exact solver:
  * `dual_certificate`: the oracle's duals are feasible (c_ij - u_i - v_j >= -1e-6 for EVERY entry, evaluated in float64)
    and tight on the assignment, so total - (sum u + sum v) bounds the distance from the optimum: a proof of optimality
    that needs no second solver;
  * scipy.optimize.linear_sum_assignment on the same float32 costs (as float64) must give the same permutation
    (uniform instances) / the same spot for every cell and the same total (instances with duplicated spot rows, where
    the slot within a spot is arbitrary: SURVEY.md 8c, GV5).  It runs in a child process with a time limit (scipy needs
    hours on 50 000 rows that are duplicated ten times); `scipy_checked` records whether it finished;
  * `unique`: re-solving after moving every entry by one float32 ulp in a random direction leaves the answer unchanged
    (SURVEY.md 8d's uniqueness certificate), so ANY exact solver -- lapjv included -- must return these indices.
The -m gpu tests (tests/test_large_gpu.py) regenerate the instance on the GPU box and compare the HIP solver with
these files: no oracle run and nothing of /root/reference is needed there.  Nothing is imported from the reference.
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scipy.optimize import linear_sum_assignment  # noqa: E402

from oracle.jv import jv_oracle, jv_oracle_wide  # noqa: E402
from tools import instances  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SCIPY_LIMIT_S = 2400


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _scipy_child(c32, q):
    r, c = linear_sum_assignment(c32.astype(np.float64))
    colsol = np.empty(len(r), np.int32)
    colsol[c] = r
    q.put(colsol)


def scipy_colsol(c32, limit_s):
    """linear_sum_assignment in a forked child (the cost is shared copy-on-write); None if it exceeds limit_s."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_scipy_child, args=(c32, q))
    p.start()
    try:
        out = q.get(timeout=limit_s)
    except Exception:
        out = None
    if out is None:
        p.terminate()
    p.join()
    return out


def dual_certificate(cost, o):
    """(min reduced cost over all entries, max |reduced cost| on the assignment, duality gap), all in float64."""
    u, v = o["u"].astype(np.float64), o["v"].astype(np.float64)
    n = len(u)
    mn, tight = np.inf, 0.0
    for lo in range(0, n, 1024):
        red = cost[lo:lo + 1024].astype(np.float64) - u[lo:lo + 1024, None] - v[None, :]
        mn = min(mn, float(red.min()))
        rs = o["rowsol"][lo:lo + 1024]
        tight = max(tight, float(np.abs(red[np.arange(len(rs)), rs]).max()))
    total = float(cost[np.arange(n), o["rowsol"]].astype(np.float64).sum())
    return mn, tight, total - float(u.sum() + v.sum())


def perturbed(c32, seed):
    """every entry moved by one float32 ulp up or down (random), block-wise to bound memory"""
    rng = np.random.default_rng(seed)
    out = np.empty_like(c32)
    for lo in range(0, len(c32), 2048):
        blk = c32[lo:lo + 2048]
        up = rng.integers(0, 2, blk.shape, dtype=np.int8).astype(bool)
        out[lo:lo + 2048] = np.where(up, np.nextafter(blk, np.float32(np.inf)), np.nextafter(blk, np.float32(-np.inf)))
    return out


def make(tag):
    t0 = time.time()
    if tag.startswith("u"):
        n = int(tag[1:])
        cost, loc = instances.uniform_cost(n), None
    elif tag.startswith("c3s"):
        n = int(tag[3:])
        cost, loc = instances.c3_shaped_cost(n)
    elif tag.startswith("c4s"):
        n = int(tag[3:])
        cost, loc = instances.c4_chunk_cost(n)
    elif tag.startswith("t") or tag.startswith("k"):     # SURVEY 8(d) "cytospace-like" solver-only instances: few cell types, slots == 1
        n, cost, loc = instance(tag)
    else:
        raise SystemExit(f"unknown tag {tag}")
    print(f"[{tag}] instance in {time.time() - t0:.1f}s, sha256(cost)={sha(cost)[:16]}", flush=True)
    t = time.time()
    o = jv_oracle(cost, np.float32)
    t_or = time.time() - t
    colsol = o["colsol"]
    total = float(cost[colsol, np.arange(n)].astype(np.float64).sum())
    print(f"[{tag}] oracle {t_or:.1f}s total={total:.9f} scans={o['stats'].as_dict()}", flush=True)
    mn, tight, gap = dual_certificate(cost, o)
    cert = bool(mn >= -1e-6 and tight <= 1e-6 and abs(gap) <= 1e-6 * n)
    print(f"[{tag}] dual certificate: min reduced cost {mn:.3e}, max |reduced| on the assignment {tight:.3e}, "
          f"gap {gap:.3e} -> {cert}", flush=True)
    if not cert:
        raise SystemExit(f"[{tag}] the oracle's duals do not certify its assignment: not a golden")
    key = (lambda x: x) if loc is None else (lambda x: loc[x])
    t = time.time()
    sp = scipy_colsol(cost, SCIPY_LIMIT_S) if n <= 50000 else None       # (beyond that its float64 copy alone is 39 GB)
    t_sp = time.time() - t
    if sp is None:
        print(f"[{tag}] scipy did not finish within {SCIPY_LIMIT_S}s (not checked)", flush=True)
    else:
        sp_total = float(cost[sp, np.arange(n)].astype(np.float64).sum())
        same = bool(np.array_equal(key(colsol), key(sp)))
        print(f"[{tag}] scipy {t_sp:.1f}s total={sp_total:.9f} same={'slot' if loc is None else 'spot'}-level: {same}", flush=True)
        if not same or abs(sp_total - total) > 1e-5 * max(1.0, abs(total)):
            raise SystemExit(f"[{tag}] oracle and scipy disagree: not a golden")
    t = time.time()
    # (any exact solver will do for the re-solve: the wide restatement is the faster one on few-cell-type instances)
    p = (jv_oracle_wide if tag[0] in "tk" else jv_oracle)(perturbed(cost, 99), np.float32)
    unique = bool(np.array_equal(key(p["colsol"]), key(colsol)))
    print(f"[{tag}] one-ulp perturbation re-solve {time.time() - t:.1f}s: answer unchanged = {unique}", flush=True)
    st = o["stats"].as_dict()
    np.savez_compressed(
        os.path.join(OUT, f"large_{tag}.npz"), n=n, colsol=colsol.astype(np.int32), total=total,
        cost_sha256=sha(cost), u_sha256=sha(o["u"]), v_sha256=sha(o["v"]), rowsol_sha256=sha(o["rowsol"]),
        unique=unique, spot_level=loc is not None, oracle_seconds=t_or, scipy_seconds=t_sp, scipy_checked=sp is not None,
        dual_certificate=np.array([mn, tight, gap]),
        stats_keys=np.array(list(st.keys())), stats_vals=np.array(list(st.values()), np.int64))
    print(f"[{tag}] written ({os.path.getsize(os.path.join(OUT, f'large_{tag}.npz')) / 1e3:.0f} kB), {time.time() - t0:.0f}s", flush=True)


def instance(tag):
    if tag.startswith("u"):
        n = int(tag[1:])
        return n, instances.uniform_cost(n), None
    if tag.startswith("c3s"):
        n = int(tag[3:])
        return (n,) + instances.c3_shaped_cost(n)
    if tag.startswith("c4s"):
        n = int(tag[3:])
        return (n,) + instances.c4_chunk_cost(n)
    if tag.startswith("t"):
        n = int(tag[1:])
        return n, instances.typed_unique_cost(n, n, 20)[0], None
    if tag.startswith("k"):                   # k<K>t<n>: the same generator with K cell types instead of ten (seed 20 + K)
        K, n = (int(x) for x in tag[1:].split("t"))
        return n, instances.typed_unique_cost(n, n, 20 + K, K=K)[0], None
    raise SystemExit(f"unknown tag {tag}")


def make_wide(tag):
    """The wide-mode restatement (oracle/jv_oracle_impl.h, WIDE MODE) on the same instance: what the HIP wide solver must return
    bit for bit (rowsol / u / v by sha256, its counters).  Certified by its own duals and, where large_<tag>.npz exists, by giving
    the classic mode's indices (spot level where spot rows are duplicated)."""
    t0 = time.time()
    n, cost, loc = instance(tag)
    t = time.time()
    o = jv_oracle_wide(cost, np.float32)
    t_or = time.time() - t
    colsol = o["colsol"]
    total = float(cost[colsol, np.arange(n)].astype(np.float64).sum())
    print(f"[{tag} wide] oracle {t_or:.1f}s total={total:.9f} stats={o['stats'].as_dict()}", flush=True)
    mn, tight, gap = dual_certificate(cost, o)
    cert = bool(mn >= -1e-6 and tight <= 1e-6 and abs(gap) <= 1e-6 * n)
    print(f"[{tag} wide] dual certificate: min reduced cost {mn:.3e}, max |reduced| on the assignment {tight:.3e}, gap {gap:.3e} -> {cert}", flush=True)
    if not cert:
        raise SystemExit(f"[{tag} wide] the duals do not certify the assignment: not a golden")
    same = None
    cpath = os.path.join(OUT, f"large_{tag}.npz")
    if os.path.exists(cpath):
        d = np.load(cpath)
        key = (lambda x: x) if loc is None else (lambda x: loc[x])
        same = bool(np.array_equal(key(colsol), key(d["colsol"])))
        print(f"[{tag} wide] same {'slot' if loc is None else 'spot'}-level answer as the classic golden: {same}", flush=True)
        if not same or abs(total - float(d["total"])) > 1e-5 * max(1.0, abs(total)):
            raise SystemExit(f"[{tag} wide] wide and classic mode disagree: not a golden")
    st = o["stats"].as_dict()
    np.savez_compressed(
        os.path.join(OUT, f"large_{tag}_wide.npz"), n=n, colsol=colsol.astype(np.int32), total=total, cost_sha256=sha(cost),
        u_sha256=sha(o["u"]), v_sha256=sha(o["v"]), rowsol_sha256=sha(o["rowsol"]), spot_level=loc is not None,
        oracle_seconds=t_or, same_as_classic=-1 if same is None else int(same), dual_certificate=np.array([mn, tight, gap]),
        stats_keys=np.array(list(st.keys())), stats_vals=np.array(list(st.values()), np.int64))
    print(f"[{tag} wide] written, {time.time() - t0:.0f}s", flush=True)


def make_f64(tag, warm=False):
    """float64 solve (the force_doubles / lapjv_compat precision) of a uniform instance: the float32 matrix as float64, plus a
    float64 term below float32's resolution when warm (else the narrowed matrix IS the matrix).  warm: the warm-started restatement
    (oracle/jv_oracle.c: jv_oracle_warm_f64) -> large_<tag>_f64_warm.npz; must give the cold golden's indices where that exists."""
    t0 = time.time()
    n, cost, loc = instance(tag)
    c64 = cost.astype(np.float64)
    if warm:
        c64 += np.random.default_rng(n).random((n, n)) * 2.0 ** -30
    t = time.time()
    o = jv_oracle(c64, np.float64, warm=warm)
    t_or = time.time() - t
    colsol = o["colsol"]
    total = float(c64[colsol, np.arange(n)].sum())
    print(f"[{tag} f64] oracle {t_or:.1f}s total={total:.12f} scans={o['stats'].as_dict()}", flush=True)
    mn, tight, gap = dual_certificate(c64, o)
    cert = bool(mn >= -1e-12 and tight <= 1e-12 and abs(gap) <= 1e-12 * n)
    print(f"[{tag} f64] dual certificate: {mn:.3e} {tight:.3e} {gap:.3e} -> {cert}", flush=True)
    if not cert:
        raise SystemExit(f"[{tag} f64] not certified")
    st = o["stats"].as_dict()
    if warm:
        sp = scipy_colsol(c64.astype(np.float64), SCIPY_LIMIT_S)
        if sp is None or not np.array_equal(sp, colsol):
            raise SystemExit(f"[{tag} f64 warm] scipy disagrees or did not finish: not a golden")
        print(f"[{tag} f64 warm] scipy gives the same permutation", flush=True)
    np.savez_compressed(
        os.path.join(OUT, f"large_{tag}_f64{'_warm' if warm else ''}.npz"), n=n, colsol=colsol.astype(np.int32), total=total, cost_sha256=sha(cost),
        u_sha256=sha(o["u"]), v_sha256=sha(o["v"]), rowsol_sha256=sha(o["rowsol"]), oracle_seconds=t_or,
        dual_certificate=np.array([mn, tight, gap]), stats_keys=np.array(list(st.keys())), stats_vals=np.array(list(st.values()), np.int64))
    print(f"[{tag} f64] written, {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--wide" in sys.argv:
        for tg in (args or ["u20000", "u50000", "c3s50000", "c4s10000"]):
            make_wide(tg)
    elif "--f64warm" in sys.argv:
        for tg in (args or ["u17000"]):
            make_f64(tg, warm=True)
    elif "--f64" in sys.argv:
        for tg in (args or ["u17000"]):
            make_f64(tg)
    else:
        for tg in (args or ["c3s50000", "c4s10000", "u20000", "u33000", "u50000"]):
            make(tg)
