"""The DEFAULT solver's indices against the classic order and an independent exact solver, on instances whose optimum is certified
unique (the wide solver's own restatement shares its constants with the kernel -- a comparison against it cannot
see an answer that depends on them).

  python tools/cross_unique.py --make [--count 320] [--procs 4]      CPU, in the build container: the instances' answers
                                 -> tests/golden/cross_unique.npz: per instance (kind, n, seed, K) the sha256 of colsol, the total
  python tools/cross_unique.py [--first K] [--opts k=v ...]          GPU: the same instances through lap_solve's defaults (certify on)

An instance: `uniform` (default_rng(seed).random((n, n)) as float32) or `typed` (tools/instances.typed_unique_cost: K cell
types, exact integer contraction -- both reproducible bit for bit on any machine).  It enters the file only if (a) the classic
oracle (oracle/jv_oracle.c) and scipy.optimize.linear_sum_assignment return the same permutation and (b) re-solving after moving every
entry by one float32 ulp in a random direction leaves it unchanged (SURVEY 8d's uniqueness certificate): ANY exact solver -- lapjv
included -- must return these indices.  The GPU side compares indices element for element and checks the float64 certificate
(cyto_lap_info.gap_f64: total - optimum <= gap)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import instances  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cross_unique.npz")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def plan(count):
    """(kind, n, seed, K) of candidate k: sizes 300 ... 8 200, more of the small ones; every second one typed with 3 ... 11 cell types"""
    out = []
    for k in range(count):
        r = np.random.default_rng(77000 + k)
        n = int(300 + (8200 - 300) * r.random() ** 1.7)
        if k % 2:
            out.append(("typed", n, 5000 + k, int(r.integers(3, 12))))
        else:
            out.append(("uniform", n, 5000 + k, 0))
    return out


def instance(kind, n, seed, K):
    if kind == "uniform":
        return instances.uniform_cost(n, seed)
    return instances.typed_unique_cost(n, n, seed, K=K, G=64)[0]


def perturbed(c32, seed):
    rng = np.random.default_rng(seed)
    up = rng.integers(0, 2, c32.shape, dtype=np.int8).astype(bool)
    return np.where(up, np.nextafter(c32, np.float32(np.inf)), np.nextafter(c32, np.float32(-np.inf)))


def answer(item):
    from scipy.optimize import linear_sum_assignment
    from oracle.jv import jv_oracle, jv_oracle_wide
    kind, n, seed, K = item
    t = time.time()
    c = instance(kind, n, seed, K)
    o = jv_oracle(c, np.float32)
    r, cc = linear_sum_assignment(c.astype(np.float64))
    sp = np.empty(n, np.int32)
    sp[cc] = r
    same = bool(np.array_equal(sp, o["colsol"]))
    p = jv_oracle_wide(perturbed(c, 99), np.float32)
    unique = bool(np.array_equal(p["colsol"], o["colsol"]))
    total = float(c[o["colsol"], np.arange(n)].astype(np.float64).sum())
    return dict(kind=kind, n=n, seed=seed, K=K, ok=same and unique, colsol_sha=sha(o["colsol"].astype(np.int32)), total=total, secs=time.time() - t)


def make(count, procs):
    import multiprocessing as mp
    os.environ.setdefault("JV_ORACLE_THREADS", "1")
    items = plan(count)
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:
        res = []
        for k, a in enumerate(pool.imap(answer, items, chunksize=1)):
            res.append(a)
            print(f"{k:4d} {a['kind']:8s} n={a['n']:5d} K={a['K']:2d} seed={a['seed']}: {'unique' if a['ok'] else 'NOT certified (left out)'} {a['secs']:.1f}s", flush=True)
    keep = [a for a in res if a["ok"]]
    np.savez_compressed(OUT, kind=np.array([a["kind"] for a in keep]), n=np.array([a["n"] for a in keep], np.int32),
                        seed=np.array([a["seed"] for a in keep], np.int64), K=np.array([a["K"] for a in keep], np.int32),
                        colsol_sha256=np.array([a["colsol_sha"] for a in keep]), total=np.array([a["total"] for a in keep]),
                        candidates=len(res))
    print(f"{len(keep)} of {len(res)} instances certified unique -> {OUT} ({time.time() - t0:.0f}s)")


def run(first, opts):
    from cytospace_amd.lap import lap_solve
    d = np.load(OUT)
    m = len(d["n"]) if first <= 0 else min(first, len(d["n"]))
    bad, worst_gap, t0 = 0, 0.0, time.time()
    for k in range(m):
        kind, n, seed, K = str(d["kind"][k]), int(d["n"][k]), int(d["seed"][k]), int(d["K"][k])
        c = instance(kind, n, seed, K)
        g = lap_solve(c, np.float32, return_info=True, opts=dict(certify=1, **opts))
        i = g["info"]
        ok = sha(g["colsol"].astype(np.int32)) == str(d["colsol_sha256"][k])
        tot_ok = abs(g["total"] - float(d["total"][k])) <= 1e-5 * max(1.0, abs(float(d["total"][k])))
        gap_ok = i.certified == 1 and 0.0 <= i.gap_f64 <= 1e-5 * max(1.0, abs(g["total"]))
        worst_gap = max(worst_gap, float(i.gap_f64))
        bad += not (ok and tot_ok and gap_ok)
        print(f"{k:4d} {kind:8s} n={n:5d} K={K:2d}: indices {'==' if ok else '!='} classic oracle == scipy; total {'ok' if tot_ok else 'OFF'}; certified gap {i.gap_f64:.3e} "
              f"({i.gap_rows} rows, max {i.gap_max_f64:.2e}) scaled {i.wide_scaled}/{i.wide_phases} {i.ms_total:.1f} ms", flush=True)
    print(f"{m} certified-unique instances, {bad} index / total / certificate mismatches, largest certified gap {worst_gap:.3e}, {time.time() - t0:.0f}s")
    return bad


if __name__ == "__main__":
    def flag(name, default):
        if name in sys.argv:
            k = sys.argv.index(name)
            v = int(sys.argv[k + 1])
            del sys.argv[k:k + 2]
            return v
        return default
    if "--make" in sys.argv:
        make(flag("--count", 320), flag("--procs", 4))
    else:
        first = flag("--first", 0)
        opts = {}
        if "--opts" in sys.argv:
            for kv in sys.argv[sys.argv.index("--opts") + 1:]:
                k, v = kv.split("=")
                opts[k] = int(v)
        sys.exit(1 if run(first, opts) else 0)
