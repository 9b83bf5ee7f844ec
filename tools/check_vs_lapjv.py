"""One command that pins parity against lapjv ITSELF wherever the wheel exists ("add a one-off cross-check script").

The reference's solver is `from lapjv import lapjv` (PyPI lapjv==1.3.14; /root/reference/cytospace/linear_assignment_solvers/
linear_assignment_solvers.py:16-18, called as `_, y, _ = lapjv(cost)` at :38).  The wheel is in neither the build container nor the GPU
box's image, so the repo's parity claim is "bit-exact vs the in-tree restatement of JV, equal to scipy on certified-unique instances".
On a machine that HAS the wheel (pip install lapjv==1.3.14) this script closes the gap:

  python tools/check_vs_lapjv.py [--gpu] [--large]

  * every certified-unique instance of tests/golden/cross_unique.npz (uniform and few-cell-type, n = 300 ... 8 200) and, with --large,
    the true-size goldens (tests/golden/large_*.npz) is solved by lapjv.lapjv; its `y` (the second element: the row of every column,
    what CytoSPACE keeps) must equal the committed indices -- those are the classic oracle's == scipy's, and every -m gpu test pins the
    HIP solvers to them;
  * the tie-heavy generators of tools/stress_lap.py (duplicated rows, small integers): the optimum is not unique there, so lapjv's
    TOTAL must equal the oracle's (1e-5, BASELINE.json) -- which assignment it picks among the optimal ones is its own business;
  * --gpu: the HIP default solver on the same instances, side by side (needs a gfx950 device).
Exit status 0: everything agrees.  Without the wheel: says so and exits 2 (nothing is checked, nothing is claimed)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    try:
        from lapjv import lapjv
    except ImportError:
        print("lapjv is not installed here (pip install lapjv==1.3.14): parity against the wheel itself stays unpinned on this machine")
        return 2
    from oracle.jv import jv_oracle
    from tools import cross_unique, stress_lap
    gpu = "--gpu" in sys.argv
    if gpu:
        from cytospace_amd.lap import lap_solve
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.int32).tobytes()).hexdigest()   # noqa: E731
    bad = 0
    d = np.load(cross_unique.OUT)
    for k in range(len(d["n"])):
        c = cross_unique.instance(str(d["kind"][k]), int(d["n"][k]), int(d["seed"][k]), int(d["K"][k]))
        _, y, _ = lapjv(c)
        ok = sha(y) == str(d["colsol_sha256"][k])
        if gpu:
            ok = ok and np.array_equal(lap_solve(c, np.float32)["colsol"], y)
        bad += not ok
        print(f"unique {k:4d} {d['kind'][k]} n={d['n'][k]}: lapjv {'==' if ok else '!='} the committed indices", flush=True)
    if "--large" in sys.argv:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden_large as mg
        for f in sorted(os.listdir(os.path.join(ROOT, "tests", "golden"))):
            if not (f.startswith("large_") and f.endswith(".npz")) or "_wide" in f or "_f64" in f:
                continue
            g = np.load(os.path.join(ROOT, "tests", "golden", f))
            n, c, loc = mg.instance(f[6:-4])
            _, y, _ = lapjv(c)
            key = (lambda x: x) if loc is None else (lambda x: loc[x])
            ok = np.array_equal(key(np.asarray(y)), key(g["colsol"]))
            bad += not ok
            print(f"{f}: lapjv {'==' if ok else '!='} the golden ({'slot' if loc is None else 'spot'} level)", flush=True)
    for s in range(24):                                           # ties: totals only
        rng = np.random.default_rng(1000 + s)
        kind = ["dup", "ints", "constcols"][s % 3]
        c = stress_lap.make(kind, int(rng.integers(500, 3000)), rng)
        _, y, _ = lapjv(c)
        n = len(c)
        t = float(c[np.asarray(y), np.arange(n)].astype(np.float64).sum())
        o = jv_oracle(c, np.float32)
        ok = abs(t - o["total"]) <= 1e-5 * max(1.0, abs(o["total"])) and np.array_equal(np.sort(y), np.arange(n))
        bad += not ok
        print(f"ties {s:2d} {kind} n={n}: lapjv total {t:.9f} {'==' if ok else '!='} oracle {o['total']:.9f}", flush=True)
    print(f"{bad} disagreements with lapjv")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
