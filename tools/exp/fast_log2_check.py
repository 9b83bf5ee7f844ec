"""numpy restatement of cost.hip's log2_ge1 (table-driven double log2 for arguments >= 1) against long-double log2:
prints the maximum absolute and relative error on [1, 1e7], next to 1, on exact powers of two and on [1, 1 + 1/64); also prints
the table and the polynomial coefficients the kernel carries.  (numpy evaluates Horner without FMAs: the kernel is no worse.)"""
import math
import numpy as np

idx = np.arange(64)
c = (129 + 2 * idx) / 128.0
c[0] = 1.0
inv_c = 1.0 / c
log2_c = np.array([math.log2(x) for x in c])
DEG = 9
coef = [1.4426950408889634 / k * (1 if k % 2 else -1) for k in range(1, DEG + 1)]


def fast_log2(t):
    mant, ex = np.frexp(t)
    m = mant * 2.0
    e = (ex - 1).astype(np.float64)
    i = ((m.view(np.uint64) >> np.uint64(46)) & np.uint64(63)).astype(np.int64)
    r = (m - c[i]) * inv_c[i]
    p = np.full_like(r, coef[DEG - 1])
    for k in range(DEG - 2, -1, -1):
        p = p * r + coef[k]
    return (e + log2_c[i]) + r * p


def main():
    rng = np.random.default_rng(0)
    cases = [("[1, 1e7]", np.exp(rng.uniform(0, math.log(1e7), 2000000))),
             ("1 + [1e-12, 1]", 1 + np.exp(rng.uniform(math.log(1e-12), 0, 2000000))),
             ("powers of two", 2.0 ** np.arange(0, 60)),
             ("[1, 1 + 1/64)", rng.uniform(1, 1 + 1 / 64, 2000000))]
    for name, t in cases:
        y = fast_log2(t)
        ref = np.log2(t.astype(np.longdouble))
        err = np.abs(y.astype(np.longdouble) - ref)
        nz = ref != 0
        rel = float((err[nz] / np.abs(ref[nz])).max()) if nz.any() else 0.0
        print(f"{name:>16}: max abs {float(err.max()):.3e}  max rel {rel:.3e}  log2(1) == 0: {bool((y[t == 1.0] == 0).all())}")
    print("coefficients:", coef)


if __name__ == "__main__":
    main()
