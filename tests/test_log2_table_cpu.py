"""cost.hip's table-driven log2 (normalize_data's log2(1 + x), common.py:142-147, as the transform kernels evaluate it): the table and
the polynomial the KERNEL SOURCE carries, read out of cost.hip and evaluated by a numpy restatement of log2_ge1, against long-double
log2 -- one ulp, log2(1) == 0 exactly.  (The GPU side: test_cost_gpu.py's gv1 / gv2 goldens run through these kernels.)"""
import math
import os
import re

import numpy as np

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cytospace_amd", "csrc", "cost.hip")


def _kernel_constants():
    s = open(SRC).read()
    body = s[s.index("LOG2_TBL[64][2] = {"):]
    body = body[:body.index("};")]
    pairs = re.findall(r"\{\s*([-0-9.e]+)\s*,\s*([-0-9.e]+)\s*\}", body)
    assert len(pairs) == 64
    tbl = np.array([[float(a), float(b)] for a, b in pairs])
    f = s[s.index("double log2_ge1("):]
    f = f[:f.index("return fma(r, p")]
    lead = float(re.search(r"double p = ([-0-9.e]+);", f).group(1))
    rest = [float(x) for x in re.findall(r"p = fma\(p, r, ([-0-9.e]+)\);", f)]
    return tbl, [lead] + rest                                 # highest degree first


def _log2_ge1(t, tbl, coef):
    mant, ex = np.frexp(t)
    m = mant * 2.0
    e = (ex - 1).astype(np.float64)
    i = ((m.view(np.uint64) >> np.uint64(46)) & np.uint64(63)).astype(np.int64)
    c = np.where(i > 0, (129 + 2 * i) * 0.0078125, 1.0)
    r = (m - c) * tbl[i, 0]
    p = np.full_like(r, coef[0])
    for a in coef[1:]:
        p = p * r + a
    return (e + tbl[i, 1]) + r * p


def test_table_is_what_the_comment_says():
    tbl, coef = _kernel_constants()
    for i in range(64):
        c = 1.0 if i == 0 else (129 + 2 * i) / 128.0
        assert tbl[i, 0] == 1.0 / c and tbl[i, 1] == math.log2(c), i
    assert len(coef) == 9
    for k, a in enumerate(reversed(coef), start=1):           # log2(1 + r) = log2(e) * sum (-1)^(k+1) r^k / k
        assert a == 1.4426950408889634 / k * (1 if k % 2 else -1), k


def test_one_ulp_against_long_double_log2():
    tbl, coef = _kernel_constants()
    rng = np.random.default_rng(0)
    for t in (np.exp(rng.uniform(0, math.log(1e7), 400000)), 1 + np.exp(rng.uniform(math.log(1e-12), 0, 400000)),
              rng.uniform(1, 1 + 1 / 64, 400000), 2.0 ** np.arange(0, 900, 7)):
        y = _log2_ge1(t, tbl, coef)
        ref = np.log2(t.astype(np.longdouble))
        nz = ref != 0
        rel = np.abs(y.astype(np.longdouble)[nz] - ref[nz]) / np.abs(ref[nz])
        assert float(rel.max()) <= 2.3e-16
        assert np.all(y[t == 1.0] == 0.0)
    # a count of zero is t == 1 exactly -> 0; the largest argument the fast path takes
    assert _log2_ge1(np.array([1.0]), tbl, coef)[0] == 0.0
    big = np.array([7.9e307])
    assert abs(_log2_ge1(big, tbl, coef)[0] - math.log2(7.9e307)) <= 2.3e-13
