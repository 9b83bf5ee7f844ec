"""The streaming selection of the row-cache builder (cytospace_amd/csrc/lap_jv.hip, cb_stream / cb_compress), restated step by step in
numpy -- the same order of the columns (lane l of the wave owns the quads l, l + 64, ...; U quads per lane and step), the same first
threshold (the 17th ... 35th smallest of the 64 lane minima of the first step, by the length of the row), the same cut when the 64 staging slots run over
(K = max(24, min(48, 54 m / n)) kept, the K-th smallest staged reduced cost becomes the threshold), the same give-up rules.  What is
pinned here is the INVARIANT the solvers' exactness rests on (a cache holds every column of its row whose reduced cost lies
below its floor) and the quality the measurements rest on (48-63 columns as a rule), on orders of the columns the GPU instances do not
have.  The kernel itself: tests/test_lap_gpu.py::test_row_cache_builders_agree and every wide-solver test."""
import numpy as np
import pytest

KC, KCU = 64, 63


def cb_stream(h, U=8):
    """returns (ok, T, staged columns) for one row of reduced costs h (float32), or ok = False where the kernel falls back"""
    n = len(h)
    nfull, ntail = n >> 2, n & 3
    assert nfull >= 64 * U
    lane_of_quad = np.arange(nfull) % 64
    # first step: lane minima of the first U quads of every lane
    first = h[:64 * U * 4].reshape(64 * U, 4).min(axis=1)
    m0 = np.full(64, np.inf, np.float32)
    np.minimum.at(m0, lane_of_quad[:64 * U], first)
    c0 = min(49.0, max(18.0, 54.0 * (256 * U) / n))
    i0 = int(np.float32(64.0) * (np.float32(1.0) - np.exp(np.float32(-(c0 + 1.0) * 0.015625))))
    T = np.sort(m0)[i0]
    stage = []                                                     # (h, column)
    fail = False

    def compress(K):
        nonlocal T, stage
        if len(stage) <= K:
            return
        srt = sorted(x[0] for x in stage)
        T = srt[K]
        stage = [x for x in stage if x[0] < T]

    def one(cols, base):                                           # one element per lane: cols[l] or -1
        nonlocal fail, stage
        hit = [c for c in cols if c >= 0 and h[c] < T]
        if not hit:
            return
        if len(stage) + len(hit) > KC:
            K = int(min(48, max(24, 54 * (4 * base) // n)))
            compress(K)
            hit = [c for c in hit if h[c] < T]
            if not hit:
                return
            if len(stage) + len(hit) > KC:
                fail = True
                return
        stage += [(h[c], c) for c in hit]

    def quads(qs, base):                                           # qs[l] = quad of lane l (or -1)
        if not any(q >= 0 and h[4 * q:4 * q + 4].min() < T for q in qs):
            return
        for e in range(4):
            one([4 * q + e if q >= 0 else -1 for q in qs], base)

    base = 0
    while base + 64 * U <= nfull:
        for u in range(U):
            quads([base + 64 * u + l for l in range(64)], base)
        base += 64 * U
    while base < nfull:
        quads([base + l if base + l < nfull else -1 for l in range(64)], base)
        base += 64
    if ntail:
        one([nfull * 4 + l if l < ntail else -1 for l in range(64)], base)
    if not fail and len(stage) > KCU:
        compress(48)
    ok = (not fail) and len(stage) <= KCU and len(stage) >= 16 and np.isfinite(T)
    return ok, T, sorted(c for _, c in stage)


def _check(h, expect_ok=None):
    h = np.asarray(h, np.float32)
    ok, T, cols = cb_stream(h)
    if ok:
        assert cols == sorted(np.flatnonzero(h < T).tolist())      # EVERY column below the floor, and nothing else
        assert 16 <= len(cols) <= KCU
    if expect_ok is not None:
        assert ok == expect_ok
    return ok, len(cols)


@pytest.mark.parametrize("n", [2048, 2051, 4999, 20000, 50001])
def test_every_column_below_the_floor_is_staged(n):
    rng = np.random.default_rng(n)
    counts = []
    for trial in range(6):
        h = rng.random(n).astype(np.float32) ** (1 + trial % 3)    # uniform, and two skewed distributions
        ok, k = _check(h, expect_ok=True)
        counts.append(k)
    assert min(counts) >= 24 and np.mean(counts) >= 40             # full caches: what the row reduction's full-row bids depend on


def test_few_levels_and_heavy_ties():
    rng = np.random.default_rng(5)
    # a few-cell-type row: clusters of near-equal values
    h = (rng.integers(0, 6, 20000) * 0.1 + rng.random(20000) * 1e-3).astype(np.float32)
    _check(h, expect_ok=True)
    # integer costs: the threshold cannot separate the ties -- either a valid (smaller) cache or the fallback, never a wrong one
    for levels in (2, 7, 40):
        h = rng.integers(0, levels, 20000).astype(np.float32)
        ok, k = _check(h)
        assert ok or k >= 0
    _check(np.zeros(4096, np.float32), expect_ok=False)             # a constant row: nothing lies below any floor


def test_adversarial_orders():
    n = 20000
    # descending: every column undercuts everything before it -- the slots run over again and again; the cuts keep the invariant
    _check(np.linspace(1.0, 0.0, n, dtype=np.float32))
    # ascending: the first step already holds the smallest values
    _check(np.linspace(0.0, 1.0, n, dtype=np.float32), expect_ok=True)
    # large values first, then 3 000 columns far below them: one step brings more than 64 columns under ANY threshold a cut can reach
    h = np.concatenate([np.full(4096, 5.0, np.float32) + np.random.default_rng(1).random(4096).astype(np.float32),
                        np.random.default_rng(2).random(n - 4096).astype(np.float32) * 1e-3])
    ok, _ = _check(h)
    # (whether the selection survives this order or gives up, it never returns an incomplete cache: _check asserted the invariant)
    assert ok in (True, False)
