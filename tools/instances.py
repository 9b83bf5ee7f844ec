"""Seeded synthetic instances shared by bench.py, the -m gpu tests and tests/golden/make_golden_large.py.

Every generator here is reproducible BIT FOR BIT on any machine (the committed goldens depend on it):
numpy's PCG64 streams are platform independent, and where a contraction is needed it is done on small integers,
whose float32/float64 sums are exact in any summation order (no BLAS dependence).
SURVEY.md section 8(d) defines the shapes; citations `file:line` are into /root/reference/.
"""
import numpy as np


def uniform_cost_blocks(n, seed=None, block=2048):
    """SURVEY 8(d) "uniform": default_rng(n).random((n, n)).astype(float32) -- drawn in float64, then cast --
    as consecutive row blocks (same stream, row-major fill): n = 50 000 then needs no 20 GB temporary."""
    rng = np.random.default_rng(n if seed is None else seed)
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        yield lo, rng.random((hi - lo, n)).astype(np.float32)


def uniform_cost(n, seed=None):
    out = np.empty((n, n), np.float32)
    for lo, blk in uniform_cost_blocks(n, seed):
        out[lo:lo + len(blk)] = blk
    return out


def blocks_to_device(blocks, n, device_id=0):
    """Upload a row-block generator into one n x n float32 device matrix (cytospace_amd._lib.DeviceBuffer)."""
    from cytospace_amd import _lib
    buf = _lib.DeviceBuffer(n * n * 4, device_id)
    for lo, blk in blocks:
        blk = np.ascontiguousarray(blk, dtype=np.float32)
        _lib.check(_lib.lib().cyto_memcpy_h2d(buf.ptr + lo * n * 4, blk.ctypes.data, blk.nbytes, device_id))
    return buf


def _mix32(x):
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def _hash01(rows, cols, seed):
    """20-bit hash of (row, col) as float64 in [0, 1): uint32 integer arithmetic only (exact, platform independent).
    The per-row and per-column words are mixed separately; one multiply per matrix entry combines them."""
    hr = _mix32(rows.astype(np.uint64) * np.uint64(2654435761) + np.uint64(seed))
    hc = _mix32(cols.astype(np.uint64) * np.uint64(40503) + np.uint64(977) * np.uint64(seed) + np.uint64(1))
    with np.errstate(over="ignore"):
        x = hr[:, None] ^ hc[None, :]
        x *= np.uint32(0x9E3779B1)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x85EBCA6B)
    x >>= np.uint32(12)
    return x.astype(np.float64) * (1.0 / (1 << 20))


def typed_unique_cost(S, C, seed, K=10, G=256, proto_amp=88, noise=72, mix=4, own_cells=False, block=1000):
    """S x C float32 "minus correlation"-like cost of S spots against C cells of K cell types
    (what calculate_cost's -matrix_correlation_pearson looks like on typed data,
    cytospace/linear_assignment_solvers/linear_assignment_solvers.py:55): integer expression profiles
    (type prototype + per-cell noise; a spot is the mean of `mix` prototypes + noise -- sub-spot chunks, whose cells are
    unrelated to the spots' own composition -- or, with own_cells, the mean of `mix` of the problem's own cells, as in a
    Visium problem where every spot is the sum of the cells that belong to it), contracted exactly
    (G * max|z|^2 <= 2^24: every partial sum is an integer < 2^24, exact in float32 in ANY summation order), scaled by a
    power of two, plus a hashed sub-grid term that makes the low mantissa bits generic.  Rows are UNIQUE spots.
    proto_amp / noise were calibrated so that the CPU oracle's work counters on a sub-spot chunk (ARR scans per row,
    scans per search, free rows after ARR) match those of the real pipeline's chunk (5 000 genes, SURVEY 8d generator)."""
    zmax = int(np.sqrt((1 << 24) / G))            # G * zmax^2 <= 2^24
    assert G * zmax * zmax <= (1 << 24)
    rng = np.random.default_rng(seed)
    proto = rng.integers(-proto_amp, proto_amp + 1, (K, G))
    ctype = rng.integers(0, K, C)
    zsc = np.clip(proto[ctype] + rng.integers(-noise, noise + 1, (C, G)), -zmax, zmax).astype(np.float32)      # C x G
    members = None
    if own_cells:
        members = rng.permutation(C)[: S * mix].reshape(S, mix)
        zst = np.floor_divide(zsc[members].sum(1), mix).astype(np.float32)                                    # S x G
    else:
        m = rng.integers(0, K, (S, mix))
        zst = np.clip(proto[m].sum(1) // mix + rng.integers(-noise, noise + 1, (S, G)), -zmax, zmax).astype(np.float32)
    out = np.empty((S, C), np.float32)
    cols = np.arange(C)
    for lo in range(0, S, block):
        hi = min(S, lo + block)
        dot = (zst[lo:hi] @ zsc.T).astype(np.float64)              # exact integers (|dot| < 2^24)
        if members is not None:   # a spot's own cells share their counts with it: a clear (exact, integer) preference
            dot[np.arange(hi - lo)[:, None], members[lo:hi]] += float(1 << 19)
        out[lo:hi] = -dot / float(1 << 24) + _hash01(np.arange(lo, hi), cols, seed) * (1.0 / (1 << 24))
    return out, ctype


def c3_shaped_cost(n=50000, slots_per_spot=10, seed=3):
    """BASELINE config c3's LAP shape: n cells x (n / slots) spots, every spot row repeated `slots` times, contiguous,
    in spot order (calculate_cost's np.repeat gather, linear_assignment_solvers.py:63-66).
    Returns (cost n x n float32, location_repeat int64[n])."""
    uniq, loc = c3_shaped_unique(n, slots_per_spot, seed)
    return uniq[loc], loc


def c3_shaped_unique(n=50000, slots_per_spot=10, seed=3):
    """The unique spot rows of c3_shaped_cost (S x n) and location_repeat."""
    S = n // slots_per_spot
    assert S * slots_per_spot == n
    uniq, _ = typed_unique_cost(S, n, seed, G=1024, proto_amp=44, noise=36, mix=slots_per_spot, own_cells=True)
    return uniq, np.repeat(np.arange(S), slots_per_spot)


def repeated_row_blocks(uniq, loc, block=1000):
    """Row blocks of uniq[loc] without materialising it (10 GB at n = 50 000)."""
    for lo in range(0, len(loc), block):
        yield lo, uniq[loc[lo:lo + block]]


def c4_chunk_cost(n=10000, seed=4, dup_frac=0.073):
    """The LAP of one --sampling-sub-spots chunk of config c4 (cytospace.py:436-439): n cells of 10 types against
    the spots that receive cells in this chunk; most spots take one cell, a few take two or three.
    These are the deep-search instances (thousands of near-equal columns per search)."""
    rng = np.random.default_rng(seed)
    S = int(n * (1.0 - dup_frac))
    slots = np.ones(S, np.int64)
    extra = n - S
    np.add.at(slots, rng.integers(0, S, extra), 1)
    uniq, _ = typed_unique_cost(S, n, seed)
    loc = np.repeat(np.arange(S), slots)
    return uniq[loc], loc


def synth_expression(G, C, S, seed=1, K=10, dtype=np.float32):
    """SURVEY 8(d) pipeline generator: gene means LogNormal(0, 1.5), K cell types with LogNormal(0, 0.75) multipliers,
    scRNA counts Poisson(0.3 m_g mult), a spot = the sum of its slots' cells.  Returns (sc G x C, st G x S, slots)."""
    rng = np.random.default_rng(seed)
    m = rng.lognormal(0.0, 1.5, G).astype(np.float32)
    mult = rng.lognormal(0.0, 0.75, (K, G)).astype(np.float32)
    types = rng.integers(0, K, C)
    slots = np.full(S, C // S, np.int64)
    slots[: C - slots.sum()] += 1
    sc = np.empty((G, C), dtype)
    for lo in range(0, C, 5000):
        hi = min(C, lo + 5000)
        rate = 0.3 * m[:, None] * mult[types[lo:hi]].T
        sc[:, lo:hi] = rng.poisson(rate)
    st = np.zeros((G, S), dtype)
    perm = rng.permutation(C)
    order = np.argsort(np.repeat(np.arange(S), slots), kind="stable")
    spot_of = np.repeat(np.arange(S), slots)
    # column sums by spot without a Python loop over spots: sort the permuted cells by spot and reduceat
    cells = perm[order]
    starts = np.concatenate([[0], np.cumsum(slots)[:-1]])
    nz = slots > 0
    for lo in range(0, G, 2000):
        hi = min(G, lo + 2000)
        blk = sc[lo:hi][:, cells]
        st[lo:hi][:, nz] = np.add.reduceat(blk, starts[nz], axis=1)
    del spot_of
    return sc, st, slots


def single_cell_expression(G, C, S, seed=5, K=10, depth_st=0.6, dtype=np.float32):
    """Single-cell-resolution ST (BASELINE configs[4]: every spot holds ONE cell, and its cells are not the scRNA cells): the same
    gene means and cell-type multipliers as synth_expression, scRNA counts Poisson(0.3 m mult[type]), spot counts
    Poisson(depth_st * 0.3 m mult[type']) for independently drawn types.  Returns (sc G x C, st G x S)."""
    rng = np.random.default_rng(seed)
    m = rng.lognormal(0.0, 1.5, G).astype(np.float32)
    mult = rng.lognormal(0.0, 0.75, (K, G)).astype(np.float32)
    sc = np.empty((G, C), dtype)
    st = np.empty((G, S), dtype)
    for out, n, depth in ((sc, C, 1.0), (st, S, depth_st)):
        for lo in range(0, n, 5000):
            hi = min(n, lo + 5000)
            ty = rng.integers(0, K, hi - lo)
            out[:, lo:hi] = rng.poisson(depth * 0.3 * m[:, None] * mult[ty].T)
    return sc, st
