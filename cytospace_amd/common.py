"""Device-backed counterparts of the reference's numeric helpers (cytospace/common/common.py).

Same names, argument meaning and error behaviour; the arithmetic runs in HIP kernels
(cytospace_amd/csrc/cost.hip) through the C ABI.  No numpy fallback exists.
"""
import ctypes

import numpy as np

from . import _lib

_BK, _BM = 32, 128


def _as_matrix(a):
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("expected a 2-D genes x columns matrix")
    if a.dtype == np.float32:
        return np.ascontiguousarray(a), 0
    if a.dtype == np.uint16:                       # (raw counts: CYTO_DTYPE_U16 / _U8, widened on the device)
        return np.ascontiguousarray(a), 2
    if a.dtype == np.uint8:
        return np.ascontiguousarray(a), 3
    return np.ascontiguousarray(a, dtype=np.float64), 1


def normalize_data(data, device_id=0):
    """cytospace/common/common.py:142-147 on the GPU: nan_to_num, per-column CPM, log2(x+1).
    Returns a float64 array like the reference."""
    x, is64 = _as_matrix(data)
    G, C = x.shape
    out = np.empty((G, C), np.float64)
    _lib.check(_lib.lib().cyto_normalize_data(G, C, x.ctypes.data, C, is64, out.ctypes.data, C, device_id))
    return out


METRICS = {"Pearson_correlation": 0, "Spearman_correlation": 1, "Euclidean": 2}   # CYTO_METRIC_*


def is_sparse(x):
    try:
        import scipy.sparse as sp
    except ImportError:
        return False
    return sp.issparse(x)


def sparse_to_device(x, device_id=0):
    """A scipy.sparse genes x columns count matrix -> dense float32 G x ld matrix in HBM (C ABI: cyto_csc_to_dense_f32): only the
    non-zeros cross PCIe and the expansion runs on the device.  The reference densifies on the host
    (`pd.DataFrame.sparse.from_spmatrix(...).sparse.to_dense()`, cytospace/common/common.py:57).
    Returns (DeviceBuffer, G, C, ld).  Values must be exact in float32 (counts are)."""
    csc = x.tocsc()
    csc.sum_duplicates()
    G, C = csc.shape
    vals = np.ascontiguousarray(csc.data, dtype=np.float32)
    if not np.array_equal(vals.astype(csc.data.dtype, copy=False), csc.data):
        raise ValueError("sparse values are not exactly representable in float32; pass a dense float64 matrix instead")
    colptr = np.ascontiguousarray(csc.indptr, dtype=np.int64)
    rowidx = np.ascontiguousarray(csc.indices, dtype=np.int32)
    ld = -(-C // 4) * 4
    buf = _lib.DeviceBuffer(max(G, 1) * ld * 4, device_id)
    _lib.check(_lib.lib().cyto_csc_to_dense_f32(G, C, len(vals), colptr.ctypes.data, rowidx.ctypes.data if len(vals) else None,
                                                vals.ctypes.data if len(vals) else None, buf.ptr, ld, device_id, None))
    return buf, G, C, ld


def read_file(file_path, keep_sparse=True):
    """cytospace/common/common.py:16-82.  A MatrixMarket file (.mtx / .mtx.gz, genes x cells, with genes|features and
    cells|barcodes lists beside it) or a delimited text table (.csv: ','; otherwise tab) with gene ids in the first column.
    Returns a pandas DataFrame like the reference -- except that, with keep_sparse, a MatrixMarket input stays sparse
    (a DataFrame of pandas sparse columns, `df.sparse.to_coo()` recovers the matrix): ExpressionContext uploads it as non-zeros."""
    import os
    import pandas as pd
    if file_path.endswith(".mtx") or file_path.endswith(".mtx.gz"):
        import scipy.io
        if not os.path.isfile(file_path):
            raise IOError("Cannot locate file: {}".format(file_path))
        base = os.path.dirname(file_path) + os.path.sep

        def find(names):
            for name in names:
                for ext in (".tsv", ".csv", ".tsv.gz", ".csv.gz"):
                    if os.path.isfile(f"{base}{name}{ext}"):
                        return f"{base}{name}{ext}", ext
            raise IOError(f"Required files not found for base path: {base}")
        gpath, gext = find(["genes", "features"])
        cpath, cext = find(["cells", "barcodes"])
        genes = pd.read_csv(gpath, sep="\t" if ".tsv" in gext else ",", header=None).iloc[:, 0].to_numpy()
        cells = pd.read_csv(cpath, sep="\t" if ".tsv" in cext else ",", header=None).iloc[:, 0].to_numpy()
        m = scipy.io.mmread(file_path)
        if m.shape != (len(genes), len(cells)):
            raise IOError("The dimensions of the provided sparse matrix does not match the corresponding gene and cell lists. "
                          f"Please check the following files: {gpath}, {cpath}.")
        df = pd.DataFrame.sparse.from_spmatrix(m, index=genes, columns=cells)
        return df if keep_sparse else df.sparse.to_dense()
    sep = "," if file_path.lower().endswith(".csv") else "\t"
    return pd.read_csv(file_path, sep=sep, header=0, index_col=0)


def downsample(data_df, target_count):
    """cytospace/common/common.py:149-173.  Every cell (column) with more than target_count transcripts is reduced to
    target_count draws WITH replacement from its transcripts (np.random.choice over the expanded gene list, legacy
    RandomState: same draws as the reference for the same np.random.seed); other cells are kept.  Host code."""
    import pandas as pd
    index = data_df.index
    values = data_df.to_numpy()
    out = np.array(values, copy=True)
    for c in range(values.shape[1]):
        col = values[:, c]
        if col.sum() <= target_count:
            continue
        picked = np.random.choice(np.repeat(np.arange(len(index)), col), target_count)
        out[:, c] = np.bincount(picked, minlength=len(index))
    return pd.DataFrame(out, index=index, columns=data_df.columns)


class StandardizedMatrix:
    """A gene x column matrix as the float32 GEMM operand of a metric, zero padded, resident in HBM:
    standardised values (Pearson), standardised average-tie ranks (Spearman) or the plain values (Euclidean)."""

    def __init__(self, data, already_normalized=False, device_id=0, metric="Pearson_correlation"):
        x, is64 = _as_matrix(data)
        self.G, self.C = x.shape
        self.Gpad = -(-self.G // _BK) * _BK
        self.ld = -(-self.C // _BM) * _BM
        self.device_id = device_id
        self.buf = _lib.DeviceBuffer(self.Gpad * self.ld * 4, device_id)
        _lib.check(_lib.lib().cyto_transform(METRICS[metric], self.G, self.C, x.ctypes.data, self.C, is64, 0,
                                             int(already_normalized), self.buf.ptr, self.ld, self.Gpad, device_id, None))

    def to_numpy(self):
        return self.buf.to_numpy((self.Gpad, self.ld), np.float32)[:self.G, :self.C]


def pearson_cost_device(sc_norm, st_norm, slots, device_id=0, already_normalized=True, metric="Pearson_correlation"):
    """The cost matrix of calculate_cost's lapjv branch (-Pearson, -Spearman or Euclidean distance, spots x cells)
    with every spot row repeated slots[s] times, left in HBM.

    Returns (DeviceBuffer cost, N, ld, gemm_ms).  sc_norm: G x C, st_norm: G x S."""
    sc_norm = np.asarray(sc_norm)
    st_norm = np.asarray(st_norm)
    if sc_norm.shape[0] != st_norm.shape[0]:
        raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                         "ST and scRNA data must have the same genes")
    slots = np.ascontiguousarray(slots, dtype=np.int64)
    if slots.ndim != 1 or len(slots) != st_norm.shape[1] or (slots < 0).any():
        raise ValueError("cell_number_to_node_assignment must hold one non-negative count per spot")
    if metric not in METRICS:
        raise ValueError(f"unknown distance_metric {metric!r}")
    zsc = StandardizedMatrix(sc_norm, already_normalized, device_id, metric)
    zst = StandardizedMatrix(st_norm, already_normalized, device_id, metric)
    N = int(slots.sum())
    C = zsc.C
    ld = -(-C // 4) * 4
    cost = _lib.DeviceBuffer(max(N, 1) * ld * 4, device_id)
    ms = ctypes.c_double()
    _lib.check(_lib.lib().cyto_cost_metric(METRICS[metric], zst.Gpad, zst.C, C, zst.buf.ptr, zst.ld, zsc.buf.ptr, zsc.ld,
                                           slots.ctypes.data, cost.ptr, ld, ctypes.byref(ms), device_id, None))
    zsc.buf.free()
    zst.buf.free()
    return cost, N, ld, ms.value


def matrix_correlation_pearson(v1, v2, device_id=0):
    """cytospace/common/common.py:190-199 on the GPU: corr[s, c] of column s of v2 and column c of v1.
    float32 result (standardise-then-contract on the fp32 matrix cores)."""
    v1 = np.asarray(v1)
    v2 = np.asarray(v2)
    if v1.shape[0] != v2.shape[0]:
        raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                         "ST and scRNA data must have the same genes")
    S, C = v2.shape[1], v1.shape[1]
    cost, N, ld, _ = pearson_cost_device(v1, v2, np.ones(S, np.int64), device_id, already_normalized=True)
    out = cost.to_numpy((N, ld), np.float32)[:, :C]
    cost.free()
    return -out


def matrix_correlation_spearman(v1, v2, device_id=0):
    """cytospace/common/common.py:202-215 on the GPU: Pearson correlation of the per-column average-tie ranks."""
    v1 = np.asarray(v1)
    v2 = np.asarray(v2)
    if v1.shape[0] != v2.shape[0]:
        raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                         "ST and scRNA data must have the same genes")
    S, C = v2.shape[1], v1.shape[1]
    cost, N, ld, _ = pearson_cost_device(v1, v2, np.ones(S, np.int64), device_id, True, "Spearman_correlation")
    out = cost.to_numpy((N, ld), np.float32)[:, :C]
    cost.free()
    return -out
