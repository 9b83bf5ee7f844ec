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
