"""ctypes binding of oracle/liboracle.so (CPU Jonker-Volgenant oracle).

TEST INFRASTRUCTURE ONLY -- see oracle/jv_oracle_impl.h for what is restated
(the `lapjv` call at /root/reference/cytospace/linear_assignment_solvers/
linear_assignment_solvers.py:38) and why parity against lapjv itself is unpinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class JVStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int64) for k in (
        "scans_colred", "scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax",
        "augmentations", "path_hops", "free_after_colred", "free_after_arr1", "free_after_arr2",
        "arr_budget_hit")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}

    @property
    def row_scans(self):
        return int(self.scans_colred + self.scans_redtransfer + self.scans_arr
                   + self.scans_aug_init + self.scans_aug_relax)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def _usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        # (libgomp's default is one thread per core of the HOST: in a container with 16 of 256 cores the rounds of the wide mode
        #  would run on 256 spinning threads)
        _LIB.jv_oracle_set_threads(int(os.environ.get("JV_ORACLE_THREADS", max(1, min(_usable_cores(), 32)))))
    return _LIB


def jv_oracle(cost, dtype=np.float32, warm=False):
    """Solve the square LAP on the CPU.  Returns dict(rowsol, colsol, u, v, total, total_T, stats).
    warm (float64 only): start from the prices of the float32 wide solve of the narrowed matrix, every row free
    (jv_oracle_warm_f64: what the HIP float64 path computes by default).

    `cost` is cast to `dtype` first (lapjv 1.3.14 is recalled to down-cast to float32
    unless force_doubles is set -- SURVEY.md section 8c, UNVERIFIED)."""
    c = np.ascontiguousarray(cost, dtype=dtype)
    if c.ndim != 2 or c.shape[0] != c.shape[1]:
        raise ValueError("cost must be a square 2-D array")
    n = c.shape[0]
    rowsol = np.empty(n, np.int32)
    colsol = np.empty(n, np.int32)
    u = np.empty(n, dtype)
    v = np.empty(n, dtype)
    tot = ctypes.c_double()
    st = JVStats()
    if dtype == np.float32:
        fn, tt = _lib().jv_oracle_f32, ctypes.c_float()
    elif dtype == np.float64:
        fn, tt = (_lib().jv_oracle_warm_f64 if warm else _lib().jv_oracle_f64), ctypes.c_double()
    else:
        raise TypeError("dtype must be float32 or float64")
    if warm and dtype != np.float64:
        raise TypeError("the warm start is a float64 mode")
    rc = fn(ctypes.c_int(n), c.ctypes.data_as(ctypes.c_void_p),
            rowsol.ctypes.data_as(ctypes.c_void_p), colsol.ctypes.data_as(ctypes.c_void_p),
            u.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p),
            ctypes.byref(tot), ctypes.byref(tt), ctypes.byref(st))
    if rc == 2:
        raise ValueError("cost matrix contains NaN/Inf")
    if rc != 0:
        raise RuntimeError(f"jv_oracle failed with status {rc}")
    return dict(rowsol=rowsol, colsol=colsol, u=u, v=v, total=tot.value, total_T=tt.value, stats=st)


def jv_oracle_trace(cost, rows=1 << 16):
    """Debugging aid: the float32 classic solve plus, per search of the augmentation, (scans, free row, levels, sink column,
    columns scanned at the final distance, unassigned columns at the final distance, final distance)."""
    buf = np.zeros((rows, 7), np.float64)
    L = _lib()
    L.jv_oracle_set_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(rows))
    try:
        o = jv_oracle(cost, np.float32)
    finally:
        L.jv_oracle_set_trace(None, ctypes.c_int(0))
    return o, buf[:int(o["stats"].augmentations)]


class JVWideStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int64) for k in (
        "scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax", "augmentations", "path_hops",
        "free_after_colred", "free_after_arr", "arr_rounds", "arr_retired", "arr_active_left",
        "arr_scaled", "arr_phases", "gap_exp")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def jv_oracle_wide(cost, dtype=np.float32, max_rounds=-1, stop_phase=0):
    """The wide-mode restatement (jv_oracle_impl.h, second half): Jacobi reduction transfer, Jacobi rounds of augmenting row
    reduction, succ-clamped shortest-path augmentation -- what the HIP "wide" solver computes bit for bit.  Same optimum as
    jv_oracle.  stop_phase 1 / 2: the state after reduction transfer / after the row-reduction rounds."""
    c = np.ascontiguousarray(cost, dtype=dtype)
    if c.ndim != 2 or c.shape[0] != c.shape[1]:
        raise ValueError("cost must be a square 2-D array")
    n = c.shape[0]
    rowsol = np.empty(n, np.int32)
    colsol = np.empty(n, np.int32)
    u = np.empty(n, dtype)
    v = np.empty(n, dtype)
    tot = ctypes.c_double()
    st = JVWideStats()
    if dtype == np.float32:
        fn, tt = _lib().jv_oracle_wide_f32, ctypes.c_float()
    elif dtype == np.float64:
        fn, tt = _lib().jv_oracle_wide_f64, ctypes.c_double()
    else:
        raise TypeError("dtype must be float32 or float64")
    fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int]
    rc = fn(n, c.ctypes.data, rowsol.ctypes.data, colsol.ctypes.data, u.ctypes.data, v.ctypes.data,
            ctypes.addressof(tot), ctypes.addressof(tt), ctypes.addressof(st), int(max_rounds), int(stop_phase))
    if rc == 2:
        raise ValueError("cost matrix contains NaN/Inf")
    if rc != 0:
        raise RuntimeError(f"jv_oracle_wide failed with status {rc}")
    return dict(rowsol=rowsol, colsol=colsol, u=u, v=v, total=tot.value, total_T=tt.value, stats=st)
