"""Python face of the HIP Jonker-Volgenant solver (C ABI: cyto_lap_f32 / cyto_lap_f64).

`lapjv_hip(cost)` has the call shape of `lapjv.lapjv(cost)` as CytoSPACE uses it
(/root/reference/cytospace/linear_assignment_solvers/linear_assignment_solvers.py:38):
it returns the 3-tuple `(row_ind, col_ind, (total, u, v))`; CytoSPACE keeps `col_ind`
(`_, y, _ = solver(cost_scaled)`), where `col_ind[j]` is the row (spot slot) given to column
(cell) j.  It is a module-level function, hence picklable for ProcessPoolExecutor.submit
(/root/reference/cytospace/cytospace.py:446-451); libcytohip.so is loaded lazily in the
process that calls it.
"""
import ctypes

import numpy as np

from . import _lib


def lap_solve_rows(rows, rowmap, device_id=0, return_info=False, device_ptr=None, nu=None, ld=None, opts=None):
    """Solve the LAP whose row i is stored row rowmap[i] (C ABI: cyto_lap_f32_rowmap): `rows` holds every DISTINCT cost row
    once (nu x n, float32; or a device pointer with nu and ld), rowmap is what np.repeat(arange(nu), slots) gives.
    Same result, bit for bit, as lap_solve(rows[rowmap])."""
    L = _lib.lib()
    rowmap = np.ascontiguousarray(rowmap, dtype=np.int32)
    n = len(rowmap)
    if device_ptr is None:
        c = np.ascontiguousarray(rows, dtype=np.float32)
        if c.ndim != 2 or c.shape[1] != n:
            raise ValueError("rows must be nu x n with n = len(rowmap)")
        nu, ld, ptr, on_device = c.shape[0], n, c.ctypes.data, 0
    else:
        if nu is None:
            raise ValueError("nu is required with device_ptr")
        ld = n if ld is None else ld
        ptr, on_device = int(device_ptr), 1
    rowsol, colsol = np.empty(n, np.int32), np.empty(n, np.int32)
    u, v = np.empty(n, np.float32), np.empty(n, np.float32)
    total = ctypes.c_double()
    info = _lib.LapInfo()
    o = _lib.LapOpts(**opts) if opts else None
    st = L.cyto_lap_f32_rowmap(n, ptr, ld, int(nu), on_device, rowmap.ctypes.data, rowsol.ctypes.data, colsol.ctypes.data,
                               u.ctypes.data, v.ctypes.data, ctypes.byref(total), ctypes.byref(info), device_id, None,
                               ctypes.byref(o) if o is not None else None)
    _lib.check(st)
    out = dict(rowsol=rowsol, colsol=colsol, u=u, v=v, total=total.value)
    if return_info:
        out["info"] = info
    return out


def lap_solve(cost, dtype=np.float32, device_id=0, return_info=False, device_ptr=None, n=None, ld=None, opts=None):
    """Solve a square LAP on the GPU.

    cost        2-D square array-like (host) -- cast to `dtype` -- or None with `device_ptr`
    device_ptr  int address of a device-resident row-major matrix (then give n and ld)
    opts        None, or a dict of cyto_lap_opts fields (chain_variant, augmentation, no_handover, inject_exceptions):
                kernel selection only, the results never depend on it
    Returns dict(rowsol, colsol, u, v, total[, info]).
    """
    L = _lib.lib()
    if dtype not in (np.float32, np.float64):
        raise TypeError("dtype must be numpy.float32 or numpy.float64")
    narrow_on_device = False
    if device_ptr is None:
        c = np.asarray(cost)
        # the reference hands lapjv a float64 array that is solved in float32: narrow it on the device
        narrow_on_device = dtype == np.float32 and c.dtype == np.float64 and c.ndim == 2 and c.flags.c_contiguous
        if not narrow_on_device:
            c = np.ascontiguousarray(c, dtype=dtype)
        if c.ndim != 2 or c.shape[0] != c.shape[1]:
            raise ValueError("cost must be a square 2-D matrix")
        n = c.shape[0]
        ld = n
        ptr, on_device = c.ctypes.data, 0
        if n == 0:
            raise ValueError("cost must be non-empty")
    else:
        if n is None:
            raise ValueError("n is required with device_ptr")
        ld = n if ld is None else ld
        ptr, on_device = int(device_ptr), 1
    rowsol = np.empty(n, np.int32)
    colsol = np.empty(n, np.int32)
    u = np.empty(n, dtype)
    v = np.empty(n, dtype)
    total = ctypes.c_double()
    info = _lib.LapInfo()
    if opts:
        o = _lib.LapOpts(**opts)
        if narrow_on_device:
            c = np.ascontiguousarray(c, dtype=dtype)
            ptr = c.ctypes.data
        fn = L.cyto_lap_f32_opts if dtype == np.float32 else L.cyto_lap_f64_opts
        st = fn(n, ptr, ld, on_device, rowsol.ctypes.data, colsol.ctypes.data, u.ctypes.data, v.ctypes.data,
                ctypes.byref(total), ctypes.byref(info), device_id, None, ctypes.byref(o))
    elif narrow_on_device:
        st = L.cyto_lap_f32_from_f64(n, ptr, ld, rowsol.ctypes.data, colsol.ctypes.data, u.ctypes.data, v.ctypes.data,
                                     ctypes.byref(total), ctypes.byref(info), device_id, None)
    else:
        fn = L.cyto_lap_f32 if dtype == np.float32 else L.cyto_lap_f64
        st = fn(n, ptr, ld, on_device, rowsol.ctypes.data, colsol.ctypes.data, u.ctypes.data, v.ctypes.data,
                ctypes.byref(total), ctypes.byref(info), device_id, None)
    _lib.check(st)
    out = dict(rowsol=rowsol, colsol=colsol, u=u, v=v, total=total.value)
    if return_info:
        out["info"] = info
    return out


def lapjv_hip(cost, verbose=0, force_doubles=False):
    """Drop-in for `lapjv.lapjv`: returns (row_ind, col_ind, (total_cost, u, v)).

    Like lapjv 1.3.14 (recalled, unverified -- SURVEY.md section 8c) the solve runs in float32
    unless force_doubles=True.
    """
    r = lap_solve(cost, dtype=np.float64 if force_doubles else np.float32)
    return r["rowsol"], r["colsol"], (r["total"], r["u"], r["v"])


def lap_solve_batch(costs, device_id=0, max_concurrent=0, return_info=False, opts=None):
    """Solve several independent square LAPs concurrently on one GPU (C ABI: cyto_lap_batch_f32[_opts]).

    costs: list of 2-D float arrays (host).  opts: None or a dict of cyto_lap_opts fields for the whole batch.
    Returns a list of dicts like lap_solve()."""
    L = _lib.lib()
    nb = len(costs)
    if nb == 0:
        return []
    mats = [np.ascontiguousarray(c, dtype=np.float32) for c in costs]
    for c in mats:
        if c.ndim != 2 or c.shape[0] != c.shape[1] or c.shape[0] == 0:
            raise ValueError("every cost must be a non-empty square 2-D matrix")
    ns = (ctypes.c_int * nb)(*[c.shape[0] for c in mats])
    lds = (ctypes.c_int64 * nb)(*[c.shape[0] for c in mats])
    outs = [dict(rowsol=np.empty(c.shape[0], np.int32), colsol=np.empty(c.shape[0], np.int32),
                 u=np.empty(c.shape[0], np.float32), v=np.empty(c.shape[0], np.float32)) for c in mats]

    def ptrs(key):
        return (ctypes.c_void_p * nb)(*[o[key].ctypes.data for o in outs])

    cptr = (ctypes.c_void_p * nb)(*[c.ctypes.data for c in mats])
    totals = (ctypes.c_double * nb)()
    infos = (_lib.LapInfo * nb)()
    status = (ctypes.c_int * nb)()
    o = _lib.LapOpts(**opts) if opts else None
    st = L.cyto_lap_batch_f32_opts(nb, ns, cptr, lds, 0, ptrs("rowsol"), ptrs("colsol"), ptrs("u"), ptrs("v"), totals, infos,
                                   status, max_concurrent, device_id, ctypes.byref(o) if o is not None else None)
    _lib.check(st)
    for b, o in enumerate(outs):
        o["total"] = totals[b]
        if return_info:
            o["info"] = infos[b]
    return outs


def lap_solve_batch_device(device_ptrs, ns, lds=None, device_id=0, max_concurrent=0, return_info=False, opts=None):
    """lap_solve_batch for cost matrices that are already resident in HBM (row-major float32, ld elements per row)."""
    L = _lib.lib()
    nb = len(device_ptrs)
    if nb == 0:
        return []
    lds = list(ns) if lds is None else list(lds)
    n_arr = (ctypes.c_int * nb)(*[int(x) for x in ns])
    ld_arr = (ctypes.c_int64 * nb)(*[int(x) for x in lds])
    outs = [dict(rowsol=np.empty(n, np.int32), colsol=np.empty(n, np.int32), u=np.empty(n, np.float32),
                 v=np.empty(n, np.float32)) for n in ns]

    def ptrs(key):
        return (ctypes.c_void_p * nb)(*[o[key].ctypes.data for o in outs])

    cptr = (ctypes.c_void_p * nb)(*[int(p) for p in device_ptrs])
    totals = (ctypes.c_double * nb)()
    infos = (_lib.LapInfo * nb)()
    status = (ctypes.c_int * nb)()
    o = _lib.LapOpts(**opts) if opts else None
    st = L.cyto_lap_batch_f32_opts(nb, n_arr, cptr, ld_arr, 1, ptrs("rowsol"), ptrs("colsol"), ptrs("u"), ptrs("v"), totals, infos,
                                   status, max_concurrent, device_id, ctypes.byref(o) if o is not None else None)
    _lib.check(st)
    for b, o in enumerate(outs):
        o["total"] = totals[b]
        if return_info:
            o["info"] = infos[b]
    return outs
