"""Per-chunk solve and chunk fan-out, mirroring the reference's driver functions
(/root/reference/cytospace/cytospace.py:150-209, 304-351, 354-469) for the HIP path.

`solve_linear_assignment_problem` keeps the reference's signature; with solver_method == "lapjv_hip"
the whole chunk (cost build + LAP) stays on the GPU (C ABI: cyto_assign_pearson).
`assign_chunks` is the multi-chunk seam: independent square sub-LAPs, all chunks of a rank solved together (a workgroup
per chunk); across ranks (one process per GPU) the only collective is the broadcast of the transformed ST operand
(cytospace.py:430-451 ships the whole ST matrix to every worker of a process pool).
"""
import ctypes
import os
import threading
import time

import numpy as np

from . import _lib
from .linear_assignment_solvers import calculate_cost, call_solver, match_solution


def partition_indices(indices, split_by_category_list=None, split_by_interval_int=None, shuffle=True):
    """cytospace.py:150-209: split `indices` at category boundaries and every `interval` inside a
    category longer than the interval (np.array_split); optional in-place legacy shuffle."""
    indices = np.asarray(indices)
    num = len(indices)
    if shuffle:
        np.random.shuffle(indices)
    cuts = {0, num}
    if split_by_category_list is not None:
        if np.sum(split_by_category_list) != num:
            print('Warning: sum of counts in each category does not match the full length')
        cuts.update(int(b) for b in np.cumsum(split_by_category_list))
    base = sorted(cuts)
    if split_by_interval_int is not None:
        for lo, hi in zip(base[:-1], base[1:]):
            if hi - lo > split_by_interval_int:
                cuts.update(range(lo, hi, split_by_interval_int))
    return np.array_split(indices, sorted(cuts)[1:-1])


_DTYPE_CODE = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint16): 2, np.dtype(np.uint8): 3}   # CYTO_DTYPE_*


def _boundary_matrix(a):
    """A genes x columns matrix as the C ABI takes it: C-contiguous float32 / uint16 / uint8 as they are (raw counts: the narrower,
    the less crosses PCIe), anything else as float64 (the dtype of the reference's arrays).  Returns (array, CYTO_DTYPE_* code)."""
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("sc and st must be 2-D genes x columns matrices")
    if a.dtype not in _DTYPE_CODE:
        a = a.astype(np.float64)
    a = np.ascontiguousarray(a)
    return a, _DTYPE_CODE[a.dtype]


def _matrix_struct(a, code):
    m = _lib.Matrix()
    m.data, m.ld, m.is_f64, m.on_device = a.ctypes.data, a.shape[1], code, 0
    return m


def _pair(sc, st):
    """Both matrices as the boundary takes them (each in its own dtype)."""
    sc, csc = _boundary_matrix(sc)
    st, cst = _boundary_matrix(st)
    if sc.shape[0] != st.shape[0]:
        raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                         "ST and scRNA data must have the same genes")
    return sc, st, csc, cst


def assign_pearson(sc, st, slots, already_normalized=True, device_id=0, return_info=False,
                   distance_metric="Pearson_correlation"):
    """Fused chunk solve on one GPU: returns mapped_st_index (np.int64[C]) [, total, info]."""
    from .common import METRICS
    if distance_metric not in METRICS:
        raise ValueError(f"unknown distance_metric {distance_metric!r}")
    sc, st, csc, cst = _pair(sc, st)
    slots = np.ascontiguousarray(slots, dtype=np.int64)
    G, C = sc.shape
    S = st.shape[1]
    if len(slots) != S:
        raise ValueError("one slot count per spot is required")
    mapped = np.empty(C, np.int64)
    total = ctypes.c_double()
    info = _lib.AssignInfo()
    msc, mst = _matrix_struct(sc, csc), _matrix_struct(st, cst)
    _lib.check(_lib.lib().cyto_assign_metric_ex(METRICS[distance_metric], G, ctypes.byref(msc), C, ctypes.byref(mst), S,
                                                slots.ctypes.data, int(already_normalized), mapped.ctypes.data,
                                                ctypes.byref(total), ctypes.byref(info), device_id))
    if return_info:
        return mapped, total.value, info
    return mapped


class ExpressionContext:
    """Both expression matrices uploaded and transformed once (standardised / ranked / plain float32 operands of the
    metric), resident in HBM; every chunk of apply_linear_assignment gathers its columns out of them."""

    def __init__(self, sc, st, already_normalized=True, device_id=0, distance_metric="Pearson_correlation",
                 comm=None, root=0, n_spots=None):
        """comm (a _lib.Communicator, one process per GPU): only rank `root` passes the ST matrix (`st`; the others may pass
        None with n_spots); root transforms it once and the float32 operand reaches the other ranks with one RCCL broadcast.
        `sc` holds THIS rank's cells only."""
        from .common import METRICS
        if distance_metric not in METRICS:
            raise ValueError(f"unknown distance_metric {distance_metric!r}")
        self.bcast_ms = None
        self._h = ctypes.c_void_p()
        from . import common as _common
        if _common.is_sparse(sc) or (st is not None and _common.is_sparse(st)):
            self._create_from_mixed(sc, st, already_normalized, device_id, METRICS[distance_metric], comm, root, n_spots)
            return
        sc, csc = _boundary_matrix(sc)
        have_st = st is not None
        if have_st:
            sc, st, csc, cst = _pair(sc, st)
            n_spots = st.shape[1]
        elif comm is None:
            raise ValueError("the ST matrix is required without a communicator")
        else:
            # (neither a root without the ST matrix nor a rank without n_spots is rejected HERE: the other ranks are already on
            #  their way into the collective, so the call goes through -- with an invalid extent the library turns into its status --
            #  and the status word / the root's extents fail every rank together instead of leaving the others blocked)
            n_spots = (1 if comm.rank == root else 0) if n_spots is None else n_spots
        self.G, self.C = sc.shape
        self.S = int(n_spots)
        msc = _matrix_struct(sc, csc)
        mst = _matrix_struct(st, cst) if have_st else None
        ms = ctypes.c_double()
        _lib.check(_lib.lib().cyto_ctx_create_ex(METRICS[distance_metric], self.G, ctypes.byref(msc), self.C,
                                                 ctypes.byref(mst) if have_st else None, self.S, int(already_normalized),
                                                 comm.handle if comm is not None else None, int(root),
                                                 comm.rank if comm is not None else 0, device_id, ctypes.byref(self._h),
                                                 ctypes.byref(ms)))
        if comm is not None:
            self.bcast_ms = ms.value

    def _create_from_mixed(self, sc, st, already_normalized, device_id, metric, comm, root, n_spots):
        """Inputs of which at least one is a scipy.sparse matrix: sparse ones are uploaded as non-zeros and expanded on the
        device (cyto_csc_to_dense_f32), dense ones go up as they are (C ABI: cyto_ctx_create_ex)."""
        from . import common as _common
        keep = []

        def describe(x):
            m = _lib.Matrix()
            if _common.is_sparse(x):
                buf, G, C, ld = _common.sparse_to_device(x, device_id)
                keep.append(buf)
                m.data, m.ld, m.is_f64, m.on_device = buf.ptr, ld, 0, 1
                return m, G, C
            a, code = _boundary_matrix(x)
            keep.append(a)
            m.data, m.ld, m.is_f64, m.on_device = a.ctypes.data, a.shape[1], code, 0
            return m, a.shape[0], a.shape[1]
        msc, self.G, self.C = describe(sc)
        mst = None
        if st is not None:
            mst, Gs, self.S = describe(st)
            if Gs != self.G:
                raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                                 "ST and scRNA data must have the same genes")
        else:
            if comm is None or n_spots is None:
                raise ValueError("the ST matrix (or, on non-root ranks of a communicator, n_spots) is required")
            self.S = int(n_spots)
        ms = ctypes.c_double()
        _lib.check(_lib.lib().cyto_ctx_create_ex(metric, self.G, ctypes.byref(msc), self.C, ctypes.byref(mst) if mst is not None else None,
                                                 self.S, int(already_normalized), comm.handle if comm is not None else None, int(root),
                                                 comm.rank if comm is not None else 0, device_id, ctypes.byref(self._h), ctypes.byref(ms)))
        self.bcast_ms = ms.value if comm is not None else None
        for k in keep:
            if isinstance(k, _lib.DeviceBuffer):
                k.free()

    def assign_chunks(self, chunks, max_concurrent=0, return_info=False):
        """All chunks of this rank in one call (C ABI: cyto_ctx_assign_chunks): per chunk the gathers and the cost GEMM, then
        every chunk's LAP together, a workgroup per chunk.  chunks: list of (index_sc, slots[, index_st]).
        Returns a list of mapped_st_index arrays (or (mapped, total, info) tuples)."""
        if self._h is None:
            raise RuntimeError("context was closed")
        nb = len(chunks)
        if nb == 0:
            return []
        arr = (_lib.Chunk * nb)()
        keep = []
        for k, ch in enumerate(chunks):
            idx_sc = np.ascontiguousarray(ch[0], dtype=np.int64)
            slots = np.ascontiguousarray(ch[1], dtype=np.int64)
            idx_st = None if len(ch) < 3 or ch[2] is None else np.ascontiguousarray(ch[2], dtype=np.int64)
            nst = self.S if idx_st is None else len(idx_st)
            if len(slots) != nst:
                raise ValueError("one slot count per listed spot is required")
            mapped = np.empty(len(idx_sc), np.int64)
            keep.append((idx_sc, slots, idx_st, mapped))
            arr[k].idx_sc, arr[k].n_sc = idx_sc.ctypes.data, len(idx_sc)
            arr[k].idx_st, arr[k].n_st = (None if idx_st is None else idx_st.ctypes.data), nst
            arr[k].slots, arr[k].mapped_spot = slots.ctypes.data, mapped.ctypes.data
        st = _lib.lib().cyto_ctx_assign_chunks(self._h, nb, arr, int(max_concurrent))
        for k in range(nb):
            _lib.check(arr[k].status)
        _lib.check(st)
        if return_info:
            infos = []
            for k in range(nb):
                inf = _lib.AssignInfo()
                ctypes.memmove(ctypes.byref(inf), ctypes.byref(arr[k].info), ctypes.sizeof(inf))
                infos.append(inf)
            return [(keep[k][3], arr[k].total_cost, infos[k]) for k in range(nb)]
        return [keep[k][3] for k in range(nb)]

    def assign_chunk(self, index_sc, slots, index_st=None, return_info=False):
        """Cells index_sc against spots index_st (None: all spots) with slots[k] cells for the k-th listed spot.
        Returns mapped_st_index (np.int64, positions in the chunk's spot list) [, total, info]."""
        if self._h is None:
            raise RuntimeError("context was closed")
        idx_sc = np.ascontiguousarray(index_sc, dtype=np.int64)
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        idx_st = None if index_st is None else np.ascontiguousarray(index_st, dtype=np.int64)
        nst = self.S if idx_st is None else len(idx_st)
        if len(slots) != nst:
            raise ValueError("one slot count per listed spot is required")
        mapped = np.empty(len(idx_sc), np.int64)
        total = ctypes.c_double()
        info = _lib.AssignInfo()
        _lib.check(_lib.lib().cyto_ctx_assign_chunk(self._h, idx_sc.ctypes.data, len(idx_sc),
                                                    None if idx_st is None else idx_st.ctypes.data, nst, slots.ctypes.data,
                                                    mapped.ctypes.data, ctypes.byref(total), ctypes.byref(info)))
        if return_info:
            return mapped, total.value, info
        return mapped

    def close(self):
        if self._h is not None:
            _lib.lib().cyto_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_linear_assignment_problem(scRNA_norm_data, st_norm_data, cell_number_to_node_assignment,
                                    solver_method, solver, seed, distance_metric, process_idx=None, device_id=0):
    """cytospace.py:304-351.  Returns (mapped_st_index: list[int] of length C, process_idx).

    "lapjv_hip" (any of the three distance metrics): fused on the device (the +1e-16*rand tie-breaker of
    cytospace.py:325-327 is a no-op in float32 and is skipped).  Other shortest-augmenting-path solvers
    follow the reference's sequence with the device-built cost matrix."""
    if solver_method == "lapjv_hip" and distance_metric in ("Pearson_correlation", "Spearman_correlation", "Euclidean"):
        print('Solving linear assignment problem ...')
        t0 = time.perf_counter()
        mapped = assign_pearson(scRNA_norm_data, st_norm_data, cell_number_to_node_assignment,
                                already_normalized=True, device_id=device_id, distance_metric=distance_metric)
        print(f"Time to solve linear assignment problem: {round(time.perf_counter() - t0, 2)} seconds")
        return mapped.tolist(), process_idx
    if solver_method in ('lapjv', 'lapjv_compat', 'lapjv_hip'):
        distance_repeat, location_repeat = calculate_cost(scRNA_norm_data, st_norm_data, cell_number_to_node_assignment,
                                                          solver_method, distance_metric)
        print('Solving linear assignment problem ...')
        np.random.seed(seed)
        cost_scaled = distance_repeat + 1e-16 * np.random.rand(distance_repeat.shape[0], distance_repeat.shape[1])
        t0 = time.perf_counter()
        assignment = call_solver(solver, solver_method, cost_scaled)
        print(f"Time to solve linear assignment problem: {round(time.perf_counter() - t0, 2)} seconds")
        return np.transpose(location_repeat[assignment]).tolist(), process_idx
    if solver_method == 'lap_CSPR':
        # cytospace.py:334-347: integerised cost 10^6 d + 10 rand + 1 (legacy RandomState stream), workers = cells
        distance_repeat, location_repeat = calculate_cost(scRNA_norm_data, st_norm_data, cell_number_to_node_assignment,
                                                          solver_method, distance_metric)
        print('Solving linear assignment problem ...')
        np.random.seed(seed)
        cost_scaled = 10**6 * distance_repeat.astype(np.float64) + 10 * np.random.rand(*distance_repeat.shape) + 1
        cost_scaled_int = np.transpose(cost_scaled).astype(int)
        t0 = time.perf_counter()
        assignment = match_solution(cost_scaled_int)
        print(f"Time to solve linear assignment problem: {round(time.perf_counter() - t0, 2)} seconds")
        return location_repeat[assignment[:, 0].astype(int)].tolist(), process_idx
    raise ValueError("Invalid solver_method provided")


def _counts_matrix(a):
    """A genes x columns count matrix as the narrowest dtype that holds it exactly: uint8 / uint16 for integer counts that fit
    (a quarter / half of the float32 upload; the device widens them), float32 when every value survives the cast, else float64
    (the reference's dtype)."""
    a = np.asarray(a)
    if a.dtype in (np.float32, np.uint8, np.uint16):
        return np.ascontiguousarray(a)
    if a.dtype.kind in "iub":
        if a.size == 0:
            return np.ascontiguousarray(a, dtype=np.float32)
        lo, hi = int(a.min()), int(a.max())
        if lo >= 0 and hi < (1 << 8):
            return np.ascontiguousarray(a, dtype=np.uint8)
        if lo >= 0 and hi < (1 << 16):
            return np.ascontiguousarray(a, dtype=np.uint16)
        if max(-lo, hi) < (1 << 24):
            return np.ascontiguousarray(a, dtype=np.float32)
        return np.ascontiguousarray(a, dtype=np.float64)
    a64 = np.ascontiguousarray(a, dtype=np.float64)
    a32 = a64.astype(np.float32)
    return a32 if np.array_equal(a32, a64, equal_nan=True) else a64


class _RankContext:
    """ExpressionContext over only the columns this rank's chunks touch (a rank neither uploads nor transforms the cells of
    other ranks' chunks), with the chunk index lists remapped accordingly.  --sampling-sub-spots chunks (every chunk against
    ALL spots, cytospace.py:438) with a communicator: the ST matrix is transformed on rank 0 only and broadcast
    (ExpressionContext, comm=...).  --single-cell chunks (index_st_list: every chunk against its OWN spots, cytospace.py:434-435)
    need no collective at all: each rank uploads only the spots of its own chunks, the communicator is not used and
    every rank must hold the ST matrix."""

    def __init__(self, sc, st, mine, index_sc_list, index_st_list, device_id, distance_metric, already_normalized=True,
                 comm=None, n_spots=None):
        sc = np.asarray(sc)
        self._sc_cols = self._st_cols = None
        self.ctx = None
        if index_st_list is not None:
            comm = None                               # (decided by the MODE, the same on every rank: nobody enters a collective)
            if st is None and mine:
                raise ValueError("--single-cell chunks take their spots from the ST matrix of the rank that solves them: "
                                 "every rank needs st (no broadcast in this mode)")
        if not mine and comm is None:
            return                                    # nothing to solve on this rank: no upload, no transform
        if mine and len(mine) < len(index_sc_list):
            cols = np.unique(np.concatenate([np.asarray(index_sc_list[i]) for i in mine]))
            if len(cols) < sc.shape[1]:
                self._sc_cols, sc = cols, sc[:, cols]
            if index_st_list is not None and comm is None:
                st = np.asarray(st)
                cols = np.unique(np.concatenate([np.asarray(index_st_list[i]) for i in mine]))
                if len(cols) < st.shape[1]:
                    self._st_cols, st = cols, st[:, cols]
        elif not mine:
            sc = sc[:, :1]                            # (takes part in the broadcast only)
        self.ctx = ExpressionContext(sc, st, already_normalized, device_id, distance_metric, comm=comm, n_spots=n_spots)

    def remap(self, index_sc, index_st=None):
        index_sc = np.asarray(index_sc)
        if self._sc_cols is not None:
            index_sc = np.searchsorted(self._sc_cols, index_sc)
        if index_st is not None and self._st_cols is not None:
            index_st = np.searchsorted(self._st_cols, np.asarray(index_st))
        return index_sc, index_st

    def assign_chunks(self, chunks, max_concurrent=0):
        """chunks: list of (index_sc, slots, index_st or None) in ORIGINAL column numbering."""
        if not chunks:
            return []
        out = []
        for index_sc, slots, index_st in chunks:
            isc, ist = self.remap(index_sc, index_st)
            out.append((isc, slots, ist))
        return self.ctx.assign_chunks(out, max_concurrent)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.close()


def schedule_chunks(sizes, n_devices):
    """Longest-processing-time-first placement of independent sub-LAPs on devices (cost ~ n^2.5)."""
    order = np.argsort([-float(s) ** 2.5 for s in sizes], kind="stable")
    load = [0.0] * n_devices
    owner = [0] * len(sizes)
    for idx in order:
        d = int(np.argmin(load))
        owner[idx] = d
        load[d] += float(sizes[idx]) ** 2.5
    return owner


def assign_chunks(scRNA, st, cell_number_to_node_assignment, index_sc_list, index_st_list=None,
                  subsampled_slots_list=None, rank=0, world_size=1, device_id=0, max_concurrent=0,
                  distance_metric="Pearson_correlation", already_normalized=True, comm=None):
    """The chunk fan-out of apply_linear_assignment (cytospace.py:405-467) for one rank of a
    one-process-per-GPU job: this rank solves the chunks the LPT schedule gives it -- all of them in one batched call, a
    workgroup per chunk in every chain phase -- and returns {chunk index: mapped_st_index}.
    scRNA / st: genes x cells / genes x spots numpy arrays, normalised (already_normalized=True, as the reference hands them
    to its workers) or raw counts (False: normalised on the device).  comm: a _lib.Communicator; then only rank 0 needs `st`
    (the others may pass None) and its transformed operand is broadcast over xGMI."""
    if (index_st_list is not None) and (subsampled_slots_list is not None):
        raise ValueError("index_st_list and subsampled_cell_number_to_node_assignment_list cannot both be specified")
    n_chunks = len(index_sc_list)
    owner = schedule_chunks([len(ix) for ix in index_sc_list], world_size)
    mine = [idx for idx in range(n_chunks) if owner[idx] == rank]
    slots_all = None if cell_number_to_node_assignment is None else np.asarray(cell_number_to_node_assignment)
    n_spots = np.asarray(st).shape[1] if st is not None else (len(slots_all) if slots_all is not None
                                                              else len(subsampled_slots_list[0]))
    with _RankContext(scRNA, st, mine, index_sc_list, index_st_list, device_id, distance_metric, already_normalized,
                      comm=comm, n_spots=n_spots) as ctx:
        chunks = []
        for idx in mine:
            if index_st_list is not None:
                chunks.append((index_sc_list[idx], slots_all[np.asarray(index_st_list[idx])], index_st_list[idx]))
            elif subsampled_slots_list is not None:
                chunks.append((index_sc_list[idx], subsampled_slots_list[idx], None))
            else:
                chunks.append((index_sc_list[idx], slots_all, None))
        return dict(zip(mine, ctx.assign_chunks(chunks, max_concurrent)))


def visible_devices(devices=None):
    """The (logical) device list the chunk fan-out runs on: `devices` if given, else the environment variable
    CYTOSPACE_HIP_DEVICES ("0,1,2,3"; a device may be listed more than once: several workers on one GPU), else every HIP
    device this process sees.  Raises ValueError for a device that does not exist."""
    nvis = _lib.device_count()
    if devices is None:
        env = os.environ.get("CYTOSPACE_HIP_DEVICES", "").strip()
        devices = [int(x) for x in env.split(",") if x.strip() != ""] if env else list(range(nvis))
    devices = [int(d) for d in devices]
    if nvis < 1:
        raise _lib.CytoHipError("no HIP device visible (there is no CPU fallback)")
    bad = [d for d in devices if d < 0 or d >= nvis]
    if bad or not devices:
        raise ValueError(f"devices {devices}: this process sees {nvis} HIP device(s)")
    return devices


def assign_chunks_on_devices(scRNA, st, cell_number_to_node_assignment, index_sc_list, index_st_list=None,
                             subsampled_slots_list=None, devices=None, max_concurrent=0,
                             distance_metric="Pearson_correlation", already_normalized=True):
    """The chunk fan-out of apply_linear_assignment over EVERY device of this process -- what the reference's
    ProcessPoolExecutor(number_of_processors) (cytospace.py:430-451) becomes on a multi-GPU node, with no launcher: one host
    thread per (logical) device, chunk -> device by the LPT schedule, all chunks of a device in one batched call.
    --sampling-sub-spots chunks share the ST operand: device 0 transforms it ONCE and it reaches the others with one
    broadcast (RCCL over xGMI between distinct devices; a device-to-device copy between logical ranks of one device);
    the other devices never see the host ST matrix.  Returns {chunk index: mapped_st_index}, every chunk."""
    devices = visible_devices(devices)
    n_chunks = len(index_sc_list)
    W = max(1, min(len(devices), n_chunks))
    devices = devices[:W]
    if W == 1:
        return assign_chunks(scRNA, st, cell_number_to_node_assignment, index_sc_list, index_st_list, subsampled_slots_list,
                             rank=0, world_size=1, device_id=devices[0], max_concurrent=max_concurrent,
                             distance_metric=distance_metric, already_normalized=already_normalized)
    # what every rank would reject is rejected HERE, before any thread can wait for a peer that never arrives
    if (index_st_list is not None) and (subsampled_slots_list is not None):
        raise ValueError("index_st_list and subsampled_cell_number_to_node_assignment_list cannot both be specified")
    from .common import METRICS
    if distance_metric not in METRICS:
        raise ValueError(f"unknown distance_metric {distance_metric!r}")
    if st is None or np.asarray(scRNA).shape[0] != np.asarray(st).shape[0]:
        raise ValueError("The two matrices v1 and v2 must have equal dimensions; "
                         "ST and scRNA data must have the same genes")
    shared_st = index_st_list is None                       # sub-spot mode: one ST operand for everybody
    comms = _lib.Communicator.init_local(devices) if shared_st else [None] * W
    results, errors = [None] * W, [None] * W

    def work(r):
        try:
            results[r] = assign_chunks(scRNA, st if (r == 0 or not shared_st) else None, cell_number_to_node_assignment,
                                       index_sc_list, index_st_list, subsampled_slots_list, rank=r, world_size=W,
                                       device_id=devices[r], max_concurrent=max_concurrent, distance_metric=distance_metric,
                                       already_normalized=already_normalized, comm=comms[r])
        except BaseException as e:      # noqa: BLE001 (re-raised on the calling thread)
            errors[r] = e
            # peers waiting in a collective this rank will not reach must fail instead of hanging: EVERY communicator of the
            # process is aborted (the in-process kind wakes its group; the RCCL kind needs ncclCommAbort on each peer's own
            # communicator -- aborting only this rank's leaves the others inside ncclBroadcast / ncclAllReduce for ever)
            for c in comms:
                if c is not None:
                    try:
                        c.abort()
                    except Exception:   # noqa: BLE001
                        pass
    threads = [threading.Thread(target=work, args=(r,), name=f"cytohip-dev{devices[r]}-rank{r}", daemon=True) for r in range(W)]
    for t in threads:
        t.start()
    # join with a deadline after the FIRST failure: a healthy run may take as long as it takes, but once a rank has failed its
    # peers have been aborted and must return promptly -- if one does not (a collective that ignores the abort), raise instead of
    # hanging the caller (the stuck thread is a daemon; its communicator is leaked on purpose: destroying it could block too)
    deadline = None
    grace = float(os.environ.get("CYTOSPACE_HIP_ABORT_GRACE_S", "60"))
    stuck = []
    for t in threads:
        while t.is_alive():
            if deadline is None and any(e is not None for e in errors):
                deadline = time.monotonic() + grace
            t.join(0.05 if deadline is None else max(0.0, min(0.05, deadline - time.monotonic())))
            if deadline is not None and time.monotonic() >= deadline and t.is_alive():
                stuck.append(t.name)
                break
    for r, c in enumerate(comms):
        if c is not None and threads[r].name not in stuck:
            c.close()
    if stuck:
        first = next(e for e in errors if e is not None)
        raise _lib.CytoHipError(f"rank thread(s) {stuck} did not return within {grace:.0f} s of a peer's failure ({first!r})") from first
    real = [e for e in errors if e is not None and f"status {_lib.CYTO_ERR_PEER}" not in str(e)]
    if real or any(e is not None for e in errors):
        raise (real[0] if real else next(e for e in errors if e is not None))
    merged = {}
    for r in range(W):
        merged.update(results[r])
    return merged


def apply_linear_assignment(scRNA_data, st_data, coordinates_data, cell_number_to_node_assignment,
                            solver_method, solver, seed, distance_metric, number_of_processors,
                            index_sc_list, index_st_list=None, subsampled_cell_number_to_node_assignment_list=None,
                            rank=0, world_size=1, device_id=None, comm=None, devices=None):
    """cytospace/cytospace.py:354-469 with the reference's arguments (pandas DataFrames as read by read_data):
    normalise once, solve every chunk, map the assigned spot indices to coordinates.

    Returns (assigned_locations: pd.DataFrame, cell_ids_selected: np.ndarray); the nth cell id is mapped to the
    nth row of assigned_locations.  The count matrices go to the GPU ONCE, as raw counts (float32 when exact), and are
    normalised and transformed there (common.py:142-147 on the device); nothing normalised comes back to the host.

    With the reference's own arguments (nothing after subsampled_...) the chunks are scheduled over EVERY HIP device this
    process sees (`devices`, or CYTOSPACE_HIP_DEVICES; assign_chunks_on_devices): the reference forks one process per chunk,
    here one host thread drives each device and all chunks of a device go through the solver together
    (`number_of_processors` bounds how many are in flight per device); results come back in chunk (submission) order -- the
    reference concatenates in completion order: compare as a set of (cell, spot) pairs.
    One process per GPU under an external launcher instead: pass rank / world_size / device_id (and comm: then only rank 0's
    ST matrix is used and its operand is broadcast); each rank then returns the chunks the LPT schedule gives it."""
    import pandas as pd
    if (index_st_list is not None) and (subsampled_cell_number_to_node_assignment_list is not None):
        raise ValueError("index_st_list and subsampled_cell_number_to_node_assignment_list cannot both be specified")
    if solver_method != "lapjv_hip":
        raise ValueError("apply_linear_assignment of this package drives the lapjv_hip solver")
    sc_counts = _counts_matrix(scRNA_data.to_numpy())
    st_counts = _counts_matrix(st_data.to_numpy()) if st_data is not None else None
    cell_ids = scRNA_data.columns.values
    slots_all = np.asarray(cell_number_to_node_assignment)
    launcher = world_size > 1 or comm is not None or device_id is not None       # the caller places this rank itself
    if (index_st_list is None) and (subsampled_cell_number_to_node_assignment_list is None):
        print('Solving linear assignment problem ...')
        t0 = time.perf_counter()
        one_dev = (0 if device_id is None else device_id) if launcher else visible_devices(devices)[0]   # one LAP: one GPU
        mapped = assign_pearson(sc_counts[:, index_sc_list[0]], st_counts, slots_all, already_normalized=False,
                                device_id=one_dev, distance_metric=distance_metric)
        print(f"Time to solve linear assignment problem: {round(time.perf_counter() - t0, 2)} seconds")
        return coordinates_data.iloc[mapped.tolist()], cell_ids[index_sc_list[0]]
    n_chunks = len(index_st_list) if index_st_list is not None else len(subsampled_cell_number_to_node_assignment_list)
    print(f"Number of required processors: {n_chunks}")
    if launcher:
        res = assign_chunks(sc_counts, st_counts, slots_all, index_sc_list, index_st_list,
                            subsampled_cell_number_to_node_assignment_list, rank=rank, world_size=world_size,
                            device_id=0 if device_id is None else device_id, max_concurrent=int(number_of_processors),
                            distance_metric=distance_metric, already_normalized=False, comm=comm)
    else:
        res = assign_chunks_on_devices(sc_counts, st_counts, slots_all, index_sc_list, index_st_list,
                                       subsampled_cell_number_to_node_assignment_list, devices=devices,
                                       max_concurrent=int(number_of_processors), distance_metric=distance_metric,
                                       already_normalized=False)
    assigned_locations_list, cell_ids_selected_list = [], []
    for idx in sorted(res):
        mapped = res[idx]
        loc = coordinates_data.iloc[index_st_list[idx]].iloc[mapped] if index_st_list is not None \
            else coordinates_data.iloc[mapped]
        assigned_locations_list.append(loc)
        cell_ids_selected_list.append(cell_ids[index_sc_list[idx]])
    if not res:
        return coordinates_data.iloc[[]], cell_ids[[]]
    return pd.concat(assigned_locations_list), np.concatenate(cell_ids_selected_list, axis=0)


# ---- upstream of the chunk fan-out: the per-spot cell counts, the per-type cell numbers and the
# cell sampler whose output apply_linear_assignment consumes.  Host code like the reference's; the one heavy step of
# the first (normalize_data over the whole ST matrix) runs on the device. ----

def estimate_cell_number_RNA_reads(st_data, mean_cell_numbers, device_id=0):
    """cytospace/cytospace.py:116-134.  Cells per spot from a straight line through (min, 0 or 1) and
    (mean, mean_cell_numbers) of the per-spot sums of the normalised expression; truncated to int."""
    from . import common
    reads = common.normalize_data(st_data.values.astype(float), device_id).sum(axis=0, dtype=float)
    lo, mid = reads.min(), reads.mean()
    line = np.poly1d(np.polyfit(np.array([lo, mid]), np.array([1 if lo > 0 else 0, mean_cell_numbers]), 1))
    return line(reads).astype(int)


def get_cell_type_fraction(number_of_cells, cell_type_fraction_data):
    """cytospace/cytospace.py:137-147.  Fractions (1 x types) -> integer cell numbers (types x 1): truncate
    fraction * number_of_cells, then give the shortfall to the FIRST cell type."""
    numbers = cell_type_fraction_data.transpose()
    numbers.iloc[:, 0] = (numbers.values * number_of_cells).astype(int)[:, 0]
    numbers.loc[numbers.index[0], numbers.columns[0]] += number_of_cells - sum(numbers.iloc[:, 0])
    return numbers


def sample_single_cells(scRNA_data, cell_type_data, cell_type_numbers_int, sampling_method, seed):
    """cytospace/cytospace.py:212-301.  Draw, per cell type and in the order of cell_type_numbers_int.index, as many
    cells as that type needs.  Enough cells: python `random.sample` without replacement; too few: every cell once plus
    `np.random.choice` with replacement ("duplicates") or synthetic cells whose genes are drawn independently from
    the type's cells ("place_holders").  Both generators are seeded with `seed` first, and are consumed in the
    reference's order, so the same seed reproduces the reference's sample."""
    import random
    import pandas as pd
    if sampling_method not in ("duplicates", "place_holders"):
        raise ValueError("Invalid sampling_method provided")
    np.random.seed(seed)
    random.seed(seed)
    labels = cell_type_data.values[:, 0]
    ids = scRNA_data.columns.values
    picked, blocks, names = [], [], []
    for cell_type in cell_type_numbers_int.index.values:
        members = np.nonzero(labels == cell_type)[0].tolist()
        if not members:
            raise ValueError(f"Cell type {cell_type} in the ST dataset is not available in the scRNA-seq dataset.")
        want = cell_type_numbers_int.loc[cell_type].iloc[0]
        short = want - len(members)
        if short <= 0:
            chosen = random.sample(members, want)
            if sampling_method == "duplicates":
                picked.append(chosen)
            else:
                names.append(ids[chosen]); blocks.append(scRNA_data.iloc[:, chosen].to_numpy())
        elif sampling_method == "duplicates":
            picked.append(np.concatenate([members, np.random.choice(members, short)], axis=0))
        else:
            own = scRNA_data.iloc[:, members].to_numpy()
            np.random.choice(members, short)          # (the reference draws and discards this)
            fake = np.zeros((own.shape[0], short))
            for k in range(short):
                fake[:, k] = [np.random.choice(own[g, :]) for g in range(own.shape[0])]
            names.append(np.array(ids[members]))
            names.append(np.array([cell_type.replace('TYPE_', 'CELL_') + '_new_' + str(k + 1) for k in range(short)]))
            blocks.append(own); blocks.append(fake)
    if sampling_method == "place_holders":
        return pd.DataFrame(np.concatenate(blocks, axis=1), index=scRNA_data.index, columns=np.concatenate(names, axis=0))
    return scRNA_data.iloc[:, np.concatenate(picked, axis=0).astype(int)]
