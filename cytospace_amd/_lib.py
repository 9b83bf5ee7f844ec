"""ctypes loader for libcytohip.so -- the only bridge between the Python host code and HIP.

Loading is lazy and per-process: nothing touches HIP at import time, so the solver callable
stays picklable and fork-safe (the reference ships it through ProcessPoolExecutor.submit,
/root/reference/cytospace/cytospace.py:446-451).  There is NO fallback: if the library is
missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CYTOHIP_LIB") or os.path.join(_HERE, "libcytohip.so")     # (CYTOHIP_LIB: a developer's A/B build)
_lib = None

CYTO_OK = 0
_EXC = {1: ValueError, 2: ValueError, 3: MemoryError, 7: ValueError, 8: ValueError}
CYTO_ERR_PEER = 9        # another rank of the communicator failed (raised as CytoHipError)


class CytoHipError(RuntimeError):
    pass


class LapInfo(ctypes.Structure):
    _fields_ = [("ms_colred", ctypes.c_double), ("ms_cache", ctypes.c_double), ("ms_chain", ctypes.c_double),
                ("ms_total", ctypes.c_double)] + \
        [(k, ctypes.c_int64) for k in (
            "scans_colred", "scans_redtransfer", "scans_arr", "scans_aug_init", "scans_aug_relax",
            "augmentations", "path_hops", "free_after_colred", "free_after_arr1", "free_after_arr2",
            "hbm_row_reads", "dense_refreshes")] + [("ms_arr", ctypes.c_double), ("ms_aug", ctypes.c_double),
                                                        ("aug_scans_skipped", ctypes.c_int64),
                                                        ("row_groups", ctypes.c_int64), ("aug_dense_scans", ctypes.c_int64),
                                                        ("aug_sparse_inits", ctypes.c_int64), ("aug_handover", ctypes.c_int64)] + \
        [(k, ctypes.c_int64) for k in ("wide", "wide_rounds", "wide_retired", "wide_dense_arr", "wide_dense_aug", "wide_aug_rounds",
                                       "wide_aug_settled", "wide_trivial", "wide_verify_passes", "wide_list_rounds", "wide_chain_rounds")] + \
        [(k, ctypes.c_double) for k in ("wide_ms_list", "wide_ms_chain", "wide_ms_aug_rounds", "wide_ms_aug_verify", "wide_ms_aug_finish",
                                        "wide_ms_aug_trivial")] + [("wide_arr_launches", ctypes.c_int64), ("wide_aug_launches", ctypes.c_int64),
         ("wide_scaled", ctypes.c_int64), ("wide_phases", ctypes.c_int64), ("wide_par_batches", ctypes.c_int64),
         ("wide_par_discarded", ctypes.c_int64), ("f64_warm", ctypes.c_int64), ("f64_warm_ms", ctypes.c_double),
         ("certified", ctypes.c_int64), ("gap_f64", ctypes.c_double), ("gap_max_f64", ctypes.c_double), ("gap_rows", ctypes.c_int64),
         ("polished", ctypes.c_int64), ("polish_ms", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}

    @property
    def row_scans(self):
        return int(self.scans_colred + self.scans_redtransfer + self.scans_arr
                   + self.scans_aug_init + self.scans_aug_relax)


class LapOpts(ctypes.Structure):
    """cyto_lap_opts (include/cytohip.h): kernel-selection options; results never depend on them."""
    _fields_ = [("chain_variant", ctypes.c_int32), ("augmentation", ctypes.c_int32), ("no_handover", ctypes.c_int32),
                ("inject_exceptions", ctypes.c_int32), ("group_state_global", ctypes.c_int32), ("aux_state_global", ctypes.c_int32), ("mode", ctypes.c_int32), ("wide_rounds", ctypes.c_int32), ("wide_groups", ctypes.c_int32), ("wide_rebuild", ctypes.c_int32),
                ("wide_par", ctypes.c_int32), ("wide_wipe", ctypes.c_int32),
                ("cache_waves", ctypes.c_int32), ("cache_unroll", ctypes.c_int32), ("cache_stream", ctypes.c_int32),
                ("certify", ctypes.c_int32), ("polish", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3)]


class AssignInfo(ctypes.Structure):
    _fields_ = [("ms_standardize", ctypes.c_double), ("ms_gemm", ctypes.c_double), ("gemm_flops", ctypes.c_double),
                ("lap", LapInfo)]


class Matrix(ctypes.Structure):
    """cyto_matrix (include/cytohip.h): a dense genes x columns array on the host or on the device."""
    _fields_ = [("data", ctypes.c_void_p), ("ld", ctypes.c_int64), ("is_f64", ctypes.c_int32), ("on_device", ctypes.c_int32)]


class Chunk(ctypes.Structure):
    """cyto_chunk (include/cytohip.h)."""
    _fields_ = [("idx_sc", ctypes.c_void_p), ("n_sc", ctypes.c_int32), ("idx_st", ctypes.c_void_p), ("n_st", ctypes.c_int32),
                ("slots", ctypes.c_void_p), ("mapped_spot", ctypes.c_void_p), ("total_cost", ctypes.c_double),
                ("status", ctypes.c_int32), ("info", AssignInfo)]


def lib():
    """Return the loaded library (loading it on first use in this process)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CytoHipError(
                f"{LIB_PATH} not found: build it with `python -m cytospace_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # Independent chunk solves run on separate HIP streams; the runtime's default of 4 hardware queues
        # would serialise them (measured: 32 concurrent 10k x 10k LAPs 3.6 s -> 1.06 s with 64 queues).
        # Must be set before the HIP runtime initialises, i.e. before the first call into the library.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
        L = ctypes.CDLL(LIB_PATH)
        L.cyto_strerror.restype = ctypes.c_char_p
        L.cyto_strerror.argtypes = [ctypes.c_int]
        L.cyto_last_hip_error.restype = ctypes.c_char_p
        L.cyto_version.restype = ctypes.c_char_p
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.cyto_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
        L.cyto_device_name.argtypes = [i32, ctypes.c_char_p, ctypes.c_size_t]
        L.cyto_malloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t, i32]
        L.cyto_free.argtypes = [vp, i32]
        L.cyto_memcpy_h2d.argtypes = [vp, vp, ctypes.c_size_t, i32]
        L.cyto_memcpy_d2h.argtypes = [vp, vp, ctypes.c_size_t, i32]
        L.cyto_device_synchronize.argtypes = [i32]
        for name in ("cyto_lap_f32", "cyto_lap_f64"):
            getattr(L, name).argtypes = [i32, vp, i64, i32, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(LapInfo), i32, vp]
        for name in ("cyto_lap_f32_opts", "cyto_lap_f64_opts"):
            getattr(L, name).argtypes = [i32, vp, i64, i32, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(LapInfo), i32, vp, ctypes.POINTER(LapOpts)]
            getattr(L, name).restype = ctypes.c_int
        L.cyto_lap_f32_rowmap.argtypes = [i32, vp, i64, i32, i32, vp, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(LapInfo), i32, vp, ctypes.POINTER(LapOpts)]
        L.cyto_lap_f32_rowmap.restype = ctypes.c_int
        L.cyto_trim_device_cache.argtypes = [i32]
        L.cyto_trim_device_cache.restype = ctypes.c_int
        dp = ctypes.POINTER(ctypes.c_double)
        L.cyto_lap_f32_from_f64.argtypes = [i32, vp, i64, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(LapInfo), i32, vp]
        L.cyto_normalize_data.argtypes = [i32, i32, vp, i64, i32, vp, i64, i32]
        L.cyto_standardize.argtypes = [i32, i32, vp, i64, i32, i32, i32, vp, i64, i32, i32, vp]
        L.cyto_cost_pearson.argtypes = [i32, i32, i32, vp, i64, vp, i64, vp, vp, i64, dp, i32, vp]
        L.cyto_transform.argtypes = [i32, i32, i32, vp, i64, i32, i32, i32, vp, i64, i32, i32, vp]
        L.cyto_cost_metric.argtypes = [i32, i32, i32, i32, vp, i64, vp, i64, vp, vp, i64, dp, i32, vp]
        L.cyto_assign_metric.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, vp, dp, ctypes.POINTER(AssignInfo), i32]
        L.cyto_assign_metric_ex.argtypes = [i32, i32, ctypes.POINTER(Matrix), i32, ctypes.POINTER(Matrix), i32, vp, i32, vp, dp,
                                            ctypes.POINTER(AssignInfo), i32]
        L.cyto_assign_metric_ex.restype = ctypes.c_int
        L.cyto_ctx_create.argtypes = [i32, i32, i32, i32, vp, vp, i32, i32, ctypes.POINTER(vp)]
        L.cyto_ctx_assign_chunk.argtypes = [vp, vp, i32, vp, i32, vp, vp, dp, ctypes.POINTER(AssignInfo)]
        L.cyto_ctx_destroy.argtypes = [vp]
        L.cyto_assign_metric_typed.argtypes = [i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, dp, ctypes.POINTER(AssignInfo), i32]
        L.cyto_ctx_create_typed.argtypes = [i32, i32, i32, i32, vp, vp, i32, i32, i32, ctypes.POINTER(vp)]
        L.cyto_ctx_destroy.restype = None
        L.cyto_ctx_create_shared.argtypes = [i32, i32, i32, i32, vp, vp, i32, i32, vp, i32, i32, i32, ctypes.POINTER(vp), dp]
        L.cyto_ctx_create_shared.restype = ctypes.c_int
        L.cyto_ctx_create_ex.argtypes = [i32, i32, ctypes.POINTER(Matrix), i32, ctypes.POINTER(Matrix), i32, i32, vp, i32, i32, i32,
                                         ctypes.POINTER(vp), dp]
        L.cyto_ctx_create_ex.restype = ctypes.c_int
        L.cyto_csc_to_dense_f32.argtypes = [i32, i32, i64, vp, vp, vp, vp, i64, i32, vp]
        L.cyto_csc_to_dense_f32.restype = ctypes.c_int
        L.cyto_ctx_assign_chunks.argtypes = [vp, i32, ctypes.POINTER(Chunk), i32]
        L.cyto_ctx_assign_chunks.restype = ctypes.c_int
        L.cyto_memcpy_d2d.argtypes = [vp, vp, ctypes.c_size_t, i32]
        L.cyto_memcpy_d2d.restype = ctypes.c_int
        L.cyto_assign_pearson.argtypes = [i32, i32, i32, vp, vp, vp, i32, vp, dp, ctypes.POINTER(AssignInfo), i32]
        L.cyto_lap_batch_f32.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32]
        L.cyto_lap_batch_f32_opts.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, ctypes.POINTER(LapOpts)]
        L.cyto_lap_batch_f32_opts.restype = ctypes.c_int
        L.cyto_comm_unique_id.argtypes = [ctypes.c_char_p]
        L.cyto_comm_init.argtypes = [ctypes.c_char_p, i32, i32, i32, ctypes.POINTER(vp)]
        L.cyto_comm_init_local.argtypes = [i32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp)]
        L.cyto_comm_count.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        L.cyto_comm_kind.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        L.cyto_comm_agree.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        L.cyto_comm_abort.argtypes = [vp]
        L.cyto_comm_aborted.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
        L.cyto_comm_bcast_f32.argtypes = [vp, vp, ctypes.c_size_t, i32, i32, vp]
        L.cyto_comm_destroy.argtypes = [vp]
        for name in ("cyto_comm_init_local", "cyto_comm_count", "cyto_comm_kind", "cyto_comm_agree", "cyto_comm_abort", "cyto_comm_aborted"):
            getattr(L, name).restype = ctypes.c_int
        # ABI check: the structs mirrored above must be the library's (include/cytohip.h carries no size members)
        sz = [ctypes.c_size_t() for _ in range(4)]
        L.cyto_abi_sizes.argtypes = [ctypes.POINTER(ctypes.c_size_t)] * 4
        L.cyto_abi_sizes(*[ctypes.byref(x) for x in sz])
        mine = [ctypes.sizeof(t) for t in (LapInfo, LapOpts, AssignInfo, Chunk)]
        if [x.value for x in sz] != mine:
            raise CytoHipError(f"{LIB_PATH} was built from another include/cytohip.h: struct sizes {[x.value for x in sz]} != {mine} "
                               "(rebuild: python -m cytospace_amd.build --force)")
        for name in ("cyto_normalize_data", "cyto_standardize", "cyto_cost_pearson", "cyto_assign_pearson",
                     "cyto_lap_batch_f32", "cyto_lap_batch_f32_opts", "cyto_comm_unique_id", "cyto_comm_init", "cyto_comm_bcast_f32",
                     "cyto_comm_destroy"):
            getattr(L, name).restype = ctypes.c_int
        for name in ("cyto_device_count", "cyto_device_name", "cyto_malloc", "cyto_free", "cyto_memcpy_h2d",
                     "cyto_memcpy_d2h", "cyto_device_synchronize", "cyto_lap_f32", "cyto_lap_f64"):
            getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


def check(status):
    """Turn a C status code into the Python exception the reference's path would raise."""
    if status == CYTO_OK:
        return
    L = lib()
    msg = L.cyto_strerror(status).decode()
    if status == 5:
        msg += ": " + L.cyto_last_hip_error().decode()
    raise _EXC.get(status, CytoHipError)(f"cytohip status {status}: {msg}")


def device_count():
    n = ctypes.c_int(0)
    st = lib().cyto_device_count(ctypes.byref(n))
    return n.value if st == CYTO_OK else 0


class DeviceBuffer:
    """A caller-owned HBM allocation (so a cost matrix can stay resident across solves)."""

    def __init__(self, nbytes, device_id=0):
        self.device_id = device_id
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(lib().cyto_malloc(ctypes.byref(p), self.nbytes, device_id))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr, device_id=0):
        import numpy as np
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes, device_id)
        check(lib().cyto_memcpy_h2d(buf.ptr, arr.ctypes.data, arr.nbytes, device_id))
        return buf

    def clone(self):
        """A device-to-device copy (same device)."""
        out = DeviceBuffer(self.nbytes, self.device_id)
        check(lib().cyto_memcpy_d2d(out.ptr, self.ptr, self.nbytes, self.device_id))
        return out

    def to_numpy(self, shape, dtype):
        import numpy as np
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib().cyto_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, self.device_id))
        return out

    def free(self):
        if self.ptr:
            check(lib().cyto_free(self.ptr, self.device_id))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Communicator:
    """The communicator of the chunk fan-out (C ABI: cyto_comm_*; csrc/comm.hip).  One process per GPU: the 128-byte RCCL
    unique id is made on rank 0 (`Communicator.unique_id()`) and handed to the other ranks by the launcher
    (cytospace_amd.rendezvous.FileStore, MPI, a file ...).  One process, one thread per device: `Communicator.init_local`."""

    def __init__(self, unique_id, rank, nranks, device_id=0, _handle=None):
        self.rank, self.nranks, self.device_id = int(rank), int(nranks), int(device_id)
        if _handle is not None:
            self._h = _handle
            return
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(lib().cyto_comm_init(buf, self.rank, self.nranks, self.device_id, ctypes.byref(self._h)))

    @classmethod
    def init_local(cls, device_ids):
        """One communicator per entry of device_ids, for the host threads of THIS process (rank r drives device_ids[r]).
        Distinct devices: RCCL (ncclCommInitAll).  A device listed twice (logical ranks): the in-process kind, whose
        broadcast is a device-to-device copy."""
        n = len(device_ids)
        devs = (ctypes.c_int * n)(*[int(d) for d in device_ids])
        hs = (ctypes.c_void_p * n)()
        check(lib().cyto_comm_init_local(n, devs, hs))
        return [cls(None, r, n, int(device_ids[r]), _handle=ctypes.c_void_p(hs[r])) for r in range(n)]

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib().cyto_comm_unique_id(buf))
        return buf.raw

    @property
    def handle(self):
        return self._h

    def count(self):
        """Ranks the communicator spans, as RCCL itself counts them (ncclCommCount)."""
        n = ctypes.c_int()
        check(lib().cyto_comm_count(self._h, ctypes.byref(n)))
        return n.value

    def kind(self):
        k = ctypes.c_int()
        check(lib().cyto_comm_kind(self._h, ctypes.byref(k)))
        return "rccl" if k.value == 0 else "in-process"

    def agree(self, status=0):
        """Collective: the largest status any rank brought (0: every rank is fine)."""
        s = ctypes.c_int(int(status))
        check(lib().cyto_comm_agree(self._h, ctypes.byref(s)))
        return s.value

    def abort(self):
        """This rank cannot reach a collective its peers wait in: release them (they fail with CYTO_ERR_PEER).  Communicators
        made by init_local are aborted TOGETHER (every sibling, whichever rank calls)."""
        if self._h:
            lib().cyto_comm_abort(self._h)

    def aborted(self):
        a = ctypes.c_int()
        check(lib().cyto_comm_aborted(self._h, ctypes.byref(a)))
        return bool(a.value)

    def close(self):
        if self._h:
            lib().cyto_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
