"""ctypes binding of include/cvr_amd.h (libcvr_amd.so).  Mirrors the reference's call sequence
(main, spmv.cpp:1771-1938): load -> create/preprocess (pre_processing) -> spmv (spmv_compute_kernel) -> verdict."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_IO, ERR_NOMEM, ERR_STATE, ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7
MM_REFCOMPAT, MM_STRICT = 0, 1


_torch_first = None      # was torch already imported when libcvr_amd.so was loaded?


class CvrError(RuntimeError):
    def __init__(self, code, where):
        import sys as _sys
        self.code = code
        hint = ""
        if code == ERR_NO_DEVICE and _torch_first is False and "torch" in _sys.modules:
            hint = (" -- PyTorch was imported AFTER libcvr_amd.so: two copies of the HIP runtime share the process and the second one"
                    " finds no GPU; import torch first (or set CVR_TORCH_PRELOAD=1)")
        super().__init__(f"{where}: error {code}: {last_error()}{hint}")


class CsrView(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p),
                ("vals", C.c_void_p), ("is_f32", C.c_int32), ("arrays_on_device", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("steps_per_chunk", C.c_int32), ("split_threshold", C.c_int64),
                ("xcd_swizzle", C.c_int32), ("x_window", C.c_int32), ("waves_per_block", C.c_int32),
                ("col_panels", C.c_int32), ("value_dict", C.c_int32), ("col_phases", C.c_int32), ("hub_table", C.c_int32), ("narrow_cols", C.c_int32),
                ("hub_reorder", C.c_int32), ("row_tags16", C.c_int32), ("row_bands", C.c_int32), ("piece_max", C.c_int32), ("interleave", C.c_int32), ("gang", C.c_int32), ("reserved", C.c_int32 * 3)]


class Timing(C.Structure):
    _fields_ = [("iters", C.c_int32), ("mean_s", C.c_double), ("min_s", C.c_double), ("max_s", C.c_double),
                ("total_s", C.c_double), ("h2d_s", C.c_double), ("d2h_s", C.c_double), ("median_s", C.c_double),
                ("step_mean_s", C.c_double), ("step_min_s", C.c_double), ("step_median_s", C.c_double), ("step_max_s", C.c_double),
                ("gather_mean_s", C.c_double)]


class Info(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64), ("is_f32", C.c_int32),
                ("steps_per_chunk", C.c_int32), ("nchunks", C.c_int64), ("nslots", C.c_int64), ("nshared", C.c_int64),
                ("image_bytes", C.c_int64), ("yext_elems", C.c_int64), ("x_elems", C.c_int64),
                ("plan_s", C.c_double), ("upload_s", C.c_double), ("convert_s", C.c_double),
                ("col_panels", C.c_int32), ("value_dict", C.c_int32), ("col_phases", C.c_int32), ("waves_per_block", C.c_int32),
                ("x_window", C.c_int32), ("lds_bytes", C.c_int32), ("nsegments", C.c_int64), ("chunk_row_cap", C.c_int64), ("near_diagonal_share", C.c_double), ("hub_entries", C.c_int32), ("narrow_cols", C.c_int32), ("hub_reorder", C.c_int32), ("row_tags16", C.c_int32), ("hub_share", C.c_double),
                ("hub_select_s", C.c_double), ("probe_s", C.c_double), ("dict_s", C.c_double), ("preprocess_wall_s", C.c_double), ("row_bands", C.c_int32), ("piece_max", C.c_int32), ("spmv_launches", C.c_int32), ("preprocess_fused", C.c_int32), ("interleave", C.c_int32), ("gang", C.c_int32)]


class MmMatrix(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64), ("ref_numRows", C.c_int64),
                ("ref_numCols", C.c_int64), ("ref_nItems", C.c_int64), ("ref_nItemsRaw", C.c_int64),
                ("row_ptr", C.POINTER(C.c_int64)), ("col_idx", C.POINTER(C.c_int32)), ("vals", C.POINTER(C.c_double))]


# every symbol include/cvr_amd.h declares (tests check the library exports all of them)
SYMBOLS = ["cvr_default_options", "cvr_last_error", "cvr_version", "cvr_device_count", "cvr_create", "cvr_preprocess",
           "cvr_get_info", "cvr_destroy", "cvr_spmv", "cvr_spmv_device", "cvr_spmv_device_repeat", "cvr_x_device", "cvr_y_device", "cvr_stream",
           "cvr_spmv_bench", "cvr_debug_phase_clocks", "cvr_device_copy_bench", "cvr_export_image", "cvr_export_gang", "cvr_comm_info", "cvr_plan_bound", "cvr_plan_chunks", "cvr_plan_selfcheck", "cvr_mm_read", "cvr_mm_free", "cvr_mm_write_bin", "cvr_mm_read_bin",
           "cvr_fill_x", "cvr_csr_spmv_host", "cvr_verdict",
           "cvr_tune_steps", "cvr_tune", "cvr_auto_panels", "cvr_power_iteration", "cvr_comm_unique_id", "cvr_comm_create", "cvr_comm_destroy", "cvr_comm_all_gather", "cvr_spmv_gather_repeat",
           "cvr_source_key_of", "cvr_mm_write_bin_keyed", "cvr_mm_read_bin_keyed", "cvr_mm_read_cached", "cvr_save_image", "cvr_load_image",
           "cvr_row_partition", "cvr_row_partition_cost", "cvr_create_multi", "cvr_preprocess_multi", "cvr_spmv_multi", "cvr_multi_shards", "cvr_multi_info", "cvr_multi_uses_rccl", "cvr_destroy_multi", "cvr_multi_from_handles", "cvr_multi_handle"]


def lib_path():
    return os.path.join(_HERE, "libcvr_amd.so")


def lib():
    """Loads libcvr_amd.so; raises if it is not built -- there is no fallback."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError(f"{p} is missing: build it with `make -C cvr_amd/csrc` (or __graft_entry__.build())")
        # A process that also uses PyTorch must load torch FIRST: its wheel carries its own copy of the HIP runtime, and whichever
        # copy is loaded second finds no GPU.  The order is the caller's (importing torch here would put a second
        # runtime into processes that never wanted one: the CSR comparators and rocprofv3 crash on that); what this module does is
        # say so when it sees the wrong order behind a "no device" error (CvrError), and CVR_TORCH_PRELOAD=1 imports torch here.
        import sys as _sys
        global _torch_first
        if "torch" not in _sys.modules and os.environ.get("CVR_TORCH_PRELOAD"):      # opt-in: a caller that will import torch later
            try:
                import torch  # noqa: F401
            except Exception:  # noqa: BLE001
                pass
        _torch_first = "torch" in _sys.modules
        L = C.CDLL(p)
        L.cvr_last_error.restype = C.c_char_p
        L.cvr_version.restype = C.c_char_p
        L.cvr_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(CsrView), C.POINTER(Options)]
        L.cvr_preprocess.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.cvr_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.cvr_destroy.argtypes = [C.c_void_p]
        L.cvr_spmv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Timing)]
        L.cvr_spmv_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvr_spmv_device_repeat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        for f in ("cvr_x_device", "cvr_y_device", "cvr_stream"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_void_p
        L.cvr_spmv_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.cvr_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.cvr_export_image.argtypes = [C.c_void_p] * 5
        L.cvr_export_gang.argtypes = [C.c_void_p] * 3
        L.cvr_device_copy_bench.argtypes = [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double)]
        L.cvr_plan_selfcheck.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cvr_plan_selfcheck.restype = C.c_int
        L.cvr_plan_bound.argtypes = [C.c_int64, C.c_int64, C.c_int32]
        L.cvr_plan_bound.restype = C.c_int64
        L.cvr_plan_chunks.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_int64] + [C.c_void_p] * 4
        L.cvr_plan_chunks.restype = C.c_int64
        L.cvr_mm_read.argtypes = [C.c_char_p, C.c_int, C.POINTER(MmMatrix)]
        L.cvr_mm_free.argtypes = [C.POINTER(MmMatrix)]
        L.cvr_mm_write_bin.argtypes = [C.c_char_p, C.POINTER(MmMatrix)]
        L.cvr_mm_read_bin.argtypes = [C.c_char_p, C.POINTER(MmMatrix)]
        L.cvr_fill_x.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.cvr_csr_spmv_host.argtypes = [C.c_int64] + [C.c_void_p] * 5 + [C.c_int]
        L.cvr_verdict.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.cvr_verdict.restype = C.c_int64
        L.cvr_tune_steps.argtypes = [C.POINTER(CsrView), C.POINTER(Options), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.cvr_tune.argtypes = [C.POINTER(CsrView), C.POINTER(Options), C.POINTER(Options), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.cvr_auto_panels.argtypes = [C.POINTER(CsrView), C.POINTER(C.c_double)]
        L.cvr_power_iteration.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
        L.cvr_comm_unique_id.argtypes = [C.c_void_p]
        L.cvr_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.cvr_comm_destroy.argtypes = [C.c_void_p]
        L.cvr_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.cvr_comm_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.cvr_spmv_gather_repeat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                             C.c_int64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.cvr_multi_handle.argtypes = [C.c_void_p, C.c_int32]
        L.cvr_multi_handle.restype = C.c_void_p
        L.cvr_multi_from_handles.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int32]
        L.cvr_source_key_of.argtypes = [C.c_char_p, C.c_int, C.POINTER(SourceKey)]
        L.cvr_mm_write_bin_keyed.argtypes = [C.c_char_p, C.POINTER(MmMatrix), C.POINTER(SourceKey)]
        L.cvr_mm_read_bin_keyed.argtypes = [C.c_char_p, C.POINTER(SourceKey), C.POINTER(MmMatrix)]
        L.cvr_mm_read_cached.argtypes = [C.c_char_p, C.c_int, C.POINTER(MmMatrix), C.POINTER(C.c_int)]
        L.cvr_save_image.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(SourceKey)]
        L.cvr_load_image.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.POINTER(SourceKey), C.POINTER(Options), C.POINTER(C.c_double)]
        L.cvr_row_partition.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        L.cvr_row_partition.restype = C.c_int64
        L.cvr_row_partition_cost.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.cvr_row_partition_cost.restype = C.c_int64
        L.cvr_create_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(CsrView), C.POINTER(Options), C.c_void_p, C.c_int32]
        L.cvr_preprocess_multi.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.cvr_spmv_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Timing)]
        L.cvr_multi_shards.argtypes = [C.c_void_p]
        L.cvr_multi_uses_rccl.argtypes = [C.c_void_p]
        L.cvr_multi_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Info), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.cvr_destroy_multi.argtypes = [C.c_void_p]
        _lib = L
    return _lib


ROW_COST_MILLI_DEFAULT = 1250      # include/cvr_amd.h: CVR_ROW_COST_MILLI_DEFAULT


def row_partition(row_ptr, nparts, row_cost_milli=0):
    """cvr_row_partition_cost: bounds[nparts + 1] of contiguous row blocks cut at row boundaries with balanced cost = non-zeros +
    row_cost_milli / 1000 per row (0: balanced non-zeros, the reference's rule = cvr_row_partition).  Host only."""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    bounds = np.zeros(nparts + 1, dtype=np.int64)
    rc = lib().cvr_row_partition_cost(len(rp) - 1, rp.ctypes.data, nparts, int(row_cost_milli), bounds.ctypes.data)
    if rc < 0:
        raise CvrError(int(rc), "cvr_row_partition_cost")
    return bounds


def last_error():
    return lib().cvr_last_error().decode()


def version():
    return lib().cvr_version().decode()


def device_count():
    return lib().cvr_device_count()


def device_copy_gbs(device=0, nbytes=1 << 30, iters=20):
    """measured read+write rate of a streaming copy kernel (GB/s): the achievable-HBM yardstick"""
    g = C.c_double()
    rc = lib().cvr_device_copy_bench(device, nbytes, iters, C.byref(g))
    if rc:
        raise CvrError(rc, "cvr_device_copy_bench")
    return g.value


class SourceKey(C.Structure):
    _fields_ = [("size", C.c_int64), ("mtime_ns", C.c_int64), ("hash", C.c_uint64), ("mode", C.c_int32), ("reserved", C.c_int32)]


def source_key(path, mode=MM_REFCOMPAT):
    k = SourceKey()
    rc = lib().cvr_source_key_of(os.fsencode(path), mode, C.byref(k))
    if rc:
        raise CvrError(rc, f"cvr_source_key_of({path})")
    return k


def load_mm(path, mode=MM_REFCOMPAT, cache=None):
    """cvr_mm_read -> dict(nrows, ncols, nnz, ref_*, row_ptr int64, col_idx int32, vals float64) (numpy copies).
    cache=True: through the keyed cache beside the file (cvr_mm_read_cached: <path>.ref.csrbin / .strict.csrbin, read only while
    its key -- size, mtime, hash of the first and last MiB -- is the file's); out["cache_hit"] says which.
    cache=<path>: an UNKEYED binary image; read when it exists, written after a text parse otherwise (the caller answers for
    its freshness)."""
    m = MmMatrix()
    hit = None
    if cache is True:
        h = C.c_int()
        rc = lib().cvr_mm_read_cached(os.fsencode(path), mode, C.byref(m), C.byref(h))
        if rc:
            raise CvrError(rc, f"cvr_mm_read_cached({path})")
        hit = bool(h.value)
    elif cache and os.path.exists(cache):
        rc = lib().cvr_mm_read_bin(os.fsencode(cache), C.byref(m))
        if rc:
            raise CvrError(rc, f"cvr_mm_read_bin({cache})")
    else:
        rc = lib().cvr_mm_read(os.fsencode(path), mode, C.byref(m))
        if rc:
            raise CvrError(rc, f"cvr_mm_read({path})")
        if cache and lib().cvr_mm_write_bin(os.fsencode(cache), C.byref(m)):
            raise CvrError(ERR_IO, f"cvr_mm_write_bin({cache})")
    n = max(m.ref_nItems, m.nnz)
    out = dict(nrows=m.nrows, ncols=m.ncols, nnz=m.nnz, ref_numRows=m.ref_numRows, ref_numCols=m.ref_numCols,
               ref_nItems=m.ref_nItems, ref_nItemsRaw=m.ref_nItemsRaw,
               row_ptr=np.ctypeslib.as_array(m.row_ptr, shape=(m.nrows + 1,)).copy(),
               col_idx=np.ctypeslib.as_array(m.col_idx, shape=(max(n, 1),))[:n].copy(),
               vals=np.ctypeslib.as_array(m.vals, shape=(max(n, 1),))[:n].copy())
    lib().cvr_mm_free(C.byref(m))
    if hit is not None:
        out["cache_hit"] = hit
    return out


def fill_x(n, mode=0):
    x = np.empty(n, dtype=np.float64)
    lib().cvr_fill_x(x.ctypes.data, n, mode)
    return x


def csr_spmv_host(row_ptr, col_idx, vals, x, nthreads=1):
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    ci = np.ascontiguousarray(col_idx, dtype=np.int32)
    va = np.ascontiguousarray(vals, dtype=np.float64)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(len(rp) - 1, dtype=np.float64)
    lib().cvr_csr_spmv_host(len(rp) - 1, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, xx.ctypes.data, y.ctypes.data, nthreads)
    return y


def verdict(y, yref, n):
    y = np.ascontiguousarray(y, dtype=np.float64)
    yref = np.ascontiguousarray(yref, dtype=np.float64)
    return lib().cvr_verdict(y.ctypes.data, yref.ctypes.data, n)


def plan_chunks(row_ptr, S, thr=0):
    """host planner only (runs without a GPU): dict(nz_begin[n+1], row_first[n], nseg[n], pad_cnt[n])"""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    nrows = len(rp) - 1
    bound = lib().cvr_plan_bound(nrows, int(rp[-1] - rp[0]) if nrows else 0, S)
    nzb = np.zeros(bound + 1, dtype=np.int64)
    rf, ns, pc = (np.zeros(bound, dtype=np.int64) for _ in range(3))
    n = lib().cvr_plan_chunks(nrows, rp.ctypes.data, S, thr, nzb.ctypes.data, rf.ctypes.data, ns.ctypes.data, pc.ctypes.data)
    if n < 0:
        raise CvrError(n, "cvr_plan_chunks")
    return dict(nz_begin=nzb[: n + 1].copy(), row_first=rf[:n].copy(), nseg=ns[:n].copy(), pad_cnt=pc[:n].copy())


def plan_selfcheck(row_ptr, S, thr=0, max_rows=0, device=0):
    """plans row_ptr on the device and on the host and compares the plans field by field (raises CvrError if they differ);
    returns dict(host_s, device_s, nchunks)"""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    hs, ds, n = C.c_double(), C.c_double(), C.c_int64()
    rc = lib().cvr_plan_selfcheck(device, len(rp) - 1, rp.ctypes.data, S, thr, max_rows, C.addressof(hs), C.addressof(ds), C.addressof(n))
    if rc:
        raise CvrError(rc, "cvr_plan_selfcheck")
    return dict(host_s=hs.value, device_s=ds.value, nchunks=n.value)


def auto_panels(nrows, ncols, row_ptr, col_idx, is_f32=False):
    """(column panels cvr_create would choose, estimated L2 miss share of the x gathers); host only"""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    ci = np.ascontiguousarray(col_idx, dtype=np.int32)
    view = CsrView(nrows, ncols, rp.ctypes.data, ci.ctypes.data, None, int(is_f32))
    miss = C.c_double()
    P = lib().cvr_auto_panels(C.byref(view), C.byref(miss))
    if P < 0:
        raise CvrError(P, "cvr_auto_panels")
    return P, miss.value


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes from rank 0 that every rank passes to Comm() (hand them over with torch.distributed, a file, ...)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().cvr_comm_unique_id(buf)
    if rc:
        raise CvrError(rc, "cvr_comm_unique_id")
    return buf.raw


class Comm:
    """RCCL communicator of the row-sharded SpMV, one rank per process and GPU (cvr_comm_create; collective)"""

    def __init__(self, unique_id, nranks, rank, device):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self._c = C.c_void_p()
        self.nranks, self.rank, self.device = nranks, rank, device
        rc = lib().cvr_comm_create(C.byref(self._c), unique_id, nranks, rank, device)
        if rc:
            raise CvrError(rc, "cvr_comm_create")

    def info(self):
        """(ranks, this rank, RCCL version) as the library reports them for this communicator (cvr_comm_info); -1 where it cannot say"""
        n, r, v = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        rc = lib().cvr_comm_info(self._c, C.byref(n), C.byref(r), C.byref(v))
        if rc:
            raise CvrError(rc, "cvr_comm_info")
        return n.value, r.value, v.value

    def all_gather(self, send_ptr, recv_ptr, count, is_f32=False, stream=None):
        rc = lib().cvr_comm_all_gather(self._c, send_ptr, recv_ptr, count, int(is_f32), stream)
        if rc:
            raise CvrError(rc, "cvr_comm_all_gather")

    def close(self):
        if self._c:
            lib().cvr_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CvrMatrix:
    """One matrix (or row shard) resident on one GPU: cvr_create + cvr_preprocess, then spmv()."""

    def __init__(self, nrows, ncols, row_ptr, col_idx, vals, device=0, steps_per_chunk=0, split_threshold=0,
                 xcd_swizzle=-1, x_window=-1, keep_csr=False, debug_col_mask=0,
                 col_panels=-1, value_dict=-1, tune_steps=False, waves_per_block=0, col_phases=-1, hub_table=-1, narrow_cols=-1, hub_reorder=-1,
                 row_tags16=-1, row_bands=-1, piece_max=-1, interleave=-1, gang=-1):
        """tune_steps: choose steps_per_chunk by measurement first (cvr_tune_steps; its cost is self.tuning_s).
        debug_col_mask is a profiling knob (tools/sweep.py): it travels through the environment (CVR_DEBUG_COL_MASK), not through cvr_options."""
        self._h = C.c_void_p()
        self.tuning_s = 0.0
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        ci = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.f32 = np.asarray(vals).dtype == np.float32
        self.dtype = np.float32 if self.f32 else np.float64
        va = np.ascontiguousarray(vals, dtype=self.dtype)
        if len(rp) != nrows + 1:
            raise ValueError("row_ptr must have nrows + 1 entries")
        if nrows > 0 and rp[0] < 0:
            raise ValueError("row_ptr[0] < 0")
        if nrows > 0 and (len(ci) < rp[-1] or len(va) < rp[-1]):      # the library reads row_ptr[nrows] entries of both
            raise ValueError(f"col_idx / vals hold {len(ci)} / {len(va)} entries, row_ptr[nrows] = {int(rp[-1])}")
        view = CsrView(nrows, ncols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, int(self.f32))
        self._build(view, nrows, ncols, device, steps_per_chunk, split_threshold, xcd_swizzle, x_window, keep_csr,
                    debug_col_mask, col_panels, value_dict, tune_steps, waves_per_block, col_phases, hub_table, narrow_cols, hub_reorder,
                    row_tags16, row_bands, piece_max, interleave, gang)

    def save_image(self, path, key=None):
        """cvr_save_image: the converted image on disk, keyed by `key` (capi.source_key of the .mtx file; None = no source key),
        the options and the device geometry"""
        rc = lib().cvr_save_image(self._h, os.fsencode(path), C.byref(key) if key is not None else None)
        if rc:
            raise CvrError(rc, "cvr_save_image")

    @classmethod
    def from_image(cls, path, key=None, device=0, **options):
        """cvr_load_image: a handle from a saved image (CvrError with code ERR_STATE when the file was written for another source,
        other options, another device geometry or library version)"""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.tuning_s = 0.0
        opt = Options()
        lib().cvr_default_options(C.byref(opt))
        opt.device = device
        for k, v in options.items():
            setattr(opt, k, v)
        sec = C.c_double()
        rc = lib().cvr_load_image(C.byref(self._h), os.fsencode(path), C.byref(key) if key is not None else None, C.byref(opt), C.byref(sec))
        if rc:
            self._h = C.c_void_p()
            raise CvrError(rc, "cvr_load_image")
        self.load_s = self.preprocess_s = sec.value
        self.info = Info()
        lib().cvr_get_info(self._h, C.byref(self.info))
        self.nrows, self.ncols = self.info.nrows, self.info.ncols
        self.f32 = bool(self.info.is_f32)
        self.dtype = np.float32 if self.f32 else np.float64
        return self

    @classmethod
    def from_device(cls, nrows, ncols, row_ptr_dev, col_idx_dev, vals_dev, is_f32=False, device=0, steps_per_chunk=0,
                    split_threshold=0, keep_csr=False, col_panels=-1, value_dict=-1, tune_steps=False, hub_table=-1, hub_reorder=-1, interleave=-1, gang=-1):
        """CSR arrays already in the memory of `device` (raw pointers: int64 row_ptr[nrows+1], int32 col_idx, fp64/fp32 vals),
        e.g. the .data_ptr() of torch tensors: cvr_csr_view.arrays_on_device = 1"""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.tuning_s = 0.0
        self.f32 = bool(is_f32)
        self.dtype = np.float32 if self.f32 else np.float64
        view = CsrView(nrows, ncols, row_ptr_dev, col_idx_dev, vals_dev, int(self.f32), 1)
        self._build(view, nrows, ncols, device, steps_per_chunk, split_threshold, -1, -1, keep_csr, 0, col_panels, value_dict, tune_steps,
                    hub_table=hub_table, hub_reorder=hub_reorder, interleave=interleave, gang=gang)
        return self

    def _build(self, view, nrows, ncols, device, steps_per_chunk, split_threshold, xcd_swizzle, x_window, keep_csr,
               debug_col_mask, col_panels, value_dict, tune_steps, waves_per_block=0, col_phases=-1, hub_table=-1, narrow_cols=-1, hub_reorder=-1,
               row_tags16=-1, row_bands=-1, piece_max=-1, interleave=-1, gang=-1):
        opt = Options()
        lib().cvr_default_options(C.byref(opt))
        opt.device, opt.steps_per_chunk, opt.split_threshold = device, steps_per_chunk, split_threshold
        opt.xcd_swizzle, opt.x_window, opt.col_panels, opt.value_dict = xcd_swizzle, x_window, col_panels, value_dict
        opt.waves_per_block, opt.col_phases, opt.hub_table, opt.narrow_cols = waves_per_block, col_phases, hub_table, narrow_cols
        opt.hub_reorder, opt.row_tags16, opt.row_bands, opt.piece_max = hub_reorder, row_tags16, row_bands, piece_max
        opt.interleave = interleave
        opt.gang = gang
        # profiling knobs (tools/sweep.py): cvr_create / cvr_tune read them from the environment.  Only a knob the caller passed is
        # touched, and what the environment held before comes back once the handle exists (a value the user exported stays theirs).
        saved = {}
        for name, val in (("CVR_DEBUG_COL_MASK", debug_col_mask),):
            if val:
                saved[name] = os.environ.get(name)
                os.environ[name] = str(int(val))
        try:
            self._create(view, nrows, ncols, opt, tune_steps, steps_per_chunk, keep_csr)
        finally:
            for name, old in saved.items():
                if old is None:
                    os.environ.pop(name, None)
                else:
                    os.environ[name] = old

    def _create(self, view, nrows, ncols, opt, tune_steps, steps_per_chunk, keep_csr):
        if tune_steps and steps_per_chunk == 0:          # the layout by measurement (cvr_tune): S, chunks per workgroup, x window, column phases
            best, best_t, tun = Options(), C.c_double(), C.c_double()
            rc = lib().cvr_tune(C.byref(view), C.byref(opt), C.byref(best), C.byref(best_t), C.byref(tun))
            if rc:
                raise CvrError(rc, "cvr_tune")
            opt, self.tuning_s = best, tun.value
        rc = lib().cvr_create(C.byref(self._h), C.byref(view), C.byref(opt))
        if rc:
            self._h = C.c_void_p()
            raise CvrError(rc, "cvr_create")
        sec = C.c_double()
        rc = lib().cvr_preprocess(self._h, int(keep_csr), C.byref(sec))
        if rc:
            err = CvrError(rc, "cvr_preprocess")
            self.close()
            raise err
        self.preprocess_s = sec.value
        self.info = Info()
        lib().cvr_get_info(self._h, C.byref(self.info))
        self.nrows, self.ncols = nrows, ncols

    def spmv(self, x, iters=1):
        """y = A x through host buffers; returns (y, Timing)"""
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if len(x) < self.ncols:
            raise ValueError("x is shorter than ncols")
        y = np.zeros(max(self.nrows, 1), dtype=self.dtype)
        t = Timing()
        rc = lib().cvr_spmv(self._h, x.ctypes.data, y.ctypes.data, iters, C.byref(t))
        if rc:
            raise CvrError(rc, "cvr_spmv")
        return y[: self.nrows], t

    def spmv_device(self, x_ptr, y_ptr, stream=None, repeat=1):
        """asynchronous launch(es) on caller-owned device buffers (x_ext: ncols+1 values, last one 0; y_ext)"""
        if repeat == 1:
            rc = lib().cvr_spmv_device(self._h, x_ptr, y_ptr, stream)
        else:
            rc = lib().cvr_spmv_device_repeat(self._h, x_ptr, y_ptr, stream, repeat)
        if rc:
            raise CvrError(rc, "cvr_spmv_device")

    def spmv_gather(self, comm, x_ptr, y_ptrs, yall_ptrs, max_rows, steps, stream=None, overlap=False):
        """`steps` sharded SpMVs, each followed by the all-gather of this rank's y slice over RCCL, looped inside the
        library (cvr_spmv_gather_repeat; overlap: gather of step k under the SpMV of step k+1); returns the index of
        the buffers holding the last step"""
        ys = (C.c_void_p * 2)(*y_ptrs)
        yalls = (C.c_void_p * 2)(*yall_ptrs)
        last = C.c_int()
        rc = lib().cvr_spmv_gather_repeat(self._h, comm._c, x_ptr, ys, yalls, max_rows, steps, int(overlap), stream, C.byref(last))
        if rc:
            raise CvrError(rc, "cvr_spmv_gather_repeat")
        return last.value

    def power_iteration(self, x_ptr, iters, comm=None, bounds=None, stream=None):
        """x <- A x / ||A x||, `iters` times on the device (cvr_power_iteration); returns (Rayleigh quotient, seconds per iteration)"""
        lam, sec = C.c_double(), C.c_double()
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.int64)
        rc = lib().cvr_power_iteration(self._h, None if comm is None else comm._c, None if b is None else b.ctypes.data, iters, x_ptr,
                                       C.byref(lam), C.byref(sec), stream)
        if rc:
            raise CvrError(rc, "cvr_power_iteration")
        return lam.value, sec.value

    def bench(self, warmup, iters):
        s = C.c_double()
        rc = lib().cvr_spmv_bench(self._h, warmup, iters, C.byref(s))
        if rc:
            raise CvrError(rc, "cvr_spmv_bench")
        return s.value

    def phase_clocks(self):
        """diagnostics (CVR_DEBUG=phase_clocks at creation): [workgroups][16 wavefronts][8] uint64 stamps of the last SpMV (cvr_debug_phase_clocks)"""
        n = C.c_int64()
        rc = lib().cvr_debug_phase_clocks(self._h, None, 0, C.byref(n))
        if rc:
            raise CvrError(rc, "cvr_debug_phase_clocks")
        out = np.zeros(n.value, dtype=np.uint64)
        rc = lib().cvr_debug_phase_clocks(self._h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
        if rc:
            raise CvrError(rc, "cvr_debug_phase_clocks")
        return out.reshape(-1, 16, 8)

    @property
    def x_device(self):
        return lib().cvr_x_device(self._h)

    @property
    def y_device(self):
        return lib().cvr_y_device(self._h)

    @property
    def stream(self):
        return lib().cvr_stream(self._h)

    def export_image(self):
        i = self.info
        gb = (1280 if i.value_dict else (1536 if self.f32 else 2560) if i.narrow_cols else 2048 if self.f32 else 3072) + (512 if i.row_tags16 else 0)
        image = np.zeros(i.nchunks * (i.steps_per_chunk // 4) * gb, dtype=np.uint8)
        desc = np.zeros((i.nchunks, 4), dtype=np.uint32)
        target = np.zeros((i.nchunks, 64), dtype=np.uint8)
        shared = np.zeros((i.nshared, 3), dtype=np.int64)
        rc = lib().cvr_export_image(self._h, image.ctypes.data, desc.ctypes.data, target.ctypes.data, shared.ctypes.data)
        if rc:
            raise CvrError(rc, "cvr_export_image")
        out = dict(image=image, desc=desc, target=target, shared=shared)
        if i.gang:                      # gang chunks: the groups' first columns and the gangs' group counts
            gbase = np.zeros(i.nchunks * (i.steps_per_chunk // 4), dtype=np.uint32)
            desc2 = np.zeros((i.nchunks, 2), dtype=np.uint32)
            rc = lib().cvr_export_gang(self._h, gbase.ctypes.data, desc2.ctypes.data)
            if rc:
                raise CvrError(rc, "cvr_export_gang")
            out.update(gbase=gbase, desc2=desc2)
        return out

    def close(self):
        if self._h:
            lib().cvr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiMatrix:
    """One matrix over several GPUs of this process (cvr_create_multi): rows sharded, x replicated, y all-gathered inside the
    library.  devices may name one GPU several times (copies then stand in for RCCL)."""

    def __init__(self, nrows, ncols, row_ptr, col_idx, vals, devices, **options):
        self._m = C.c_void_p()
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        ci = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.f32 = np.asarray(vals).dtype == np.float32
        self.dtype = np.float32 if self.f32 else np.float64
        va = np.ascontiguousarray(vals, dtype=self.dtype)
        if len(rp) != nrows + 1:
            raise ValueError("row_ptr must have nrows + 1 entries")
        view = CsrView(nrows, ncols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, int(self.f32))
        opt = Options()
        lib().cvr_default_options(C.byref(opt))
        for k, v in options.items():
            setattr(opt, k, v)
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        rc = lib().cvr_create_multi(C.byref(self._m), C.byref(view), C.byref(opt), devs.ctypes.data, len(devs))
        if rc:
            self._m = C.c_void_p()
            raise CvrError(rc, "cvr_create_multi")
        sec = C.c_double()
        rc = lib().cvr_preprocess_multi(self._m, 0, C.byref(sec))
        if rc:
            err = CvrError(rc, "cvr_preprocess_multi")
            self.close()
            raise err
        self.preprocess_s = sec.value
        self.nrows, self.ncols = nrows, ncols
        self.shards = lib().cvr_multi_shards(self._m)
        self.uses_rccl = bool(lib().cvr_multi_uses_rccl(self._m))

    def shard_info(self, p):
        info, b, e, d = Info(), C.c_int64(), C.c_int64(), C.c_int32()
        rc = lib().cvr_multi_info(self._m, p, C.byref(info), C.byref(b), C.byref(e), C.byref(d))
        if rc:
            raise CvrError(rc, "cvr_multi_info")
        return info, b.value, e.value, d.value

    def spmv(self, x, iters=1):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if len(x) < self.ncols:
            raise ValueError("x is shorter than ncols")
        y = np.zeros(max(self.nrows, 1), dtype=self.dtype)
        t = Timing()
        rc = lib().cvr_spmv_multi(self._m, x.ctypes.data, y.ctypes.data, iters, C.byref(t))
        if rc:
            raise CvrError(rc, "cvr_spmv_multi")
        return y[: self.nrows], t

    def close(self):
        if self._m:
            lib().cvr_destroy_multi(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
