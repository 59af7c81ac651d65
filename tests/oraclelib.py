"""ctypes view of oracle/liboracle.so -- the CHECKER (test infrastructure, never the product path)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class OrcCsr(C.Structure):
    _fields_ = [("nItems", C.c_int), ("nItemsRaw", C.c_int), ("numRows", C.c_int), ("numCols", C.c_int),
                ("val", C.POINTER(C.c_double)), ("cols", C.POINTER(C.c_int)), ("rowptr", C.POINTER(C.c_int))]


class OrcCvr8(C.Structure):
    _fields_ = [("T", C.c_int), ("nItems", C.c_int), ("numRows", C.c_int),
                ("vals", C.POINTER(C.c_double)), ("cols", C.POINTER(C.c_int)),
                ("record", C.POINTER(C.c_int)), ("record_len", C.c_int64),
                ("split", C.POINTER(C.c_int)), ("final2", C.POINTER(C.c_int)),
                ("nnz_rows", C.POINTER(C.c_int))]


class OrcCvr64(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64), ("S", C.c_int),
                ("is_f32", C.c_int), ("nchunks", C.c_int64), ("nshared", C.c_int64), ("image_bytes", C.c_int64),
                ("image", C.POINTER(C.c_uint8)), ("desc", C.POINTER(C.c_uint32)),
                ("target", C.POINTER(C.c_uint8)), ("shared", C.POINTER(C.c_int64)),
                ("nz_begin", C.POINTER(C.c_int64)), ("pad_cnt", C.POINTER(C.c_int64)),
                ("ndict", C.c_int), ("dict", C.c_uint64 * 256), ("phases", C.c_int),
                ("seg_off", C.POINTER(C.c_uint32)), ("seg_row", C.POINTER(C.c_uint16)), ("nrows_in", C.POINTER(C.c_uint32)),
                ("col_bits", C.c_int), ("hub_n", C.c_int), ("hub_cols", C.POINTER(C.c_int32)),
                ("order_n", C.c_int), ("narrow", C.c_int), ("cbase", C.POINTER(C.c_uint32)), ("tag16", C.c_int), ("ilv", C.c_int),
                ("gang", C.c_int), ("ystage", C.c_int), ("gbase", C.POINTER(C.c_uint32)), ("ggroups", C.POINTER(C.c_uint32))]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        _lib.orc_read_matrix.argtypes = [C.c_char_p, C.POINTER(OrcCsr)]
        _lib.orc_csr_spmv.argtypes = [C.c_int] + [C.c_void_p] * 5
        _lib.orc_csr_spmv64.argtypes = [C.c_int64] + [C.c_void_p] * 6
        _lib.orc_csr_spmv64_f32.argtypes = [C.c_int64] + [C.c_void_p] * 6
        _lib.orc_x_rand.argtypes = [C.c_uint64]
        _lib.orc_x_rand.restype = C.c_double
        _lib.orc_cvr8_convert.argtypes = [C.POINTER(OrcCsr), C.c_int, C.POINTER(OrcCvr8)]
        _lib.orc_cvr8_spmv.argtypes = [C.POINTER(OrcCvr8), C.c_void_p, C.c_void_p, C.c_int]
        _lib.orc_cvr64_build.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int64, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_spmv.argtypes = [C.POINTER(OrcCvr64), C.c_void_p, C.c_void_p]
        _lib.orc_cvr64_build_dict.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_build_ex.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_build_hub.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_build_all.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_build_full.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_build_tag.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_build_ilv.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int, C.POINTER(OrcCvr64)]
        _lib.orc_cvr64_build_gang.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(OrcCvr64)]
        _lib.orc_write_mtx_pattern.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    return _lib


def _np(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dtype=dt)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)


def read_matrix(path):
    """orc_read_matrix -> dict(nItems, nItemsRaw, numRows, numCols, val, cols, rowptr)"""
    m = OrcCsr()
    rc = lib().orc_read_matrix(path.encode(), C.byref(m))
    if rc:
        raise RuntimeError(f"orc_read_matrix({path}) = {rc}")
    out = dict(nItems=m.nItems, nItemsRaw=m.nItemsRaw, numRows=m.numRows, numCols=m.numCols,
               val=_np(m.val, m.nItems, np.float64), cols=_np(m.cols, m.nItems, np.int32),
               rowptr=_np(m.rowptr, m.numRows + 2, np.int32))
    lib().orc_free_csr(C.byref(m))
    return out


def write_mtx_pattern(path, nrows, ncols, rowptr, cols):
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    ci = np.ascontiguousarray(cols, dtype=np.int32)
    if lib().orc_write_mtx_pattern(os.fsencode(path), nrows, ncols, rp.ctypes.data, ci.ctypes.data):
        raise OSError(f"cannot write {path}")


def x_vec(n, mode):
    if mode == "ones":
        return np.ones(n, dtype=np.float64)
    return np.array([lib().orc_x_rand(j) for j in range(n)], dtype=np.float64)


def x_vec_fast(n, mode="rand"):
    """vectorised splitmix64 identical to orc_x_rand"""
    if mode == "ones":
        return np.ones(n, dtype=np.float64)
    j = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(0xC0FFEE) + (j + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0


def csr_spmv_ref(m, x):
    """the reference's CSR loop on reference-layout arrays (rows 0..numRows-1)"""
    y = np.zeros(m["numRows"], dtype=np.float64)
    rp = np.ascontiguousarray(m["rowptr"], dtype=np.int32)
    cl = np.ascontiguousarray(m["cols"], dtype=np.int32)
    vl = np.ascontiguousarray(m["val"], dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().orc_csr_spmv(m["numRows"], rp.ctypes.data, cl.ctypes.data, vl.ctypes.data, x.ctypes.data, y.ctypes.data)
    return y


def csr_spmv64(rowptr, cols, vals, x):
    """CSR oracle on 0-based int64/int32 arrays; returns (y, sum|a x|) in float64"""
    n = len(rowptr) - 1
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    cl = np.ascontiguousarray(cols, dtype=np.int32)
    y = np.zeros(n, dtype=np.float64)
    a = np.zeros(n, dtype=np.float64)
    if vals.dtype == np.float32:
        vl = np.ascontiguousarray(vals)
        xx = np.ascontiguousarray(x, dtype=np.float32)
        lib().orc_csr_spmv64_f32(n, rp.ctypes.data, cl.ctypes.data, vl.ctypes.data, xx.ctypes.data, y.ctypes.data, a.ctypes.data)
    else:
        vl = np.ascontiguousarray(vals, dtype=np.float64)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        lib().orc_csr_spmv64(n, rp.ctypes.data, cl.ctypes.data, vl.ctypes.data, xx.ctypes.data, y.ctypes.data, a.ctypes.data)
    return y, a


class Cvr8:
    def __init__(self, m, T):
        self._m = OrcCsr()
        self._keep = [np.ascontiguousarray(m["val"], dtype=np.float64), np.ascontiguousarray(m["cols"], dtype=np.int32),
                      np.ascontiguousarray(m["rowptr"], dtype=np.int32)]
        self._m.nItems, self._m.nItemsRaw = m["nItems"], m.get("nItemsRaw", m["nItems"])
        self._m.numRows, self._m.numCols = m["numRows"], m["numCols"]
        self._m.val = self._keep[0].ctypes.data_as(C.POINTER(C.c_double))
        self._m.cols = self._keep[1].ctypes.data_as(C.POINTER(C.c_int))
        self._m.rowptr = self._keep[2].ctypes.data_as(C.POINTER(C.c_int))
        self.c = OrcCvr8()
        self.rc = lib().orc_cvr8_convert(C.byref(self._m), T, C.byref(self.c))
        self.T = T
        self.numRows = m["numRows"]
        if self.rc == 0:
            c = self.c
            self.vals = _np(c.vals, c.nItems, np.float64)
            self.cols = _np(c.cols, c.nItems, np.int32)
            self.record = _np(c.record, c.record_len, np.int32)
            self.split = _np(c.split, 2 * T, np.int32)
            self.final2 = _np(c.final2, 16 * T, np.int32)
            self.nnz_rows = _np(c.nnz_rows, 4 * T, np.int32)

    def spmv(self, x, nthreads=1):
        y = np.zeros(self.numRows + 2, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        lib().orc_cvr8_spmv(C.byref(self.c), x.ctypes.data, y.ctypes.data, nthreads)
        return y

    def __del__(self):
        try:
            if self.rc == 0:
                lib().orc_cvr8_free(C.byref(self.c))
        except Exception:
            pass


class Cvr64:
    """CPU mirror of the device format (arrays copied to numpy)"""

    def __init__(self, nrows, ncols, rowptr, cols, vals, S, thr=0, use_dict=False, phases=1, max_rows=0, hub_max=0, narrow=False, reorder=False, tag16=False, piece_max=0, interleave=False,
                 gang=0, ystage=0):
        self.rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.cl = np.ascontiguousarray(cols, dtype=np.int32)
        self.f32 = vals.dtype == np.float32
        self.vl = np.ascontiguousarray(vals, dtype=np.float32 if self.f32 else np.float64)
        self.c = OrcCvr64()
        if gang:
            self.rc = lib().orc_cvr64_build_gang(nrows, ncols, self.rp.ctypes.data, self.cl.ctypes.data, self.vl.ctypes.data, int(self.f32), S, thr, int(use_dict),
                                                 max_rows, int(bool(tag16)), int(gang), int(ystage), C.byref(self.c))
        elif interleave:
            self.rc = lib().orc_cvr64_build_ilv(nrows, ncols, self.rp.ctypes.data, self.cl.ctypes.data, self.vl.ctypes.data, int(self.f32), S, thr, int(use_dict),
                                                max_rows, int(bool(tag16)), C.byref(self.c))
        else:
            self.rc = lib().orc_cvr64_build_tag(nrows, ncols, self.rp.ctypes.data, self.cl.ctypes.data, self.vl.ctypes.data,
                                                int(self.f32), S, thr, int(use_dict), phases, max_rows, hub_max, int(bool(reorder)), int(bool(narrow)), int(bool(tag16)), int(piece_max), C.byref(self.c))
        if self.rc:
            raise RuntimeError(f"orc_cvr64_build = {self.rc}")
        c = self.c
        self.nrows, self.ncols, self.S, self.nchunks, self.nshared = c.nrows, c.ncols, c.S, c.nchunks, c.nshared
        self.image = _np(c.image, c.image_bytes, np.uint8)
        self.desc = _np(c.desc, 4 * c.nchunks, np.uint32).reshape(-1, 4)
        self.target = _np(c.target, 64 * c.nchunks, np.uint8).reshape(-1, 64)
        self.shared = _np(c.shared, 3 * c.nshared, np.int64).reshape(-1, 3)
        self.nz_begin = _np(c.nz_begin, c.nchunks + 1 if c.nchunks else 0, np.int64)
        self.pad_cnt = _np(c.pad_cnt, c.nchunks, np.int64)
        self.ndict = c.ndict
        self.phases = c.phases
        self.hub_n = c.hub_n
        self.hub_cols = _np(c.hub_cols, c.hub_n, np.int32) if c.hub_n else np.zeros(0, np.int32)
        if c.gang:
            self.gbase = _np(c.gbase, c.nchunks * (c.S // 4), np.uint32)
            self.ggroups = _np(c.ggroups, c.nchunks, np.uint32)
        if c.phases > 1:
            self.seg_off = _np(c.seg_off, c.nchunks + 1, np.uint32)
            self.seg_row = _np(c.seg_row, int(self.seg_off[-1]) if c.nchunks else 0, np.uint16)
            self.nrows_in = _np(c.nrows_in, c.nchunks, np.uint32)

    def spmv(self, x):
        """x: ncols values (the pad slot x_ext[ncols] = 0 is appended here)"""
        dt = np.float32 if self.f32 else np.float64
        xe = np.zeros(self.ncols + 1, dtype=dt)
        xe[: self.ncols] = np.asarray(x, dtype=dt)[: self.ncols]
        y = np.zeros(max(self.nrows, 1), dtype=dt)
        lib().orc_cvr64_spmv(C.byref(self.c), xe.ctypes.data, y.ctypes.data)
        return y[: self.nrows]

    def __del__(self):
        try:
            lib().orc_cvr64_free(C.byref(self.c))
        except Exception:
            pass


def tol_check(y, yref, absy, tol=1e-12):
    """|y - y_ref| <= tol * sum_j |a_ij x_j| + tiny, per row"""
    err = np.abs(np.asarray(y, dtype=np.float64) - yref)
    bound = tol * absy + 1e-300
    bad = np.nonzero(err > bound)[0]
    return bad, (err / np.maximum(absy, 1e-300)).max() if len(err) else 0.0
