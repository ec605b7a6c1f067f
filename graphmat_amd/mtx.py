"""Binary .mtx edge-list files, host side.

Format (reference: include/GMDP/utils/edgelist.h:92-140 readLine/get_maxid_and_nnz,
:242-334 load_edgelist): int32 m, n, nnz header followed by nnz records of
(int32 src, int32 dst, E val), ids 1-based.  The reference's shipped fixtures
physically hold nnz+1 records (the last one repeated); the header count governs
and trailing records are ignored, as the reference's loader effectively does.
"""
import numpy as np


def read_mtx_bin(path, val_dtype=np.int32):
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 12:
        raise ValueError("%s: too short for a binary .mtx header" % path)
    m, n, nnz = (int(v) for v in raw[:12].view(np.int32))
    val_dtype = np.dtype(val_dtype)
    rec = 8 + val_dtype.itemsize
    if m <= 0 or n <= 0 or nnz < 0 or raw.size < 12 + nnz * rec:
        raise ValueError("%s: header (%d,%d,%d) inconsistent with file size %d" % (path, m, n, nnz, raw.size))
    body = raw[12:12 + nnz * rec].reshape(nnz, rec)
    src = np.ascontiguousarray(body[:, 0:4]).view(np.int32).ravel()
    dst = np.ascontiguousarray(body[:, 4:8]).view(np.int32).ravel()
    val = np.ascontiguousarray(body[:, 8:]).view(val_dtype).ravel()
    nv = max(m, n)  # Graph::ReadMTX squares the matrix (Graph.h:253-257)
    return nv, src, dst, val


def write_mtx_bin(path, nv, src, dst, val):
    src = np.asarray(src, np.int32)
    dst = np.asarray(dst, np.int32)
    val = np.ascontiguousarray(val)
    rec = np.zeros((src.size, 8 + val.dtype.itemsize), np.uint8)
    rec[:, 0:4] = src.view(np.uint8).reshape(-1, 4)
    rec[:, 4:8] = dst.view(np.uint8).reshape(-1, 4)
    rec[:, 8:] = val.view(np.uint8).reshape(src.size, -1)
    with open(path, "wb") as f:
        np.array([nv, nv, src.size], np.int32).tofile(f)
        rec.tofile(f)


# ---- every edge-list variant of the reference's loader/writer, through the C-ABI ---------------
_KINDS = {np.dtype(np.int32): 1, np.dtype(np.uint32): 2, np.dtype(np.float32): 3, np.dtype(np.float64): 4}


def read_edgelist(path, binary=True, header=True, weights=True, val_dtype=np.int32):
    """(m, n, src, dst, val) of an edge-list file (gm_edgelist_read).  Needs the built library."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    kind = _KINDS[np.dtype(val_dtype)]
    m, n, nnz = C.c_int(), C.c_int(), C.c_int64()
    ps, pd, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(L.gm_edgelist_read(str(path).encode(), int(binary), int(header), int(weights), kind, C.byref(m), C.byref(n),
                                  C.byref(nnz), C.byref(ps), C.byref(pd), C.byref(pv)))
    k = nnz.value
    try:
        src = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_int32)), (k,)).copy() if k else np.zeros(0, np.int32)
        dst = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_int32)), (k,)).copy() if k else np.zeros(0, np.int32)
        nb = np.dtype(val_dtype).itemsize
        val = (np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_uint8)), (k * nb,)).copy().view(val_dtype) if k
               else np.zeros(0, val_dtype))
    finally:
        L.gm_host_free(ps); L.gm_host_free(pd); L.gm_host_free(pv)
    return m.value, n.value, src, dst, val


def write_edgelist(path, m, n, src, dst, val=None, binary=True, header=True, weights=True):
    """Write an edge-list file in any of the reference's variants (gm_edgelist_write)."""
    from . import _lib
    L = _lib.lib()
    src = np.ascontiguousarray(src, np.int32)
    dst = np.ascontiguousarray(dst, np.int32)
    if val is None:
        val = np.ones(src.size, np.int32)
    val = np.ascontiguousarray(val)
    kind = _KINDS[val.dtype]
    _lib.check(L.gm_edgelist_write(str(path).encode(), int(binary), int(header), int(weights), kind, int(m), int(n),
                                   int(src.size), src.ctypes.data, dst.ctypes.data, val.ctypes.data))
