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
