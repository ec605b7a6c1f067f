"""ctypes binding of the oracle (oracle/libgm_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (graphmat_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgm_oracle.so")
# the same restatement compiled the way the reference's own Makefile compiles the reference (-O3 -march=native, the compiler's default
# floating-point contraction: Makefile:24-36 of the reference) -- on a host with FMA its multiply-adds are FUSED.  Used by
# tests/test_oracle_golden.py to measure how far the reference's results move with its compiler flags.
_SO_FMA = os.path.join(_HERE, "libgm_oracle_fma.so")
_libs = {}


def build(force=False, fused=False):
    """Compile the C++ restatement with g++ (no GPU needed)."""
    so = _SO_FMA if fused else _SO
    srcs = [os.path.join(_HERE, f) for f in ("gm_oracle_capi.cpp", "gm_oracle.hpp", "Makefile")]
    if (not force) and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)], stdout=subprocess.DEVNULL)
    return so


def lib(fused=False):
    _lib = _libs.get(bool(fused))
    if _lib is None:
        L = C.CDLL(build(fused=fused))
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        L.gmo_graph_create.restype = C.c_void_p
        L.gmo_graph_create.argtypes = [C.c_int, C.c_longlong, i32p, i32p, C.c_void_p, C.c_int]
        L.gmo_graph_destroy.argtypes = [C.c_void_p]
        L.gmo_degree.argtypes = [C.c_void_p, i32p]
        L.gmo_pagerank.argtypes = [C.c_void_p, C.c_float, C.c_int,
                                   np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), i32p, C.c_void_p, C.c_int]
        L.gmo_bfs.argtypes = [C.c_void_p, C.c_int, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"),
                              np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS"), C.c_void_p, C.c_int]
        L.gmo_sssp.argtypes = [C.c_void_p, C.c_int, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.gmo_sgd_f64.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, f64p]
        L.gmo_sgd_f32.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, f32p]
        L.gmo_mapreduce_double_sum.restype = C.c_int
        L.gmo_mapreduce_double_sum.argtypes = [i32p, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.c_int, C.c_int, C.c_int]
        L.gmo_rmse_f64.restype = C.c_double
        L.gmo_rmse_f64.argtypes = [C.c_void_p, C.c_int, f64p, C.c_void_p]
        L.gmo_rmse_f32.restype = C.c_double
        L.gmo_rmse_f32.argtypes = [C.c_void_p, C.c_int, f32p, C.c_void_p]
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        L.gmo_spmv_f64.argtypes = [C.c_void_p, C.c_int, f64p, u8p, f64p, u8p]
        L.gmo_vertex_to_native.argtypes = [C.c_int] * 3
        L.gmo_native_to_vertex.argtypes = [C.c_int] * 3
        L.gmo_set_num_threads.argtypes = [C.c_int]
        _libs[bool(fused)] = _lib = L
    return _lib


class OracleGraph:
    """Adjacency of the reference (A and AT, DCSC) for a given edge list.

    src/dst are 1-based ids; ref_threads is the OMP_NUM_THREADS of the reference
    configuration being restated (enters the id permutation, Graph.h:117).
    """

    def __init__(self, nv, src, dst, val=None, ref_threads=1, fused=False):
        self._L = lib(fused)
        self.nv = int(nv)
        self.src = np.ascontiguousarray(src, dtype=np.int32)
        self.dst = np.ascontiguousarray(dst, dtype=np.int32)
        self.nnz = int(self.src.size)
        self.val = None if val is None else np.ascontiguousarray(val, dtype=np.int32)
        self.ref_threads = int(ref_threads)
        vp = None if self.val is None else self.val.ctypes.data_as(C.c_void_p)
        self.h = self._L.gmo_graph_create(self.nv, self.nnz, self.src, self.dst, vp, self.ref_threads)

    def __del__(self):
        if getattr(self, "h", None):
            self._L.gmo_graph_destroy(self.h)
            self.h = None

    def degree(self):
        d = np.zeros(self.nv, np.int32)
        self._L.gmo_degree(self.h, d)
        return d

    def pagerank(self, iterations, alpha=0.3, pr0=None, degree=None):
        """Returns (pagerank[nv], iterations_done, changed_hist)."""
        pr = np.full(self.nv, np.float32(0.3), np.float32) if pr0 is None else np.array(pr0, np.float32)
        deg = self.degree() if degree is None else np.ascontiguousarray(degree, np.int32)
        hist = np.full(4096, -1, np.int32)
        it = self._L.gmo_pagerank(self.h, alpha, iterations, pr, deg, hist.ctypes.data_as(C.c_void_p), hist.size)
        return pr, it, hist[: min(it, hist.size)].copy()

    def bfs(self, source):
        depth = np.zeros(self.nv, np.uint32)
        parent = np.zeros(self.nv, np.uint64)
        hist = np.full(4096, -1, np.int32)
        it = self._L.gmo_bfs(self.h, int(source), depth, parent, hist.ctypes.data_as(C.c_void_p), hist.size)
        return depth, parent, it, hist[: min(it, hist.size)].copy()

    def sssp(self, source):
        dist = np.zeros(self.nv, np.uint32)
        it = self._L.gmo_sssp(self.h, int(source), dist)
        return dist, it

    def sgd(self, lv, lam, step, iterations):
        lv = np.array(lv, copy=True, order="C")
        K = lv.shape[1]
        fn = self._L.gmo_sgd_f64 if lv.dtype == np.float64 else self._L.gmo_sgd_f32
        it = fn(self.h, K, lam, step, iterations, lv)
        if it < 0:
            raise ValueError("unsupported K")
        return lv, it

    def rmse_sum(self, lv):
        lv = np.ascontiguousarray(lv)
        K = lv.shape[1]
        sq = np.zeros(self.nv, lv.dtype)
        fn = self._L.gmo_rmse_f64 if lv.dtype == np.float64 else self._L.gmo_rmse_f32
        s = fn(self.h, K, lv, sq.ctypes.data_as(C.c_void_p))
        return s, sq

    def spmv_f64(self, x, xmask, transpose):
        x = np.ascontiguousarray(x, np.float64)
        xm = np.ascontiguousarray(xmask, np.uint8)
        y = np.zeros(self.nv, np.float64)
        ym = np.zeros(self.nv, np.uint8)
        self._L.gmo_spmv_f64(self.h, int(transpose), x, xm, y, ym)
        return y, ym


def vertex_to_native(v1, nparts, n):
    return lib().gmo_vertex_to_native(int(v1), int(nparts), int(n))


def native_to_vertex(v1, nparts, n):
    return lib().gmo_native_to_vertex(int(v1), int(nparts), int(n))


def mapreduce_double_sum(values, present, nthreads=1, init=0):
    """test/test_reduce.cpp's MapReduce (map 2a, reduce a+b) over the present entries of an int vector."""
    v = np.ascontiguousarray(values, np.int32)
    m = np.ascontiguousarray(present, np.uint8)
    return int(lib().gmo_mapreduce_double_sum(v, m, len(v), int(nthreads), int(init)))
