"""Host-side mirror of the GraphMat application surface over the C-ABI.

`Graph` plays the role of GraphMat::Graph<V,E> + the run_* drivers of the
reference's example applications (src/PageRank.cpp:115-161, src/BFS.cpp:110-156,
src/SSSP.cpp:99-125, src/SGD.cpp:163-224): it owns the device adjacency, keeps
vertex state in HBM (torch tensors are used purely as device buffers) and calls
the fixed-menu programs of libgraphmat_hip.so.  Per-vertex results are returned
in ORIGINAL vertex order (index v-1 for vertex id v), like getVertexproperty(v).
"""
import ctypes as C

import time

import numpy as np
import torch

from . import _lib
from ._lib import GM_DIR_IN, GM_DIR_OUT, GM_LAYOUT_DEGREE, GM_LAYOUT_NATIVE, check

MAX_DIST = 0xFFFFFFFF


def native_index(nv, nparts):
    """native0[v-1] for v = 1..nv: the permutation of include/Graph.h:111-130 (vectorised)."""
    v = np.arange(nv, dtype=np.int64)
    h = nv // nparts
    vmax = h * nparts
    nat = np.where(v >= vmax, v, (v // nparts) + (v % nparts) * h)
    return nat


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Graph:
    def __init__(self, nv, src, dst, val=None, ref_threads=1, directions=GM_DIR_OUT | GM_DIR_IN, device=0,
                 row_range=None, keep_values=True, nranks_layout=1, layout=GM_LAYOUT_DEGREE, nshards=1, shard=0,
                 col_tiles=0, edges_local=False):
        """src/dst: 1-based ids, numpy (host) or torch.cuda int32 tensors (device).
        edges_local: src/dst/val are this rank's PART of the edge list (distributed build over the
        library's communicator, dist.init_native_rccl; graphmat_hip.h gm_graph_desc_t.edges_local).

        layout GM_LAYOUT_DEGREE (default): the library picks the device order (degree-ranked,
        dealt over `nshards`); GM_LAYOUT_NATIVE: device order = native order, optional
        row_range=(lo,hi) shard.  Results are identical in both.
        col_tiles: column tiles of the OUT adjacency (graphmat_hip.h gm_graph_tile): 0 = library
        default, 1 = none, N = that many; results are identical for every value."""
        if not torch.cuda.is_available():
            raise RuntimeError("graphmat_amd needs a GPU (no CPU fallback)")
        self.L = _lib.lib()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        check(self.L.gm_set_device(device))
        self.nv = int(nv)
        self.nparts = int(ref_threads) * 16 * int(nranks_layout)
        lo, hi = (0, self.nv) if row_range is None else row_range
        on_dev = isinstance(src, torch.Tensor)
        if on_dev:
            assert src.is_cuda and src.dtype == torch.int32 and dst.dtype == torch.int32
            src = src.contiguous()
            dst = dst.contiguous()
            sp, dp = src.data_ptr(), dst.data_ptr()
            vp = val.contiguous().data_ptr() if (val is not None and keep_values) else None
            nnz = src.numel()
        else:
            src = np.ascontiguousarray(src, np.int32)
            dst = np.ascontiguousarray(dst, np.int32)
            sp, dp = src.ctypes.data, dst.ctypes.data
            if val is not None and keep_values:
                val = np.ascontiguousarray(val, np.int32)
                vp = val.ctypes.data
            else:
                vp = None
            nnz = src.size
        self.nnz_input = int(nnz)
        d = _lib.GraphDesc(self.nv, self.nparts, int(lo), int(hi), directions, 4 if vp else 0,
                           1 if on_dev else 0, 0, layout, nshards, shard, 0, 0, int(col_tiles), 1 if edges_local else 0)
        h = C.c_void_p()
        check(self.L.gm_graph_create(C.byref(h), C.byref(d), nnz, sp, dp, vp, _stream()))
        self.h = h
        check(self.L.gm_graph_desc(self.h, C.byref(d)))
        self.row_lo, self.row_hi, self.ndevice = d.row_lo, d.row_hi, d.ndevice
        self.xchg_rows = d.xchg_rows
        self.col_tiles = d.col_tiles
        self.rows = self.row_hi - self.row_lo
        self.layout, self.nshards, self.shard = layout, nshards, shard
        self._dov = None
        self._cb = None
        self.gather_fn = None  # sharded graphs: local rows tensor -> all rows tensor (see dist.attach_exchange)

    def close(self):
        if getattr(self, "h", None):
            self.L.gm_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- id spaces ---------------------------------------------------------------------
    def maps_to_host(self):
        """(dev_of_native[nv], native_of_dev[ndevice]) as numpy int32."""
        don = np.zeros(self.nv, np.int32)
        nod = np.zeros(self.ndevice, np.int32)
        check(self.L.gm_graph_maps_to_host(self.h, don.ctypes.data, nod.ctypes.data))
        return don, nod

    @property
    def dev_of_vertex(self):
        """device slot of vertex v (index v-1): vertex id -> native id -> device id."""
        if self._dov is None:
            don, _ = self.maps_to_host()
            self._dov = torch.from_numpy(don.astype(np.int64)[native_index(self.nv, self.nparts)]).to(self.device)
        return self._dov

    def to_device_order(self, arr_vertex_order, fill=0):
        """array indexed by vertex-1 -> device tensor over THIS shard's rows (device order)."""
        t = torch.as_tensor(arr_vertex_order).to(self.device)
        out = torch.full((self.ndevice,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=self.device)
        out[self.dev_of_vertex] = t
        return out[self.row_lo:self.row_hi].contiguous()

    def to_vertex_order(self, t_rows):
        """tensor over this shard's rows -> tensor indexed by vertex-1 (gathers the shards first)."""
        if self.rows != self.ndevice:
            assert self.gather_fn is not None, "sharded graph: attach an exchange (dist.attach_exchange) first"
            t_rows = self.gather_fn(t_rows.contiguous())
        return t_rows[self.dev_of_vertex]

    def _local_slot(self, vertex):
        """row index inside this shard of a vertex, or None if another shard owns it"""
        d = int(self.dev_of_vertex[vertex - 1])
        return d - self.row_lo if self.row_lo <= d < self.row_hi else None

    def csr(self, direction):
        c = _lib.Csr()
        check(self.L.gm_graph_csr(self.h, direction, C.byref(c)))
        return c

    def tile(self, direction, t):
        """(Csr view of column tile t, device pointer of the presence bits of rows already started)."""
        c = _lib.Csr()
        prev = C.c_void_p()
        check(self.L.gm_graph_tile(self.h, direction, t, C.byref(c), C.byref(prev)))
        return c, prev.value

    def csr_to_host(self, direction):
        c = self.csr(direction)
        rp = np.zeros(c.nrows + 1, np.int64)
        ci = np.zeros(max(c.nnz, 1), np.int32)
        vv = np.zeros(max(c.nnz, 1), np.int32)
        check(self.L.gm_graph_csr_to_host(self.h, direction, rp.ctypes.data, ci.ctypes.data,
                                          vv.ctypes.data if c.vals else None))
        return rp, ci[: c.nnz], (vv[: c.nnz] if c.vals else None)

    def enable_timing(self, on=True):
        check(self.L.gm_graph_enable_timing(self.h, 1 if on else 0))

    def last_stats(self):
        s = _lib.RunStats()
        check(self.L.gm_graph_last_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    # ---- PageRank ------------------------------------------------------------------------
    def new_pr_state(self):
        """PR{pagerank=0.3f, degree=0} for every row of this shard; [rows,2] int32 view of gm_pr_t."""
        st = torch.zeros((self.rows, 2), dtype=torch.int32, device=self.device)
        st[:, 0] = int(np.float32(0.3).view(np.int32))
        return st

    def run_degree(self, state):
        it = C.c_int(0)
        check(self.L.gm_run_degree(self.h, state.data_ptr(), 1, C.byref(it), _stream()))
        return it.value

    def run_pagerank(self, state, iterations, alpha=0.3):
        it = C.c_int(0)
        check(self.L.gm_run_pagerank(self.h, state.data_ptr(), alpha, iterations, C.byref(it), _stream()))
        return it.value

    def pagerank(self, iterations, alpha=0.3):
        """Degree pass then PageRank, like run_pagerank() of the reference app.
        Returns (pagerank float32[nv], out_degree int32[nv], iterations_done), vertex order."""
        st = self.new_pr_state()
        self.run_degree(st)
        it = self.run_pagerank(st, iterations, alpha)
        pr = self.to_vertex_order(st[:, 0].contiguous().view(torch.float32)).cpu().numpy()
        deg = self.to_vertex_order(st[:, 1].contiguous()).cpu().numpy()
        return pr, deg, it

    # ---- BFS -----------------------------------------------------------------------------
    def bfs(self, source):
        """Returns (depth uint32[nv], parent uint64[nv], iterations) in vertex order."""
        n = self.rows
        st = torch.zeros((n, 3), dtype=torch.int64, device=self.device)  # gm_bfs_t = 24 bytes
        st[:, 0] = MAX_DIST          # depth (low 32 bits), pad = 0
        st[:, 1] = -1                # parent
        ids = torch.arange(1, self.nv + 1, dtype=torch.int64, device=self.device)
        st[:, 2] = self.to_device_order(ids)  # id of the vertex living in each device slot
        act = torch.zeros((n + 31) // 32 + 2, dtype=torch.int32, device=self.device)
        s_nat = self._local_slot(source)
        if s_nat is not None:
            st[s_nat, 0] = 0
            act[s_nat >> 5] = int(np.uint32(1 << (s_nat & 31)).view(np.int32))
        it = C.c_int(0)
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        check(self.L.gm_run_bfs(self.h, st.data_ptr(), act.data_ptr(), -1, C.byref(it), _stream()))
        torch.cuda.synchronize(self.device)
        self.last_wall_ms = (time.perf_counter() - t0) * 1e3  # the whole call: set-up passes, every level, host syncs
        depth = self.to_vertex_order(st[:, 0] & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)
        parent = self.to_vertex_order(st[:, 1].contiguous()).cpu().numpy().view(np.uint64)
        return depth, parent, it.value

    # ---- SSSP ----------------------------------------------------------------------------
    def sssp(self, source):
        n = self.rows
        dist = torch.full((n,), -1, dtype=torch.int32, device=self.device)  # 0xFFFFFFFF
        act = torch.zeros((n + 31) // 32 + 2, dtype=torch.int32, device=self.device)
        s_nat = self._local_slot(source)
        if s_nat is not None:
            dist[s_nat] = 0
            act[s_nat >> 5] = int(np.uint32(1 << (s_nat & 31)).view(np.int32))
        it = C.c_int(0)
        check(self.L.gm_run_sssp(self.h, dist.data_ptr(), act.data_ptr(), -1, C.byref(it), _stream()))
        return self.to_vertex_order(dist).cpu().numpy().view(np.uint32), it.value

    # ---- SGD / RMSE ------------------------------------------------------------------------
    def _latent_to_device(self, lv):
        lv = np.ascontiguousarray(lv)
        K = lv.shape[1]
        full = np.zeros((self.nv, K + 1), lv.dtype)
        full[:, :K] = lv
        return self.to_device_order(full).contiguous(), K

    def sgd(self, lv, lam, step, iterations):
        st, K = self._latent_to_device(lv)
        it = C.c_int(0)
        check(self.L.gm_run_sgd(self.h, st.data_ptr(), K, st.element_size(), lam, step, iterations, C.byref(it),
                                _stream()))
        return self.to_vertex_order(st)[:, :K].cpu().numpy(), it.value

    def sgd_bipartite(self, lv, nusers, nitems, lam, step, iterations, blocks=4):
        """gm_run_sgd_bipartite: this graph holds the ratings of THIS rank's users only (contiguous native ranges of
        users, ascending with the rank of the library's communicator; users are vertices 1..nusers, items the next
        nitems ids); only the items' running sums travel.  lv: [nv, K] float32, all vertices.  Returns (latent after
        the run in vertex order -- current for the items and for this rank's users --, iterations done, bytes this
        rank received per iteration)."""
        st, K = self._latent_to_device(lv)
        item_rows = self.dev_of_vertex[nusers:nusers + nitems].to(torch.int32).contiguous()
        it = C.c_int(0)
        check(self.L.gm_run_sgd_bipartite(self.h, st.data_ptr(), K, st.element_size(), item_rows.data_ptr(), nitems, blocks, lam, step,
                                          iterations, C.byref(it), _stream()))
        moved = C.c_int64(0)
        self.L.gm_graph_note_get(self.h, 1, C.byref(moved))
        return self.to_vertex_order(st)[:, :K].cpu().numpy(), it.value, moved.value

    def rmse_sum(self, lv):
        """sum over vertices of sqerr after one RMSE pass (caller: sqrt(sum/nnz))."""
        st, K = self._latent_to_device(lv)
        check(self.L.gm_run_rmse(self.h, st.data_ptr(), K, st.element_size(), _stream()))
        out = C.c_double(0)
        fn = self.L.gm_reduce_sum_f64 if st.dtype == torch.float64 else self.L.gm_reduce_sum_f32
        check(fn(st.data_ptr() + K * st.element_size(), self.rows, K + 1, C.byref(out), _stream()))
        sq = self.to_vertex_order(st)[:, K].cpu().numpy()
        return out.value, sq


_hip = None


def copy_from_device(host_array, dev_ptr):
    """hipMemcpy of host_array.nbytes bytes from a raw device pointer (tests look at library-owned arrays)."""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    if host_array.nbytes == 0:
        return
    rc = _hip.hipMemcpy(host_array.ctypes.data, dev_ptr, host_array.nbytes, 2)  # hipMemcpyDeviceToHost
    if rc != 0:
        raise RuntimeError("hipMemcpy failed: %d" % rc)


def copy_to_device(dev_ptr, host_array):
    """hipMemcpy of host_array.nbytes bytes to a raw device pointer (tests rewrite library-owned arrays in place)."""
    if host_array.nbytes == 0:
        return
    copy_from_device(np.zeros(0, np.uint8), None)  # (binds libamdhip64)
    rc = _hip.hipMemcpy(dev_ptr, host_array.ctypes.data, host_array.nbytes, 1)  # hipMemcpyHostToDevice
    if rc != 0:
        raise RuntimeError("hipMemcpy failed: %d" % rc)


def uniform_on_device(scale, edge_factor=16, seed=1, device=0, part=None):
    """A uniform random graph generated in HBM: V = 2^scale vertices with exactly `edge_factor` out-edges each to
    uniformly drawn destinations (the shape of the reference's generate_random_edgelist, test/generator.h:73-105; at
    this size a row's destinations are distinct with probability 1 - 2e-6 per row, repeated draws are kept like RMAT's
    duplicate edges).  torch's device generator, seeded: reproducible on one GPU type, not bit-identical to numpy.
    part=(i, n): the i-th of n consecutive chunks of the edge list."""
    dev = torch.device("cuda", device)
    nv = 1 << scale
    total = edge_factor * nv
    first, ne = 0, total
    if part is not None:
        i, n = part
        first = total * i // n
        ne = total * (i + 1) // n - first
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) * 1000003 + first)
    src = (torch.arange(first, first + ne, dtype=torch.int64, device=dev) // edge_factor + 1).to(torch.int32)
    dst = torch.randint(1, nv + 1, (ne,), dtype=torch.int32, device=dev, generator=gen)
    return nv, src, dst, None


def rmat_on_device(scale, edge_factor=16, seed=1, weights=False, device=0, part=None):
    """RMAT edges generated in HBM (bit-identical to generators.rmat_edges).
    part=(i, n): only the i-th of n consecutive chunks of the edge list (a rank's part of a distributed build)."""
    L = _lib.lib()
    dev = torch.device("cuda", device)
    nv = 1 << scale
    total = edge_factor * nv
    first, ne = 0, total
    if part is not None:
        i, n = part
        first = total * i // n
        ne = total * (i + 1) // n - first
    src = torch.empty(ne, dtype=torch.int32, device=dev)
    dst = torch.empty(ne, dtype=torch.int32, device=dev)
    val = torch.empty(ne, dtype=torch.int32, device=dev) if weights else None
    check(L.gm_rmat_generate(scale, seed, first, ne, src.data_ptr(), dst.data_ptr(),
                             val.data_ptr() if weights else None, 1 if weights else 0, _stream()))
    return nv, src, dst, val
