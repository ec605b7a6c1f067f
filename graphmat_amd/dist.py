"""Multi-GPU glue: 1-D vertex-range sharding over one process per GPU.

The reference distributes a 2-D tile grid over MPI ranks and exchanges x/y segments
with point-to-point messages (include/GMDP/multinode/spmspv.h:41-206).  Here each
GPU owns a contiguous range of native rows (complete rows, so there is no y
reduction) and the only per-iteration exchange is making the message vector x
globally visible: every rank contributes its slice (values + presence words), i.e.
an all-gather, done with torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Convergence is a 1-int MIN all-reduce
(include/GraphMatRuntime.h:226 analogue).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib

ALIGN = 64  # shard boundaries are multiples of 64 rows (whole presence words)


def edge_balanced_ranges(row_counts_cumsum, nv, nranks, align=ALIGN):
    """Split native rows [0,nv) into nranks contiguous ranges with ~equal edge counts.

    row_counts_cumsum[i] = number of edges in rows < i (length nv+1, numpy or torch on cpu).
    Boundaries are rounded to multiples of `align`; every range is non-empty where possible.
    """
    cs = np.asarray(row_counts_cumsum)
    total = int(cs[-1])
    bounds = [0]
    for r in range(1, nranks):
        target = total * r // nranks
        b = int(np.searchsorted(cs, target, side="left"))
        b = (b + align // 2) // align * align
        b = max(b, bounds[-1] + align) if bounds[-1] + align <= nv else nv
        b = min(b, nv)
        bounds.append(b)
    bounds.append(nv)
    for i in range(1, len(bounds)):  # keep monotone
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[i], bounds[i + 1]) for i in range(nranks)]


class MessageExchange:
    """Implements the library's exchange callback with torch.distributed.

    x_bytes: uint8 tensor of nv*elt_bytes (+pad) that the library uses as its x values
    buffer; x_bits: int32 tensor of (nv+31)//32 (+2) presence words.  Both are adopted by
    the graph (gm_graph_adopt_workspace), so the pointers the callback receives are these.
    """

    def __init__(self, ranges, rank, x_bytes, x_bits, group=None, live_rows=None, x_bytes2=None):
        """live_rows: only the first live_rows entries of every slice need to travel (the
        library's gm_graph_desc_t.xchg_rows); None = whole slices.
        x_bytes2: second message buffer (adopted as workspace slot 9) for the overlapped two-stage
        schedule (GM_XCHG_PART / GM_XCHG_WAIT)."""
        self.ranges = [(int(a), int(b)) for a, b in ranges]
        self.live = None if live_rows is None else int(live_rows)
        self.rank = rank
        self.x_bytes = x_bytes
        self.x_bits = x_bits
        self.group = group
        self.flag = torch.zeros(1, dtype=torch.int32, device=x_bytes.device)
        self.calls = 0
        self.no_fast_path = False
        self._stage = None
        self._stage_bits = None
        self.x_bytes2 = x_bytes2
        self.parts = 0
        self._pending = []       # (work handle, deferred copy or None)
        self._part_stage = None
        self.gather_buf = None   # adopted as workspace slot GM_WS_GATHER: blocks of (device id, message) entries
        self.sparse_gathers = 0
        self.sparse_bytes = 0

    def _buffer_of(self, d_ptr):
        if d_ptr == self.x_bytes.data_ptr():
            return self.x_bytes
        if self.x_bytes2 is not None and d_ptr == self.x_bytes2.data_ptr():
            return self.x_bytes2
        return None

    def start_part(self, buf, first, count, elt_bytes):
        """Every shard's rows [first, first+count) of its slice of `buf` are to reach all shards;
        returns once the transfer has been started."""
        n = len(self.ranges)
        S = self.ranges[0][1] - self.ranges[0][0]
        self.parts += 1
        if self._equal_slices() and dist.get_backend(self.group) == "nccl" and not self.no_fast_path:
            try:
                total = S if self.live is None else min(self.live, S)
                need = n * total * elt_bytes
                if self._part_stage is None or self._part_stage.numel() < need:
                    self._part_stage = torch.empty(need, dtype=torch.uint8, device=buf.device)
                # each part owns its own region of the staging buffer
                st = self._part_stage[n * first * elt_bytes: n * (first + count) * elt_bytes]
                lo = self.rank * S + first
                w = dist.all_gather_into_tensor(st, buf[lo * elt_bytes: (lo + count) * elt_bytes], group=self.group,
                                                async_op=True)

                def scatter(st=st, buf=buf, first=first, count=count):
                    xv = buf[: n * S * elt_bytes].view(n, S * elt_bytes)
                    xv[:, first * elt_bytes: (first + count) * elt_bytes].copy_(st.view(n, count * elt_bytes))
                self._pending.append((w, scatter))
                return
            except Exception as e:
                print("graphmat_amd.dist: WARNING: async all_gather_into_tensor failed (%r); falling back to per-slice broadcasts "
                      "(slower; MessageExchange.no_fast_path is set and bench.py reports it)" % (e,), flush=True)
                self.no_fast_path = True
        for r in range(n):
            lo = r * S + first
            self._pending.append((dist.broadcast(buf[lo * elt_bytes: (lo + count) * elt_bytes], src=r, group=self.group,
                                                 async_op=True), None))

    def wait_parts(self):
        pending, self._pending = self._pending, []
        for w, _ in pending:
            w.wait()
        for _, after in pending:
            if after is not None:
                after()

    def _equal_slices(self):
        n = len(self.ranges)
        S = self.ranges[0][1] - self.ranges[0][0]
        return S > 0 and all(lo == r * S and hi == (r + 1) * S for r, (lo, hi) in enumerate(self.ranges)) and n > 1

    def all_gather_slices(self, elt_bytes):
        if self._equal_slices() and dist.get_backend(self.group) == "nccl" and not self.no_fast_path:
            # equal slices (GM_LAYOUT_DEGREE): one in-place all-gather per array
            try:
                n = len(self.ranges)
                S = self.ranges[0][1]
                L = S if self.live is None else min(self.live, S)
                if L == S:
                    lo = self.rank * S
                    dist.all_gather_into_tensor(self.x_bytes[: n * S * elt_bytes],
                                                self.x_bytes[lo * elt_bytes: (lo + S) * elt_bytes], group=self.group)
                    W = S // 32
                    dist.all_gather_into_tensor(self.x_bits[: n * W], self.x_bits[self.rank * W: (self.rank + 1) * W],
                                                group=self.group)
                else:
                    # only the live prefix of every slice travels: gather into a compact staging
                    # buffer, then scatter the prefixes back to their slice positions
                    if self._stage is None or self._stage.numel() < n * L * elt_bytes:
                        self._stage = torch.empty(n * L * max(elt_bytes, 8), dtype=torch.uint8, device=self.x_bytes.device)
                        self._stage_bits = torch.empty(n * (L // 32), dtype=torch.int32, device=self.x_bytes.device)
                    lo = self.rank * S
                    st = self._stage[: n * L * elt_bytes]
                    dist.all_gather_into_tensor(st, self.x_bytes[lo * elt_bytes: (lo + L) * elt_bytes], group=self.group)
                    xv = self.x_bytes[: n * S * elt_bytes].view(n, S * elt_bytes)
                    xv[:, : L * elt_bytes].copy_(st.view(n, L * elt_bytes))
                    W, WL = S // 32, L // 32
                    sb = self._stage_bits[: n * WL]
                    dist.all_gather_into_tensor(sb, self.x_bits[self.rank * W: self.rank * W + WL], group=self.group)
                    self.x_bits[: n * W].view(n, W)[:, :WL].copy_(sb.view(n, WL))
                return
            except Exception as e:  # fall back to per-slice broadcasts (same result)
                print("graphmat_amd.dist: WARNING: all_gather_into_tensor path failed (%r); falling back to per-slice broadcasts "
                      "(slower; MessageExchange.no_fast_path is set and bench.py reports it)" % (e,), flush=True)
                self.no_fast_path = True
        hs = []
        for r, (lo, hi) in enumerate(self.ranges):
            if self.live is not None:
                hi = min(hi, lo + self.live)
            if hi <= lo:
                continue
            hs.append(dist.broadcast(self.x_bytes[lo * elt_bytes: hi * elt_bytes], src=r, group=self.group,
                                     async_op=True))
            hs.append(dist.broadcast(self.x_bits[lo // 32: (hi + 31) // 32], src=r, group=self.group, async_op=True))
        for h in hs:
            h.wait()

    def all_reduce_converged(self, value):
        self.flag[0] = int(value)
        dist.all_reduce(self.flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(self.flag.item())

    def exchange_state(self, converged, count):
        """convergence flag (AND) and active-set sizes (max, sum) over the shards: one all-gather of two ints"""
        n = len(self.ranges)
        mine = torch.tensor([converged, count], dtype=torch.int32, device=self.flag.device)
        parts = [torch.empty_like(mine) for _ in range(n)]
        dist.all_gather(parts, mine, group=self.group)
        allv = torch.stack(parts).cpu()
        conv = int(bool((allv[:, 0] != 0).all()))
        return conv, int(allv[:, 1].max()), min(int(allv[:, 1].to(torch.int64).sum()), 0x7FFFFFFF)

    def gather_blocks(self, block_bytes):
        """in-place all-gather of equal blocks of the adopted gather buffer (sparse message exchange)"""
        n = len(self.ranges)
        buf = self.gather_buf
        self.sparse_gathers += 1
        self.sparse_bytes += block_bytes
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(buf[: n * block_bytes], buf[self.rank * block_bytes: (self.rank + 1) * block_bytes],
                                        group=self.group)
            return
        hs = [dist.broadcast(buf[r * block_bytes: (r + 1) * block_bytes], src=r, group=self.group, async_op=True) for r in range(n)]
        for h in hs:
            h.wait()

    def callback(self):
        def fn(ctx, kind, d_ptr, elt_bytes, d_bits, h_flag):
            try:
                self.calls += 1
                if kind == _lib.GM_XCHG_MESSAGES:
                    if d_ptr != self.x_bytes.data_ptr() or d_bits != self.x_bits.data_ptr():
                        return 2
                    self.all_gather_slices(int(elt_bytes))
                elif kind == _lib.GM_XCHG_PART:
                    buf = self._buffer_of(d_ptr)
                    if buf is None:
                        return 2
                    self.start_part(buf, int(h_flag[0]), int(h_flag[1]), int(elt_bytes))
                elif kind == _lib.GM_XCHG_WAIT:
                    self.wait_parts()
                elif kind == _lib.GM_XCHG_CONVERGED:
                    h_flag[0] = self.all_reduce_converged(h_flag[0])
                elif kind == _lib.GM_XCHG_STATE:
                    conv, mx, total = self.exchange_state(int(h_flag[0]), int(h_flag[1]))
                    h_flag[0], h_flag[1], h_flag[2] = conv, mx, total
                elif kind == _lib.GM_XCHG_GATHER:
                    if self.gather_buf is None or d_ptr != self.gather_buf.data_ptr():
                        return 2
                    self.gather_blocks(int(h_flag[0]) * int(elt_bytes))
                else:
                    return 3
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("graphmat_amd.dist: exchange failed:", repr(e), flush=True)
                return 1
        return _lib.EXCHANGE_FN(fn)


def init_native_rccl(group=None, device=None):
    """Create the library's own RCCL communicator (graphmat_hip.h gm_dist_init) over the ranks of `group`:
    rank 0 draws the unique id, torch.distributed carries the 128 bytes to the others (any backend)."""
    import ctypes as C
    L = _lib.lib()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    raw = (C.c_ubyte * 128)()
    if rank == 0:
        _lib.check(L.gm_dist_unique_id(raw, 128))
    if world > 1:
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.tensor(list(raw), dtype=torch.uint8, device=(device if on_gpu else "cpu"))
        dist.broadcast(t, src=0, group=group)
        raw = (C.c_ubyte * 128)(*t.cpu().tolist())
    _lib.check(L.gm_dist_init(rank, world, raw, 128))
    return rank, world


def attach_native_exchange(g, group=None):
    """Sharded api.Graph over the library's native RCCL exchange (no Python per iteration).
    init_native_rccl() must have been called.  Installs the result-gather function like attach_exchange."""
    L = _lib.lib()
    _lib.check(L.gm_graph_use_rccl(g.h))
    world = dist.get_world_size(group) if dist.is_initialized() else 1

    def gather(t_rows):
        if world == 1:
            return t_rows
        parts = [torch.empty_like(t_rows) for _ in range(world)]
        dist.all_gather(parts, t_rows, group=group)
        return torch.cat(parts, 0)

    g.gather_fn = gather


def exchange_counters(g):
    """(exchange calls, overlapped parts started, bytes this rank contributed) of a native exchange."""
    import ctypes as C
    out = (C.c_int64 * 4)()
    _lib.check(_lib.lib().gm_graph_exchange_counters(g.h, out))
    return int(out[0]), int(out[1]), int(out[2])


def attach_exchange(g, group=None, max_elt_bytes=8, overlap=True):
    """Make a sharded api.Graph (GM_LAYOUT_DEGREE, nshards = world size) exchange its messages
    over torch.distributed: allocates the global x buffers, hands them to the library, installs
    the callback and a gather function for results.  overlap: also give the library a second
    message buffer so fixed-count ALL_VERTICES programs can exchange one stage's messages while
    the other stage computes."""
    import ctypes as C
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert g.nshards == world and g.shard == rank
    S = g.row_hi - g.row_lo
    ranges = [(r * S, (r + 1) * S) for r in range(world)]
    x_bytes = torch.zeros(g.ndevice * max_elt_bytes + 64, dtype=torch.uint8, device=g.device)
    x_bits = torch.zeros((g.ndevice + 31) // 32 + 2, dtype=torch.int32, device=g.device)
    L = _lib.lib()
    _lib.check(L.gm_graph_adopt_workspace(g.h, 1, x_bytes.data_ptr(), x_bytes.numel()))
    _lib.check(L.gm_graph_adopt_workspace(g.h, 2, x_bits.data_ptr(), x_bits.numel() * 4))
    x_bytes2 = None
    if overlap:
        x_bytes2 = torch.zeros(g.ndevice * max_elt_bytes + 64, dtype=torch.uint8, device=g.device)
        _lib.check(L.gm_graph_adopt_workspace(g.h, 9, x_bytes2.data_ptr(), x_bytes2.numel()))
    ex = MessageExchange(ranges, rank, x_bytes, x_bits, group, live_rows=g.xchg_rows, x_bytes2=x_bytes2)
    cb = ex.callback()
    g._cb = (cb, ex)  # keep alive
    _lib.check(L.gm_graph_set_exchange(g.h, cb, None))
    if max_elt_bytes <= 8:
        # sparse exchange of small active sets: world blocks of up to 65536 entries of (4 + 8) bytes
        ex.gather_buf = torch.zeros(world * 65536 * 12 + 256, dtype=torch.uint8, device=g.device)
        _lib.check(L.gm_graph_adopt_workspace(g.h, _lib.GM_WS_GATHER, ex.gather_buf.data_ptr(), ex.gather_buf.numel()))
        _lib.check(L.gm_graph_set_exchange_caps(g.h, _lib.GM_XCAP_SPARSE))

    def gather(t_rows):
        parts = [torch.empty_like(t_rows) for _ in range(world)]
        dist.all_gather(parts, t_rows, group=group)
        return torch.cat(parts, 0)

    g.gather_fn = gather
    return ex
