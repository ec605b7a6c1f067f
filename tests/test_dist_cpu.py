"""world_size-2 gloo tests of the multi-GPU host logic (no GPU): edge-balanced row ranges
and the slice all-gather / convergence all-reduce used as the library's exchange callback."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graphmat_amd import generators as gen
from graphmat_amd.dist import ALIGN, MessageExchange, edge_balanced_ranges


def test_edge_balanced_ranges_properties():
    nv, s, d, v = gen.rmat_edges(14, 16, seed=2)
    indeg = np.bincount(d - 1, minlength=nv)
    cs = np.concatenate([[0], np.cumsum(indeg)])
    for n in (1, 2, 4, 8):
        rs = edge_balanced_ranges(cs, nv, n)
        assert rs[0][0] == 0 and rs[-1][1] == nv and len(rs) == n
        for (a, b), (c, _) in zip(rs[:-1], rs[1:]):
            assert b == c and a % ALIGN == 0 and b % ALIGN == 0 and a <= b
        edges = np.array([cs[b] - cs[a] for a, b in rs])
        assert edges.sum() == len(s)
        if n > 1:  # skewed graph: vertex-equal split would give >40% to rank 0
            assert edges.max() <= 1.35 * len(s) / n + indeg.max()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nv, ranges, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elt = 4
    x = torch.zeros(nv * elt + 16, dtype=torch.uint8)
    bits = torch.zeros((nv + 31) // 32 + 2, dtype=torch.int32)
    lo, hi = ranges[rank]
    xv = x[: nv * elt].view(torch.float32)
    xv[lo:hi] = torch.arange(lo, hi, dtype=torch.float32) + 0.5   # own slice only
    bits[lo // 32: (hi + 31) // 32] = rank + 1
    ex = MessageExchange(ranges, rank, x, bits)
    ex.all_gather_slices(elt)
    ok = bool((xv == torch.arange(nv, dtype=torch.float32) + 0.5).all())
    for r, (a, b) in enumerate(ranges):
        ok = ok and bool((bits[a // 32: (b + 31) // 32] == r + 1).all())
    # convergence: AND over ranks
    c1 = ex.all_reduce_converged(1)
    c2 = ex.all_reduce_converged(1 if rank == 0 else 0)
    out[rank] = int(ok and c1 == 1 and c2 == 0)
    dist.destroy_process_group()


def test_exchange_world2_gloo():
    nv = 64 * 37
    ranges = [(0, 64 * 5), (64 * 5, nv)]  # deliberately unequal slices
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, nv, ranges, out), nprocs=2, join=True)
    assert out[0] == 1 and out[1] == 1


def _worker_parts(rank, world, port, S, out):
    """two-stage overlapped exchange (GM_XCHG_PART / GM_XCHG_WAIT): parts of a second buffer, started
    asynchronously, complete at wait_parts()"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elt = 4
    nv = S * world
    ranges = [(r * S, (r + 1) * S) for r in range(world)]
    x = torch.zeros(nv * elt + 16, dtype=torch.uint8)
    x2 = torch.zeros(nv * elt + 16, dtype=torch.uint8)
    bits = torch.zeros((nv + 31) // 32 + 2, dtype=torch.int32)
    live = S - 64  # the last 64 rows of every slice never travel
    ex = MessageExchange(ranges, rank, x, bits, live_rows=live, x_bytes2=x2)
    xv2 = x2[: nv * elt].view(torch.float32)
    want = torch.zeros(nv, dtype=torch.float32)
    split = 128
    for r in range(world):
        want[r * S: r * S + live] = torch.arange(r * S, r * S + live, dtype=torch.float32) * 2 + 1
    lo = rank * S
    # stage 1: tail rows [split, live) of the own slice, then stage 2: head rows [0, split)
    xv2[lo + split: lo + live] = want[lo + split: lo + live]
    ex.start_part(x2, split, live - split, elt)
    xv2[lo: lo + split] = want[lo: lo + split]
    ex.start_part(x2, 0, split, elt)
    ex.wait_parts()
    ok = bool((xv2 == want).all()) and ex.parts == 2 and not ex._pending
    ok = ok and bool((x == 0).all())  # the first buffer was not touched
    out[rank] = int(ok)
    dist.destroy_process_group()


def test_partial_exchange_world3_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_parts, args=(3, _free_port(), 64 * 6, out), nprocs=3, join=True)
    assert out[0] == 1 and out[1] == 1 and out[2] == 1


def _worker_sparse(rank, world, port, out):
    """sparse exchange of small active sets (GM_XCHG_STATE / GM_XCHG_GATHER): the fused flag + sizes
    all-gather and the in-place all-gather of equal blocks of (device id, message) entries"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = 256
    nv = S * world
    ranges = [(r * S, (r + 1) * S) for r in range(world)]
    ex = MessageExchange(ranges, rank, torch.zeros(nv * 8 + 16, dtype=torch.uint8), torch.zeros(nv // 32 + 2, dtype=torch.int32))
    # flag AND, sizes max / sum
    conv, mx, total = ex.exchange_state(1 if rank != 1 else 0, 10 * (rank + 1))
    ok = conv == 0 and mx == 10 * world and total == 10 * world * (world + 1) // 2
    conv, mx, total = ex.exchange_state(1, 0)
    ok = ok and conv == 1 and mx == 0 and total == 0
    # blocks of 64 entries of 12 bytes (int32 id + 8 message bytes): every rank fills its own block
    entry = np.dtype([("idx", np.int32), ("msg", np.uint64)], align=False)
    assert entry.itemsize == 12
    cap = 64
    ex.gather_buf = torch.zeros(world * cap * 12 + 64, dtype=torch.uint8)
    mine = np.zeros(cap, entry)
    mine["idx"] = -1
    k = 5 + rank
    mine["idx"][:k] = rank * S + np.arange(k) * 3
    mine["msg"][:k] = 1000 * (rank + 1) + np.arange(k)
    ex.gather_buf[rank * cap * 12: (rank + 1) * cap * 12] = torch.from_numpy(mine.view(np.uint8).copy())
    ex.gather_blocks(cap * 12)
    allv = ex.gather_buf[: world * cap * 12].numpy().view(entry).reshape(world, cap)
    for r in range(world):
        kr = 5 + r
        ok = ok and (allv[r]["idx"][:kr] == r * S + np.arange(kr) * 3).all() and (allv[r]["idx"][kr:] == -1).all()
        ok = ok and (allv[r]["msg"][:kr] == 1000 * (r + 1) + np.arange(kr)).all()
    ok = ok and ex.sparse_gathers == 1 and ex.sparse_bytes == cap * 12
    out[rank] = int(bool(ok))
    dist.destroy_process_group()


def test_sparse_exchange_world3_gloo():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_sparse, args=(3, _free_port(), out), nprocs=3, join=True)
    assert out[0] == 1 and out[1] == 1 and out[2] == 1
