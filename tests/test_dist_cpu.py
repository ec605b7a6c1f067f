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
