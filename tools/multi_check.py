#!/usr/bin/env python3
"""Sharded run (one process per shard) checked against the oracle.

Launched by tests/test_gpu_multi.py as
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ... tools/multi_check.py
On a 1-GPU box every rank uses cuda:0 and the collectives run over gloo (NCCL refuses two ranks
on one device); on an N-GPU node set GM_BACKEND=nccl.  GM_EXCHANGE=native uses the library's own RCCL
exchange (gm_dist.hip) instead of the torch.distributed callback -- one rank per GPU, so on a 1-GPU box
that is a world of 1 (every code path of the exchange runs, with a single participant).
Prints MULTI_OK on rank 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("GM_BACKEND", "gloo")
    ndev = torch.cuda.device_count()
    device = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    torch.cuda.set_device(device)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from graphmat_amd import api, generators
    from graphmat_amd.dist import attach_exchange, attach_native_exchange, exchange_counters, init_native_rccl
    native = os.environ.get("GM_EXCHANGE", "callback") == "native"
    if native:
        init_native_rccl(device=torch.device("cuda", device))

    class NativeCounters:  # same read-outs as MessageExchange
        def __init__(self, g):
            self.g = g

        @property
        def parts(self):
            return exchange_counters(self.g)[1]

        @property
        def calls(self):
            return exchange_counters(self.g)[0]

    def attach(g, **kw):
        if native:
            attach_native_exchange(g)
            return NativeCounters(g)
        return attach_exchange(g, **kw)
    from oracle import binding as ob
    if os.environ.get("GM_FORCE_FORMS") == "1":
        # the persistent kernels (large LDS hot set dealt over the shards' slices) the library only picks for large graphs
        from graphmat_amd import _lib as _l
        _l.check(_l.lib().gm_set_option(b"wave16_form", 16 + 2))
        _l.check(_l.lib().gm_set_option(b"rowwave_form", 16 + 4))
    scale = int(os.environ.get("GM_SCALE", "14"))
    nv, s, d, v = generators.rmat_edges(scale, 16, seed=21, weights="hash")
    g = api.Graph(nv, s, d, v, ref_threads=2, device=device, layout=api.GM_LAYOUT_DEGREE, nshards=world, shard=rank)
    ex = attach(g)
    og = ob.OracleGraph(nv, s, d, v, ref_threads=2)
    ok = True
    pr, deg, it = g.pagerank(8)  # fixed count, ALL_VERTICES: the overlapped two-stage schedule
    opr, oit, _ = og.pagerank(8)
    if os.environ.get("GM_NEGATIVE") == "1":
        # negative control (GRAPHMAT_DEBUG_DROP_WAIT=1: gm_dist.hip leaves out the run stream's wait for the side stream's
        # all-gathers): with a stream-ordered transport the two-stage schedule must now read messages that have not arrived
        bad = int((pr.view(np.uint32) != opr.view(np.uint32)).sum())
        flag = torch.tensor([bad], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        if rank == 0:
            print("NEGATIVE_CAUGHT" if int(flag) > 0 else "NEGATIVE_MISSED", "world=%d parts=%d differing values=%d" % (world, ex.parts, int(flag)), flush=True)
        dist.destroy_process_group()
        sys.exit(0)
    ok &= bool((deg == og.degree()).all()) and it == oit and bool((pr.view(np.uint32) == opr.view(np.uint32)).all())
    staged = ex.parts
    ok &= staged == 2 * 7  # two parts per iteration but the last
    from graphmat_amd import _lib
    _lib.lib().gm_set_option(b"debug_flags", 128)  # same run through the plain loop
    pr_plain, _, _ = g.pagerank(8)
    _lib.lib().gm_set_option(b"debug_flags", 0)
    ok &= ex.parts == staged and bool((pr_plain.view(np.uint32) == opr.view(np.uint32)).all())
    if not ok:
        print("rank %d: fixed-count PageRank mismatch (parts=%d)" % (rank, ex.parts), flush=True)
    pr2, _, it2 = g.pagerank(-1)  # until convergence: exercises the flag all-reduce
    opr2, oit2, _ = og.pagerank(-1)
    ok &= it2 == oit2 and bool((pr2.view(np.uint32) == opr2.view(np.uint32)).all())
    sparse_levels = 0
    for src in (1, 7):
        depth, parent, itb = g.bfs(src)
        od, op, oitb, _ = og.bfs(src)
        this = itb == oitb and bool((depth == od).all()) and bool((parent == op).all())
        if not this:
            print("rank %d: BFS from %d differs from the oracle (%d vs %d levels, %d depth / %d parent mismatches)" % (
                rank, src, itb, oitb, int((depth != od).sum()), int((parent != op).sum())), flush=True)
        ok &= this
        sparse_levels += g.last_stats()["sparse_exchanges"]
    # the first and last levels of a traversal have small active sets: their messages travel as lists
    if sparse_levels < 2:
        print("rank %d: BFS never used the sparse exchange (%d levels)" % (rank, sparse_levels), flush=True)
        ok = False
    _lib.lib().gm_set_option(b"debug_flags", 2048)  # same traversal with dense exchanges only
    depth2, parent2, itb2 = g.bfs(7)
    _lib.lib().gm_set_option(b"debug_flags", 0)
    this = itb2 == itb and g.last_stats()["sparse_exchanges"] == 0 and bool((depth2 == depth).all()) and bool((parent2 == parent).all())
    if not this:
        print("rank %d: dense-only BFS differs (%d levels, %d sparse exchanges)" % (rank, itb2, g.last_stats()["sparse_exchanges"]), flush=True)
    ok &= this
    dist_, its = g.sssp(1)
    odist, oits = og.sssp(1)
    this = its == oits and bool((dist_ == odist).all())
    if not this:
        print("rank %d: SSSP differs from the oracle (%d vs %d iterations, %d mismatches)" % (rank, its, oits, int((dist_ != odist).sum())), flush=True)
    ok &= this
    if native:
        # distributed build: every rank passes an uneven, consecutive part of the edge list (a block of edges
        # appears twice with different values, the copies on different ranks); the shard it builds must equal,
        # array for array, the one built from the whole list
        dup = slice(len(s) // world - 40, len(s) // world + 40) if world > 1 else slice(100, 180)
        s2, d2 = np.concatenate([s, s[dup]]), np.concatenate([d, d[dup]])
        v2 = np.concatenate([v, (v[dup] + 3).astype(v.dtype)])
        cuts = [0] + [len(s2) * (r + 1) // world + (17 if r + 1 < world else 0) for r in range(world)]
        mine = slice(cuts[rank], cuts[rank + 1])
        g_all = api.Graph(nv, s2, d2, v2, ref_threads=2, device=device, layout=api.GM_LAYOUT_DEGREE, nshards=world, shard=rank)
        for variant in ("host", "device", "all-on-rank-0"):
            on_device = variant == "device"
            if variant == "all-on-rank-0":  # (what the C++ loader does with a single input file: the other ranks pass nothing)
                mine = slice(0, len(s2)) if rank == 0 else slice(0, 0)
            part = [torch.from_numpy(a[mine].copy()).to(torch.device("cuda", device)) if on_device else a[mine] for a in (s2, d2, v2)]
            g_loc = api.Graph(nv, part[0], part[1], part[2], ref_threads=2, device=device, layout=api.GM_LAYOUT_DEGREE, nshards=world,
                              shard=rank, edges_local=True)
            same = (g_loc.row_lo, g_loc.row_hi, g_loc.ndevice, g_loc.xchg_rows) == (g_all.row_lo, g_all.row_hi, g_all.ndevice, g_all.xchg_rows)
            same &= all(bool(np.array_equal(a, b)) for a, b in zip(g_loc.maps_to_host(), g_all.maps_to_host()))
            for direction in (api.GM_DIR_OUT, api.GM_DIR_IN):
                for a, b in zip(g_loc.csr_to_host(direction), g_all.csr_to_host(direction)):
                    same &= bool(np.array_equal(a, b))
            attach(g_loc)
            pr_loc, _, _ = g_loc.pagerank(4)
            if on_device:
                attach(g_all)
                pr_all, _, _ = g_all.pagerank(4)
                same &= bool((pr_loc.view(np.uint32) == pr_all.view(np.uint32)).all())
            if not same:
                print("rank %d: distributed build (%s) differs from the whole-list build" % (rank, variant), flush=True)
            ok &= same
            g_loc.close()
        g_all.close()
        # a bad edge id on ONE rank must fail the collective build on EVERY rank (not leave the others waiting)
        mine = slice(cuts[rank], cuts[rank + 1])
        sb = s2[mine].copy()
        if rank == world - 1 and sb.size:
            sb[sb.size // 2] = nv + 5
        try:
            api.Graph(nv, sb, d2[mine], v2[mine], ref_threads=2, device=device, layout=api.GM_LAYOUT_DEGREE, nshards=world, shard=rank, edges_local=True)
            refused = False
        except RuntimeError as e:
            refused = ("outside" in str(e)) == (rank == world - 1) or "other rank" in str(e)
        if not refused:
            print("rank %d: a distributed build with a bad id on the last rank was not refused everywhere" % rank, flush=True)
        ok &= refused
    # SGD / RMSE with K=128 fp32 latent vectors (BASELINE config 5 shape) on a sharded bipartite
    # ratings graph: the dedicated kernels exchange 512-byte x rows; bit-exact against the oracle
    rng = np.random.default_rng(7)
    nu, ni, nr, K = 700, 90, 9000, 128
    rs = rng.integers(1, nu + 1, nr).astype(np.int32)
    rd = (nu + rng.integers(1, ni + 1, nr)).astype(np.int32)
    rv = rng.integers(1, 6, nr).astype(np.int32)
    lv = rng.random((nu + ni, K)).astype(np.float32)
    g2 = api.Graph(nu + ni, rs, rd, rv, ref_threads=1, device=device, layout=api.GM_LAYOUT_DEGREE, nshards=world, shard=rank)
    ex2 = attach(g2, max_elt_bytes=K * 4)
    og2 = ob.OracleGraph(nu + ni, rs, rd, rv, 1)
    _, sq = g2.rmse_sum(lv)
    _, osq = og2.rmse_sum(lv)
    ok_sgd = bool(np.array_equal(sq, osq))
    lv2, it3 = g2.sgd(lv, 0.001, 1e-4, 3)
    olv2, oit3 = og2.sgd(lv, 0.001, 1e-4, 3)
    ok_sgd &= it3 == oit3 == 3 and bool(np.array_equal(lv2, olv2)) and not np.array_equal(lv2, lv)
    ok_sgd &= ex2.calls == 4  # one x exchange for the RMSE pass + one per SGD iteration
    if not ok_sgd:
        print("rank %d: sharded SGD mismatch (exchanges=%d)" % (rank, ex2.calls), flush=True)
    ok &= ok_sgd
    if native:
        # the same SGD with the ratings split by USERS (contiguous native ranges, one per rank) and only the items' running
        # sums travelling (gm_run_sgd_bipartite): the same bits, a fraction of the bytes
        nat = api.native_index(nu + ni, 16)          # ref_threads=1
        order = np.argsort(nat[:nu], kind="stable")  # users in ascending native id
        cuts = [nu * r // world for r in range(world + 1)]
        mine_users = np.zeros(nu + ni + 1, bool)
        mine_users[order[cuts[rank]:cuts[rank + 1]] + 1] = True
        sel = mine_users[rs]
        g3 = api.Graph(nu + ni, rs[sel], rd[sel], rv[sel], ref_threads=1, device=device)
        ok_bip = True
        for blocks in (1, 3, 7):
            lv3, it4, moved = g3.sgd_bipartite(lv, nu, ni, 0.001, 1e-4, 3, blocks=blocks)
            rows_mine = np.concatenate([np.nonzero(mine_users[1:nu + 1])[0], np.arange(nu, nu + ni)])
            this = it4 == 3 and bool(np.array_equal(lv3[rows_mine], olv2[rows_mine]))
            want_moved = (ni * (K + 1) * 4 if rank > 0 else 0) + (ni * (K + 1) * 4 if rank + 1 < world else 0)
            this &= moved == want_moved
            if not this:
                print("rank %d: bipartite SGD (blocks=%d) differs: %d of %d rows, moved %d (expected %d)" % (
                    rank, blocks, int((lv3[rows_mine] != olv2[rows_mine]).any(axis=1).sum()), rows_mine.size, moved, want_moved), flush=True)
            ok_bip &= this
        if rank == 0:
            print("bipartite SGD: %d bytes received per rank and iteration instead of %d" % (2 * ni * (K + 1) * 4, (nu + ni) * K * 4), flush=True)
        ok &= ok_bip
        g3.close()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("MULTI_OK" if int(flag) == 1 else "MULTI_FAIL", "world=%d exchange=%s exchanges=%d rows/shard=%d" % (
            world, "native-rccl" if native else "callback", ex.calls, g.rows), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
