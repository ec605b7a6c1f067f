#!/usr/bin/env python3
"""The row-stationary sweep on SHARDED graphs (graphmat_hip.h: gm_sweep_t.nsub) checked against the oracle.

Launched like tools/multi_check.py (one process per shard; on the 1-GPU test box all ranks use cuda:0 and the exchange runs
over gloo, or -- GM_EXCHANGE=native -- over the library's own RCCL path on the test suite's shared-memory stand-in).  Per
rank: the shard's device order is [slice][degree rank] inside every owner's range, slices are ascending native ranges; the sweep
structure exists with nsub = world size; PageRank through it -- with and without edge values, several launches (sets), long
rows staged in one or several rounds, the giant rows gathering for themselves or through the sweep -- has the oracle's bits;
so do the distributed build (edges_local) and, as a cross-check that the other programs still work on the sliced order, BFS
and SSSP.  Prints SWEEP_MULTI_OK on rank 0."""
import ctypes as C
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
    device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from graphmat_amd import _lib, api, generators
    from graphmat_amd.dist import attach_exchange, attach_native_exchange, exchange_counters, init_native_rccl
    from oracle import binding as ob
    L = _lib.lib()
    native = os.environ.get("GM_EXCHANGE", "callback") == "native"
    if native:
        init_native_rccl(device=torch.device("cuda", device))

    def attach(g):
        if native:
            attach_native_exchange(g)
        else:
            attach_exchange(g)

    def parts_of(g):
        return exchange_counters(g)[1] if native else g._cb[1].parts

    scale = int(os.environ.get("GM_SCALE", "15"))
    threads = 2
    nv, s, d, v = generators.rmat_edges(scale, 16, seed=9, weights="hash")
    og = ob.OracleGraph(nv, s, d, v, ref_threads=threads)
    odeg = og.degree()
    opr, oit, _ = og.pagerank(6)
    ok = True

    def fail(msg):
        nonlocal ok
        ok = False
        print("rank %d: %s" % (rank, msg), flush=True)

    nat = api.native_index(nv, threads * 16)
    sn, dn = nat[s - 1], nat[d - 1]
    deg = np.bincount(sn, minlength=nv) + np.bincount(dn, minlength=nv)
    #        sweep_form, keep_values, sweep_long_row, acc_rows, long_slots, edges_local
    cases = [(0, False, 0, 10048, 512, False), (0, True, 256, 10048, 512, False), (4, False, 128, 2, 1, False), (8, True, 512, 10048, 512, False),
             (0, False, 65, 10048, 512, False)]
    if native:
        cases += [(0, False, 256, 10048, 512, True), (4, True, 0, 2, 512, True)]
    seen_sets = 1
    for form, keep, own, accl, longl, local in cases:
        _lib.check(L.gm_reset_options())
        for k_, v_ in ((b"sweep_form", form | 256), (b"sweep_long_row", own), (b"sweep_acc_rows", accl), (b"sweep_long_slots", longl)):
            _lib.check(L.gm_set_option(k_, v_))
        if local:
            cuts = [len(s) * r // world + (13 if 0 < r < world else 0) for r in range(world + 1)]
            mine = slice(cuts[rank], cuts[rank + 1])
            g = api.Graph(nv, s[mine], d[mine], v[mine] if keep else None, ref_threads=threads, device=device, keep_values=keep, nshards=world, shard=rank,
                          col_tiles=3, edges_local=True)
        else:
            g = api.Graph(nv, s, d, v if keep else None, ref_threads=threads, device=device, keep_values=keep, nshards=world, shard=rank, col_tiles=3)
        tag = "form %d values %d long_row %d acc %d long_slots %d local %d" % (form, keep, own, accl, longl, local)
        sw = _lib.Sweep()
        _lib.check(L.gm_graph_sweep(g.h, C.byref(sw)))
        S = g.row_hi - g.row_lo
        if not (sw.nrows > 0 and sw.nsub == world and sw.stride == S and sw.hot_words > 0 and sw.nslices >= 2 and sw.val_bytes == (4 if keep else 0)):
            fail("no sharded sweep structure (%s): nrows %d nsub %d stride %d (S %d) slices %d" % (tag, sw.nrows, sw.nsub, sw.stride, S, sw.nslices))
            g.close()
            continue
        seen_sets = max(seen_sets, sw.nsets)
        # the device order: every owner's range is [slice][degree rank], a slice the same positions in every range, slices ascending
        # native ranges; vertices without edges behind the live part
        don, nod = g.maps_to_host()
        cuts_ = np.zeros(sw.nslices + 1, np.int32)
        api.copy_from_device(cuts_, sw.slice_base)
        if not (cuts_[0] == 0 and (np.diff(cuts_) >= 0).all() and cuts_[-1] <= g.xchg_rows <= S and g.ndevice == world * S):
            fail("slice positions are not ascending inside the live rows (%s)" % tag)
        live = deg > 0
        if not ((nod[don] == np.arange(nv)).all() and int((nod >= 0).sum()) == nv):
            fail("the maps are no permutation (%s)" % tag)
        prev_max = -1
        for t in range(sw.nslices):
            members = []
            for q in range(world):
                ids = nod[q * S + cuts_[t]: q * S + cuts_[t + 1]]
                ids = ids[ids >= 0]
                if ids.size and not (np.diff(deg[ids]) <= 0).all():
                    fail("slice %d of owner %d is not degree-ranked (%s)" % (t, q, tag))
                members.append(ids)
            allm = np.concatenate(members) if members else np.zeros(0, np.int64)
            if allm.size:
                if not (live[allm].all() and allm.min() > prev_max):
                    fail("slice %d is no native range behind slice %d's (%s)" % (t, t - 1, tag))
                # a native RANGE: every live vertex between its ends belongs to it
                lo, hi = int(allm.min()), int(allm.max())
                if int(live[lo:hi + 1].sum()) != allm.size:
                    fail("slice %d does not hold every live vertex of its native range (%s)" % (t, tag))
                prev_max = hi
        for q in range(world):
            tail = nod[q * S + cuts_[-1]: (q + 1) * S]
            if (tail >= 0).any() and live[tail[tail >= 0]].any():
                fail("a vertex with edges sits behind owner %d's live part (%s)" % (q, tag))
        attach(g)
        c = g.csr(api.GM_DIR_OUT)
        ng = torch.tensor([c.ngiant], dtype=torch.int64)
        dist.all_reduce(ng, op=dist.ReduceOp.MAX)
        p0 = parts_of(g)
        # fixed count, ALL_VERTICES: the sharded swept schedule (engine.hpp: run_swept_sharded) -- the all-gather of every iteration but the
        # last starts before the giant rows have been folded (one GM_XCHG_PART each), their messages follow as lists -- when some shard has a
        # giant row; the plain swept loop otherwise
        pr, dg, it = g.pagerank(6)
        want_parts = 5 if int(ng) > 0 else 0
        if parts_of(g) - p0 != want_parts:
            fail("the sharded swept schedule started %d parts, expected %d (%s; most giant rows per shard: %d)" % (parts_of(g) - p0, want_parts, tag, int(ng)))
        st = g.last_stats()
        if not ((dg == odeg).all() and it == oit == 6 and (pr.view(np.uint32) == opr.view(np.uint32)).all()):
            fail("PageRank through the sharded sweep differs from the oracle (%s): %d of %d values" % (tag, int((pr.view(np.uint32) != opr.view(np.uint32)).sum()), nv))
        if st["spmv_launches"] < 6 * sw.nsets:
            fail("fewer multiply launches than the sweep needs (%s)" % tag)
        # the shard's short rows ride its sweep too (gm_sweep_t.nstream): the path must have been taken, and refusing it (sweep_form bit 7:
        # the row-block kernel) must give the same bits
        # (the groups sit behind a block's MEDIUM groups: a structure without medium rows -- sweep_long_row 65 on a small shard -- has none;
        #  small structures only take the path when sweep_form bit 8 asks for it: the cases set it)
        n4 = C.c_int64(-1)
        if sw.nstream <= 0:
            if sw.nedges > 0:
                fail("no stream groups although the shard has medium rows (%s)" % tag)
        elif not sw.wrow_stream or L.gm_graph_note_get(g.h, 4, C.byref(n4)) != 0 or n4.value != 6:
            fail("the shard's short rows did not go through the sweep (%s): nstream %d, note 4 = %d" % (tag, sw.nstream, n4.value))
        _lib.check(L.gm_set_option(b"sweep_form", form | 128))
        pr_r, _, _ = g.pagerank(6)
        _lib.check(L.gm_set_option(b"sweep_form", form | 256))
        if L.gm_graph_note_get(g.h, 4, C.byref(n4)) != 0 or n4.value != 0 or not (pr_r.view(np.uint32) == opr.view(np.uint32)).all():
            fail("the row-block kernel for the shard's short rows differs or was not taken (%s)" % tag)
        # ... and the plain loop (all-gather between send and multiply) through the same sweep
        p1 = parts_of(g)
        _lib.check(L.gm_set_option(b"debug_flags", 128))
        pr_p, _, it_p = g.pagerank(6)
        _lib.check(L.gm_set_option(b"debug_flags", 0))
        if parts_of(g) != p1 or it_p != 6 or not (pr_p.view(np.uint32) == opr.view(np.uint32)).all():
            fail("the plain swept loop differs from the oracle or started parts (%s)" % tag)
        g.close()
    if seen_sets < 2 and world <= 2:  # (a shard of three of a small graph has too few rows for a second launch)
        fail("several launches (sets) were never exercised")
    # the same graph without the sweep (sweep_slices 0: the round-5 sharded path, two-stage schedule) has the same bits, and the other
    # programs still run on the sliced order
    _lib.check(L.gm_reset_options())
    g = api.Graph(nv, s, d, v, ref_threads=threads, device=device, nshards=world, shard=rank, col_tiles=3)
    attach(g)
    pr2, _, it2 = g.pagerank(-1)
    opr2, oit2, _ = og.pagerank(-1)
    if not (it2 == oit2 and (pr2.view(np.uint32) == opr2.view(np.uint32)).all()):
        fail("PageRank until convergence through the sharded sweep differs (%d vs %d iterations)" % (it2, oit2))
    depth, parent, itb = g.bfs(3)
    od, op, oitb, _ = og.bfs(3)
    if not (itb == oitb and (depth == od).all() and (parent == op).all()):
        fail("BFS on the sliced sharded order differs from the oracle")
    dist_, its = g.sssp(1)
    odist, oits = og.sssp(1)
    if not (its == oits and (dist_ == odist).all()):
        fail("SSSP on the sliced sharded order differs from the oracle")
    g.close()
    _lib.check(L.gm_set_option(b"sweep_slices", 0))
    g = api.Graph(nv, s, d, None, ref_threads=threads, device=device, keep_values=False, nshards=world, shard=rank, col_tiles=3)
    sw = _lib.Sweep()
    _lib.check(L.gm_graph_sweep(g.h, C.byref(sw)))
    attach(g)
    pr3, _, _ = g.pagerank(6)
    if sw.nrows != 0 or sw.nstream != 0 or not (pr3.view(np.uint32) == opr.view(np.uint32)).all():
        fail("the unswept sharded path differs (sweep rows %d)" % sw.nrows)
    g.close()
    _lib.check(L.gm_reset_options())
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("SWEEP_MULTI_OK" if int(flag) == 1 else "SWEEP_MULTI_FAIL", "world=%d scale=%d exchange=%s" % (world, scale, "native" if native else "callback"), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
