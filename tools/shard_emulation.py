#!/usr/bin/env python3
"""What one shard of an N-GPU PageRank run costs, measured on a single GPU: shard `s` of `N` is built as on rank s
(degree-ranked order dealt over N shards) and its iteration kernels are timed WITHOUT the message exchange (x holds
stale values: the access pattern and the work are those of the real run, the results are not).  Tells how the
per-GPU compute -- including the serial chain of the giant rows, which does not shrink with N -- limits scaling."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--nshards", type=int, default=8)
    ap.add_argument("--shards", type=int, nargs="+", default=[0, 1, 7])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--staged", action="store_true", help="also time the two-stage schedule of a sharded run (tail rows, then head "
                    "rows, each multiplied / applied / sent separately) with an exchange that moves nothing, and the plain loop, by the wall clock")
    ap.add_argument("--lib-option", action="append", default=[], metavar="KEY=VALUE", help="gm_set_option(KEY, VALUE) before the graphs are built (experiments)")
    args = ap.parse_args()
    import time
    from graphmat_amd import _lib, api
    L = _lib.lib()
    for kv in args.lib_option:
        k, v = kv.split("=")
        _lib.check(L.gm_set_option(k.encode(), int(v)))
    nv, src, dst, _ = api.rmat_on_device(args.scale, 16, 1)
    for shard in args.shards:
        g = api.Graph(nv, src, dst, None, keep_values=False, nshards=args.nshards, shard=shard)
        st = g.new_pr_state()
        g.run_degree(st)
        g.run_pagerank(st, 2)
        g.enable_timing(True)
        torch.cuda.synchronize()
        import ctypes as C
        cnt = (C.c_int64 * 4)()
        L.gm_debug_counters(cnt)  # (reset)
        g.run_pagerank(st, args.iters)
        L.gm_debug_counters(cnt)
        s = g.last_stats()
        c = g.csr(api.GM_DIR_OUT)
        k = args.iters
        print("shard %d of %d (RMAT-%d): %d edges, %d giant rows; per iteration: total %.3f ms = send %.3f + rowblock %.3f + wave %.3f + apply %.3f; "
              "giant passes %.3f ms (overlapped on the auxiliary stream)" % (shard, args.nshards, args.scale, c.nnz, c.ngiant, s["total_ms"] / k,
              s["send_ms"] / k, s["rowblock_ms"] / k, s["wave_ms"] / k, s["apply_ms"] / k, s["giant_ms"] / k), flush=True)
        print("   giant rows' exact replay per iteration: %d 16-product groups accepted by the parallel scan, %d folded serially, %d skipped by piece maps"
              % (cnt[0] // k, cnt[1] // k, cnt[2] // k), flush=True)
        if args.staged:
            # a do-nothing exchange makes the run "sharded": the engine picks the two-stage schedule (needs the second
            # message buffer) or, with debug flag 128, the plain loop with one exchange per iteration
            dev = torch.device("cuda", 0)
            bufs = [torch.zeros(g.ndevice * 4 + 64, dtype=torch.uint8, device=dev), torch.zeros((g.ndevice + 31) // 32 + 2, dtype=torch.int32, device=dev),
                    torch.zeros(g.ndevice * 4 + 64, dtype=torch.uint8, device=dev)]
            for slot, b in zip((1, 2, 9), bufs):
                _lib.check(L.gm_graph_adopt_workspace(g.h, slot, b.data_ptr(), b.numel() * b.element_size()))
            calls = [0]

            def nothing(ctx, kind, ptr, elt, bits, flag):
                calls[0] += 1
                return 0
            cb = _lib.EXCHANGE_FN(nothing)
            _lib.check(L.gm_graph_set_exchange(g.h, cb, None))
            _lib.check(L.gm_graph_set_exchange_caps(g.h, _lib.GM_XCAP_SPARSE))  # (the sharded swept schedule sends the giant rows' messages as lists)
            g.enable_timing(False)
            res = {}
            for name, flags in (("two-stage", 0), ("plain", 128), ("late", 4096)):
                L.gm_set_option(b"debug_flags", flags)
                g.run_pagerank(st, 3)
                torch.cuda.synchronize()
                calls[0] = 0
                t0 = time.perf_counter()
                g.run_pagerank(st, args.iters)
                torch.cuda.synchronize()
                res[name] = ((time.perf_counter() - t0) * 1e3 / args.iters, calls[0])
            L.gm_set_option(b"debug_flags", 0)
            import ctypes as C2
            sw = _lib.Sweep()
            swept = L.gm_graph_sweep(g.h, C2.byref(sw)) == 0 and sw.nsub > 1
            if swept:
                print("   wall clock per iteration with a do-nothing exchange: sharded swept schedule (apply + send of the other rows while the giant rows fold) "
                      "%.3f ms (%d exchange calls), plain swept loop %.3f ms (%d)" % (res["two-stage"][0], res["two-stage"][1], res["plain"][0], res["plain"][1]), flush=True)
            else:
                print("   wall clock per iteration with a do-nothing exchange: two-stage schedule %.3f ms (%d exchange calls; %.3f ms when the giant rows "
                      "start with the head stage), plain loop %.3f ms (%d)" % (res["two-stage"][0], res["two-stage"][1], res["late"][0], res["plain"][0], res["plain"][1]), flush=True)
            del bufs
        g.close()
        del st
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
