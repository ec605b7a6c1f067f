#!/usr/bin/env python3
"""Native RCCL exchange at the benchmark's size with a single rank: the staged live-prefix all-gather (131 MB at
RMAT-26) and its 2-D scatter copy, the overlapped parts and the flag all-reduce run for real through RCCL; the result
must equal, bit for bit, the same iterations without any exchange.  Prints NATIVE_BIG_OK."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from graphmat_amd import api, dist as gdist
    nv, src, dst, _ = api.rmat_on_device(args.scale, 16, 1)
    g = api.Graph(nv, src, dst, None, keep_values=False, nshards=1, shard=0)
    del src, dst
    st0 = g.new_pr_state()
    g.run_degree(st0)
    a = st0.clone()
    g.run_pagerank(a, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.run_pagerank(a, args.iters - 1)
    torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / (args.iters - 1)
    gdist.init_native_rccl()
    gdist.attach_native_exchange(g)
    b = st0.clone()
    g.run_pagerank(b, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.run_pagerank(b, args.iters - 1)
    torch.cuda.synchronize(); t_native = (time.perf_counter() - t0) / (args.iters - 1)
    calls, parts, sent = gdist.exchange_counters(g)
    same = bool(torch.equal(a, b))
    c = st0.clone()
    it = g.run_pagerank(c, -1)  # until convergence: flag all-reduce every iteration
    print("RMAT-%d: %.3f ms/iteration without exchange, %.3f ms with the single-rank native exchange (%d calls, %d parts, %.1f MB sent); "
          "bits equal: %s; until convergence: %d iterations" % (args.scale, t_plain * 1e3, t_native * 1e3, calls, parts, sent / 1e6, same, it))
    print("NATIVE_BIG_OK" if same and parts > 0 else "NATIVE_BIG_FAIL")
    sys.exit(0 if same and parts > 0 else 1)


if __name__ == "__main__":
    main()
