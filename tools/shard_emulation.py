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
    args = ap.parse_args()
    from graphmat_amd import api
    nv, src, dst, _ = api.rmat_on_device(args.scale, 16, 1)
    for shard in args.shards:
        g = api.Graph(nv, src, dst, None, keep_values=False, nshards=args.nshards, shard=shard)
        st = g.new_pr_state()
        g.run_degree(st)
        g.run_pagerank(st, 2)
        g.enable_timing(True)
        torch.cuda.synchronize()
        g.run_pagerank(st, args.iters)
        s = g.last_stats()
        c = g.csr(api.GM_DIR_OUT)
        k = args.iters
        print("shard %d of %d (RMAT-%d): %d edges, %d giant rows; per iteration: total %.3f ms = send %.3f + rowblock %.3f + wave %.3f + apply %.3f; "
              "giant passes %.3f ms (overlapped on the auxiliary stream)" % (shard, args.nshards, args.scale, c.nnz, c.ngiant, s["total_ms"] / k,
              s["send_ms"] / k, s["rowblock_ms"] / k, s["wave_ms"] / k, s["apply_ms"] / k, s["giant_ms"] / k), flush=True)
        g.close()
        del st
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
