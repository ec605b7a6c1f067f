#!/usr/bin/env python3
"""BFS timing (BASELINE config 3): kernel-level time of gm_run_bfs on RMAT-<scale>, per source.
TEPS = edges whose source is reachable / total BFS time."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--debug-flags", type=int, default=0, help="ablation: 32 no push steps, 64 no grouped wave kernel")
    ap.add_argument("--push-permille", type=int, default=-1, help="experiment: top-down while the active set owns < this many thousandths of the edges")
    args = ap.parse_args()
    from graphmat_amd import api, _lib
    if args.push_permille >= 0:
        _lib.lib().gm_set_option(b"push_edge_permille", args.push_permille)
    if args.debug_flags:
        _lib.lib().gm_set_option(b"debug_flags", args.debug_flags)
    nv, src, dst, _ = api.rmat_on_device(args.scale, 16, 1)
    g = api.Graph(nv, src, dst, None, keep_values=False)  # both directions: the push step needs the by-source adjacency
    g.enable_timing(True)
    for source in (1, 12345, 777):
        depth, parent, it = g.bfs(source)
        st = g.last_stats()
        d = torch.from_numpy(depth.astype(np.int64)).cuda()
        e_reach = int((d[(src - 1).long()] != 0xFFFFFFFF).sum())
        print("BFS scale=%d source=%d levels=%d reached=%d: whole call %.2f ms wall (=> %.1f GTEPS); iteration kernels %.2f ms (send %.2f, multiply %.2f [rowblock %.2f wave %.2f giant %.2f], "
              "apply %.2f) => %.1f GTEPS on %d traversable edges" % (args.scale, source, it, int((d != 0xFFFFFFFF).sum()), g.last_wall_ms,
              e_reach / g.last_wall_ms / 1e6, st["total_ms"],
              st["send_ms"], st["spmv_ms"], st["rowblock_ms"], st["wave_ms"], st["giant_ms"], st["apply_ms"],
              e_reach / st["total_ms"] / 1e6, e_reach), flush=True)

if __name__ == "__main__":
    main()
