#!/usr/bin/env python3
"""Shape of the work the multiply kernels see on a tiled graph (DESIGN.md §6, round 4 "instruction diet").

For every column tile of the RMAT graph the library builds: the 16-rows-per-wave list (groups, 64-edge steps per group =
the longest of its 16 pieces, how full the 16 x 64 slots of a step are), the one-wave-per-row list (rows, 64-edge
chunks) and the row-blocks (64-row groups, 512-edge steps).  Copies row pointers and lists to the host; numpy only.

    python tools/work_shape_probe.py --scale 26
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphmat_amd import api  # noqa: E402
from graphmat_amd._lib import GM_DIR_OUT  # noqa: E402


def dev(ptr, n, dtype):
    a = np.zeros(n, dtype)
    if n:
        api.copy_from_device(a, ptr)
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    args = ap.parse_args()
    nv, s, d, _ = api.rmat_on_device(args.scale, 16, 1)
    g = api.Graph(nv, s, d, None, keep_values=False)
    del s, d
    nt = g.tiles(GM_DIR_OUT) if hasattr(g, "tiles") else None
    if nt is None:
        import ctypes as C
        n = C.c_int()
        g.L.gm_graph_tiles(g.h, GM_DIR_OUT, C.byref(n))
        nt = n.value
    print("scale %d: %d column tiles" % (args.scale, nt))
    tot = dict(w16_groups=0, w16_steps=0, w16_edges=0, wave_rows=0, wave_chunks=0, wave_edges=0, blk=0, blk_groups=0, blk_steps=0, blk_edges=0)
    views = [("whole graph (untiled pass)", g.csr(GM_DIR_OUT))] + [("tile %d" % t, g.tile(GM_DIR_OUT, t)[0]) for t in range(nt if nt > 1 else 0)]
    for name, c in views:
        rp = dev(c.rowptr, c.nrows + 1, np.int64)
        ln = np.diff(rp)
        whole = name.startswith("whole")
        if not whole or nt <= 1:
            mid = dev(c.mid_row, c.nmid, np.int32)
            nlong = min(c.nmid_long, c.nmid)
            lon = ln[mid[:nlong]]
            rest = ln[mid[nlong:]]
            ng = (len(rest) + 15) // 16
            pad = np.zeros(ng * 16, np.int64)
            pad[: len(rest)] = rest
            grp = pad.reshape(ng, 16)
            steps = (grp.max(axis=1) + 63) // 64
            e16 = int(rest.sum())
            chunks = int(((lon + 63) // 64).sum())
            print("%-28s 16-row groups %8d  steps %9d (%.2f per group)  edges %11d  slot fill %.3f  pieces: mean %.0f max %d | wave rows %7d chunks %9d edges %11d mean %.0f"
                  % (name, ng, int(steps.sum()), steps.mean() if ng else 0, e16, e16 / max(1, int(steps.sum()) * 1024), rest.mean() if len(rest) else 0,
                     int(rest.max()) if len(rest) else 0, nlong, chunks, int(lon.sum()), lon.mean() if nlong else 0))
            tot["w16_groups"] += ng; tot["w16_steps"] += int(steps.sum()); tot["w16_edges"] += e16
            tot["wave_rows"] += nlong; tot["wave_chunks"] += chunks; tot["wave_edges"] += int(lon.sum())
        if c.nblk > 0:
            seg = dev(c.seg_row, c.nseg + 1, np.int32)
            bs = dev(c.blk_seg, c.nblk, np.int32)
            r0, r1 = seg[bs], seg[bs + 1]
            nrow = (r1 - r0).astype(np.int64)
            e = rp[r1] - rp[r0]
            groups = int(((nrow + 63) // 64).sum())
            # steps of 512 edges per 64-row group ~ edges of the group / 512, at least one
            steps = int(np.maximum(1, (e + 511) // 512).sum())
            print("%-28s row-blocks %8d  rows %10d  64-row groups %9d  edges %11d (%.1f per row, %.0f per block)" %
                  (name, c.nblk, int(nrow.sum()), groups, int(e.sum()), e.sum() / max(1, nrow.sum()), e.mean()))
            tot["blk"] += c.nblk; tot["blk_groups"] += groups; tot["blk_steps"] += steps; tot["blk_edges"] += int(e.sum())
    print("totals per iteration:", tot)
    if tot["w16_steps"]:
        print("16-rows-per-wave kernel: %.1f edges per 1024-slot step (fill %.3f)" % (tot["w16_edges"] / tot["w16_steps"], tot["w16_edges"] / tot["w16_steps"] / 1024))


if __name__ == "__main__":
    main()
