#!/usr/bin/env python3
"""Where in the device order do the frontiers of a BFS traversal sit?  (For the bottom-up levels: could the presence bits of the
frontier live in an LDS window over the first device ids?)  Per level: vertices, out-edges, and the share of vertices / of their
out-edges below a few device-id limits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from graphmat_amd import api

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
nv, src, dst, _ = api.rmat_on_device(scale, 16, 1)
g = api.Graph(nv, src, dst, None, keep_values=False)
don, nod = g.maps_to_host()
outdeg = torch.bincount((src - 1).long(), minlength=nv).cpu().numpy()
for source in (1, 777):
    depth, parent, it = g.bfs(source)
    for lvl in range(0, it):
        f = np.nonzero(depth == lvl)[0]          # 0-based vertex ids = native? (depth is indexed by vertex id - 1)
        if f.size == 0:
            continue
        from graphmat_amd.api import native_index
        nat = native_index(nv, 16)[f]            # vertex -> native (0-based)
        dev = don[nat]
        e = outdeg[f]
        line = "source %d level %d: %d vertices, %d out-edges;" % (source, lvl, f.size, int(e.sum()))
        for lim in (65536, 262144, 524288, 1048576, 4194304):
            m = dev < lim
            line += " <%dK: %.1f%% v / %.1f%% e;" % (lim >> 10, 100.0 * m.mean(), 100.0 * e[m].sum() / max(1, e.sum()))
        print(line, flush=True)
