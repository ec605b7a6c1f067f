#!/bin/bash
# BFS bottom-up: rows per lane in the short rows' kernel (1 / 2 / 4): parity on each, then wall clock per traversal
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e19; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x -k "bfs or BFS or sssp or SSSP or last or dropin or apps or topo" > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from graphmat_amd import api, _lib
import torch, numpy as np
L = _lib.lib()
nv, src, dst, _ = api.rmat_on_device(26, 16, 1)
g = api.Graph(nv, src, dst, None, keep_values=False)
ref = {}
for rows in (1, 2, 4):
    L.gm_set_option(b"short_last_rows", rows)
    for source in (1, 12345, 777):
        depth, parent, it = g.bfs(source)
        if rows == 1: ref[source] = (depth.copy(), parent.copy())
        else: assert np.array_equal(depth, ref[source][0]) and np.array_equal(parent, ref[source][1]), "results differ"
        ws = []
        for rep in range(3):
            g.bfs(source); ws.append(g.last_wall_ms)
        print("short_last_rows=%d source=%d wall ms %s" % (rows, source, ["%.2f" % w for w in ws]), flush=True)
PY
python tools/bfs_bench.py --scale 26 2>&1 | grep "^BFS" | cut -c1-250
