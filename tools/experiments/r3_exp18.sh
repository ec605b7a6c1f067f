#!/bin/bash
# BFS bottom-up: lanes per row (16 / 8) x (grouped kernel on the main / auxiliary stream): parity, then wall clock per traversal
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e18; mkdir -p $out
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
for lanes in (16, 8):
    for ov in (0, 1):
        L.gm_set_option(b"last_rows_lanes", lanes)
        L.gm_set_option(b"last_rows_overlap", ov)
        for source in (1, 12345, 777):
            g.bfs(source)
            ws = []
            for rep in range(3):
                g.bfs(source); ws.append(g.last_wall_ms)
            print("lanes=%d overlap=%d source=%d wall ms %s" % (lanes, ov, source, ["%.2f" % w for w in ws]), flush=True)
PY
bash tools/bfs_timeline.sh > /dev/null 2>&1
cp gpurun_out/bfstl/bfs_timeline.md $out/bfs_timeline.md
grep -n "k_spmv\|k_apply\|k_send<\|k_bits\|k_push_bid_bits\|k_want\|k_frontier" $out/bfs_timeline.md | tail -24 | cut -c1-140
