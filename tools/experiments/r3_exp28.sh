#!/bin/bash
# fewer host calls per iteration of a steered run (one clear, one copy): whole GPU suite, then BFS wall clock per traversal
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e28; mkdir -p $out
timeout 1700 python -m pytest tests -q -m gpu -x > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from graphmat_amd import api, _lib
import torch, numpy as np
nv, src, dst, _ = api.rmat_on_device(26, 16, 1)
g = api.Graph(nv, src, dst, None, keep_values=False)
for source in (1, 12345, 777, 5, 4242):
    g.bfs(source)
    ws = []
    for rep in range(5):
        g.bfs(source); ws.append(g.last_wall_ms)
    print("source=%d wall ms %s" % (source, ["%.2f" % w for w in ws]), flush=True)
PY
