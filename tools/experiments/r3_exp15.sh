#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e15; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_multi.py -q -m gpu -x -k "lds_resident or persistent_kernels or other_programs" > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
python tools/bfs_bench.py --scale 26 2>&1 | grep "^BFS" | cut -c1-230
sed -i 's/static int v = 1;  \/\/ HOTBITS/static int v = 1;/' include/graphmat/engine.hpp
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from graphmat_amd import api, _lib
import torch, numpy as np
L = _lib.lib()
nv, src, dst, _ = api.rmat_on_device(26, 16, 1)
g = api.Graph(nv, src, dst, None, keep_values=False)
for form in (0, 1):
    L.gm_set_option(b"hot_bits_form", form)
    for source in (1, 12345, 777):
        g.bfs(source)
        ws = []
        for rep in range(3):
            g.bfs(source); ws.append(g.last_wall_ms)
        print("hot_bits_form=%d source=%d wall ms %s" % (form, source, ["%.2f" % w for w in ws]), flush=True)
PY
