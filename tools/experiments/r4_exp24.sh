#!/bin/bash
# the row-stationary sweep inside the library: correctness (bench's own checks + tiled == untiled at small scale) and timing
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e24; mkdir -p $out
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary|Error|error|differ' $out/$name.err | cut -c1-170 | head -3)"; }
run s22_t4_sweep --scale 22 --col-tiles 4 --lib-option sweep_slices=1
run s22_t4 --scale 22 --col-tiles 4
run s26_sweep --scale 26 --lib-option sweep_slices=1
run s26 --scale 26
python - <<'PY'
import numpy as np, sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from graphmat_amd import api, _lib
L = _lib.lib()
for scale, tiles in ((16, 3), (18, 4), (20, 6), (22, 8)):
    res = []
    for sweep, t in ((0, 1), (0, tiles), (1, tiles)):
        L.gm_reset_options()
        L.gm_set_option(b"sweep_slices", sweep)
        nv, s, d, _ = api.rmat_on_device(scale, 16, 1)
        g = api.Graph(nv, s, d, None, keep_values=False, col_tiles=t)
        pr, deg, it = g.pagerank(7)
        sw = _lib.Sweep()
        L.gm_graph_sweep(g.h, C.byref(sw))
        res.append((pr.view(np.uint32).copy(), sw.nrows, sw.nslices, sw.npieces, sw.nsets))
        g.close()
    print("scale %d: untiled vs %d tiles: %d differing vertices; untiled vs %d tiles + sweep (rows %d, slices %d, pieces %d, sets %d): %d differing vertices" %
          (scale, tiles, int((res[0][0] != res[1][0]).sum()), tiles, res[2][1], res[2][2], res[2][3], res[2][4], int((res[0][0] != res[2][0]).sum())))
PY
