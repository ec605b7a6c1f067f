#!/bin/bash
# sweep with permuted slots: medium rows (sweep_slices=1) or every non-giant wave row (2); parity at small scales, timing at RMAT-26
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e26; mkdir -p $out
python - <<'PY'
import numpy as np, sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from graphmat_amd import api, _lib
L = _lib.lib()
for scale, tiles in ((16, 3), (20, 6), (22, 8)):
    res = []
    for sweep, t in ((0, 1), (1, tiles), (2, tiles)):
        L.gm_reset_options()
        L.gm_set_option(b"sweep_slices", sweep)
        nv, s, d, _ = api.rmat_on_device(scale, 16, 1)
        g = api.Graph(nv, s, d, None, keep_values=False, col_tiles=t)
        pr, deg, it = g.pagerank(7)
        sw = _lib.Sweep()
        L.gm_graph_sweep(g.h, C.byref(sw))
        res.append((pr.view(np.uint32).copy(), sw.nrows, sw.nslices, sw.npieces))
        g.close()
    print("scale %d, %d tiles: sweep 1 (rows %d, pieces %d): %d differing vertices; sweep 2 (rows %d, pieces %d): %d differing vertices" %
          (scale, tiles, res[1][1], res[1][3], int((res[0][0] != res[1][0]).sum()), res[2][1], res[2][3], int((res[0][0] != res[2][0]).sum())))
PY
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary|Error|error|differ' $out/$name.err | cut -c1-150 | head -3)"; }
for sl in 1 2; do for f in 0 1 2; do run sl${sl}_form$f --lib-option sweep_slices=$sl --lib-option sweep_form=$f; done; done
run sl2_form5 --lib-option sweep_slices=2 --lib-option sweep_form=5
