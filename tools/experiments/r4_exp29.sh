#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e29; mkdir -p $out
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import numpy as np, sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from graphmat_amd import api, _lib
L = _lib.lib()
def run(scale, tiles, sweep, iters, opts=()):
    L.gm_reset_options()
    L.gm_set_option(b"sweep_slices", sweep)
    for k, v in opts: L.gm_set_option(k, v)
    nv, s, d, _ = api.rmat_on_device(scale, 16, 1)
    g = api.Graph(nv, s, d, None, ref_threads=1, keep_values=False, col_tiles=tiles)
    pr, deg, it = g.pagerank(iters)
    g.close()
    return pr.view(np.uint32).copy()
for scale in (20, 22):
    ref = run(scale, 1, 0, 10)
    for tiles in (2, 3, 4, 5, 8):
        for form in (1, 5, 9):
            pr = run(scale, tiles, 1, 10, ((b"sweep_form", form),))
            print("scale", scale, "tiles", tiles, "sweep_form", form, "differing", int((pr != ref).sum()))
PY
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary|Error|error|differ' $out/$name.err | cut -c1-150 | head -3)"; }
for f in 0 1 4 5 8 9; do run form$f --lib-option sweep_form=$f; done
run form1_gs0 --lib-option sweep_form=1 --lib-option giant_stream=0
run form9_gs0 --lib-option sweep_form=9 --lib-option giant_stream=0
