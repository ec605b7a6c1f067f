#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e28; mkdir -p $out
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
    indeg = np.bincount(d.cpu().numpy() - 1, minlength=nv)
    g = api.Graph(nv, s, d, None, ref_threads=1, keep_values=False, col_tiles=tiles)
    pr, deg, it = g.pagerank(iters)
    sw = _lib.Sweep(); L.gm_graph_sweep(g.h, C.byref(sw))
    g.close()
    return pr.view(np.uint32).copy(), indeg, (sw.nrows, sw.nslices, sw.npieces)
ref, indeg, _ = run(22, 1, 0, 10)
for tiles in (4, 6, 8):
    for opts in ((), ((b"giant_stream", 0),), ((b"sweep_form", 4),), ((b"sweep_form", 0),)):
        for rep in range(2):
            pr, _, info = run(22, tiles, 1, 10, opts)
            bad = np.nonzero(pr != ref)[0]
            print("tiles", tiles, "opts", opts, "sweep", info, "differing", len(bad), "in-degree of the first few", indeg[bad[:8]].tolist(), "max indeg", int(indeg[bad].max()) if len(bad) else 0, "min", int(indeg[bad].min()) if len(bad) else 0)
PY
