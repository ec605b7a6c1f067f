#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
bash tools/sweep.sh 26 "--no-extra" "--no-extra --lib-option sweep_waves=12" "--no-extra --lib-option sweep_waves=12"
bash tools/sweep.sh 25 "--no-extra" "--no-extra --lib-option sweep_waves=12"
bash tools/sweep.sh 24 "--no-extra" "--no-extra --lib-option sweep_waves=12"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/sweep_w12.txt
grep "==\|summary" gpurun_out/r6/sweep_w12.txt | sed 's/\[bench\] summary //' | cut -c1-150
python - <<'PY'
# bits: PageRank RMAT-20 with forced tiles through the 768-thread form == oracle
import numpy as np, ctypes as C
from graphmat_amd import api, generators as gen, _lib
from oracle import binding as ob
L=_lib.lib()
nv,s,d,v=gen.rmat_edges(16,16,seed=5,weights="hash")
og=ob.OracleGraph(nv,s,d,None,ref_threads=1); opr,_,_=og.pagerank(6)
for keep in (False,True):
    L.gm_reset_options(); L.gm_set_option(b"sweep_waves",12); L.gm_set_option(b"sweep_long_row",256)
    g=api.Graph(nv,s,d,v if keep else None,ref_threads=1,keep_values=keep,col_tiles=4)
    sw=_lib.Sweep(); L.gm_graph_sweep(g.h,C.byref(sw))
    pr,deg,it=g.pagerank(6)
    print("waves",sw.waves,"keep",keep,"bit-exact:",bool((pr.view(np.uint32)==opr.view(np.uint32)).all()))
L.gm_reset_options()
PY
