#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
python tools/app_at_scale.py 26 > /dev/null 2>&1
for gp in 1 0; do
  echo "== BFS.cpp guided_pull=$gp"
  GRAPHMAT_OPTIONS=guided_pull=$gp GRAPHMAT_VERBOSE=1 build/ref_apps/BFS /tmp/rmat26.bin.mtx 1 2>&1 | grep "GraphMat(HIP)" | grep -v "reduce_function probed\|strategy" | cut -c1-160
done > gpurun_out/r6/bfs_guided_verbose.txt
cat gpurun_out/r6/bfs_guided_verbose.txt
