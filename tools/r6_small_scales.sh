#!/bin/bash
# round 6: the other programs at the small scales under the automatic tile/sweep policy and with GRAPHMAT_COL_TILES=1 (round 5's: no tiles)
mkdir -p gpurun_out/r6
for sc in ${SCALES:-22 23}; do
  echo "== RMAT-$sc: automatic policy"
  python tools/app_at_scale.py $sc 2>&1 | grep -v amdgpu.ids | cut -c1-260
  python tools/bfs_bench.py $sc 2>&1 | grep "^BFS" | cut -c1-260
  echo "== RMAT-$sc: GRAPHMAT_COL_TILES=1"
  GRAPHMAT_COL_TILES=1 python tools/app_at_scale.py $sc 2>&1 | grep -v amdgpu.ids | cut -c1-260
  GRAPHMAT_COL_TILES=1 python tools/bfs_bench.py $sc 2>&1 | grep "^BFS" | cut -c1-260
done
