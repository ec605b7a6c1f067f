#!/bin/bash
# untiled graphs: the row-block kernel behind the giant passes on the auxiliary stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1; do
  for sc in 22 24; do
    echo "rowblock_on_aux=$v scale=$sc $(python bench.py --scale $sc --steps 30 --warmup 5 --no-extra --cpu-scale 0 --lib-option rowblock_on_aux=$v 2>&1 | grep summary | cut -c40-150)"
  done
  echo "rowblock_on_aux=$v"; python tools/shard_emulation.py --staged --shards 0 --lib-option rowblock_on_aux=$v 2>&1 | grep -v amdgpu | cut -c1-300
done
