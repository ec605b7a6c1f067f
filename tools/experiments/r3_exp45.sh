#!/bin/bash
# where the untiled rows end: short_row = tile_min_row = 64 (default) / 48 / 32 / 24
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in 64 48 32 24; do
  echo "short_row=tile_min_row=$v $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option short_row=$v --tile-min-row $v 2>&1 | grep summary | cut -c40-200)"
done
for v in 96 128; do
  echo "tile_min_row=$v $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --tile-min-row $v 2>&1 | grep summary | cut -c40-200)"
done
