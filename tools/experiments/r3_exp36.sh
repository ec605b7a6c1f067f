#!/bin/bash
# keep mode: one-wave-per-row rows ahead of / behind the giant passes on the auxiliary stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for f in 0 1 0 1; do
  echo "aux_long_first=$f $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option aux_long_first=$f 2>&1 | grep summary | cut -c40-150)"
done
echo "aux_long_first=1 tiles=10 $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option aux_long_first=1 --col-tiles 10 2>&1 | grep summary | cut -c40-150)"
echo "aux_long_first=1 scale 25 $(python bench.py --scale 25 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option aux_long_first=1 2>&1 | grep summary | cut -c40-150)"
echo "aux_long_first=0 scale 25 $(python bench.py --scale 25 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option aux_long_first=0 2>&1 | grep summary | cut -c40-150)"
