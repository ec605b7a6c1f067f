#!/bin/bash
# row classes fixed per row: threshold and tile count
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in 3072 5120 6144; do
  echo "own_wave_row=$v $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option own_wave_row=$v 2>&1 | grep summary | cut -c1-170)"
done
for t in 5 7 8 10; do
  echo "own_wave_row=4096 tiles=$t $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option own_wave_row=4096 --col-tiles $t 2>&1 | grep summary | cut -c1-170)"
done
for sc in 25 27; do
  for v in 0 4096; do
    echo "scale=$sc own_wave_row=$v $(python bench.py --scale $sc --steps 10 --warmup 3 --no-extra --cpu-scale 0 --lib-option own_wave_row=$v 2>&1 | grep summary | cut -c1-170)"
  done
done
