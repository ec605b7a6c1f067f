#!/bin/bash
# keep mode with the plain untiled pass: threshold / tile count again
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for o in 2048 3072 4096; do
  for t in 8 10; do
    echo "plain=1 own_wave_row=$o tiles=$t $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option untiled_pass_plain=1 --lib-option own_wave_row=$o --col-tiles $t 2>&1 | grep summary | cut -c40-150)"
  done
done
