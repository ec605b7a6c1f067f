#!/bin/bash
# row classes fixed per row (default now): tile counts per scale
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "scale=$1 tiles=$2 $(python bench.py --scale $1 --steps $3 --warmup 3 --no-extra --cpu-scale 0 --col-tiles $2 2>&1 | grep summary | cut -c40-110)"; }
for t in 1 3 4; do run 24 $t 20; done
for t in 4 5 6 7; do run 25 $t 20; done
for t in 8 9 12; do run 26 $t 20; done
for t in 10 12 14 16; do run 27 $t 10; done
