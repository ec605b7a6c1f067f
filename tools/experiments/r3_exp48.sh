#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for sc in 25 26 27; do
echo "scale=$sc default $(python bench.py --scale $sc --steps 12 --warmup 3 --no-extra --cpu-scale 0 2>&1 | grep summary | cut -c40-130)"
done
