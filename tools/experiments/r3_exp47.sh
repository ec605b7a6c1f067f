#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for t in 4 5 6 7 8; do
echo "scale=25 persistent forced tiles=$t $(python bench.py --scale 25 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option wave16_form=18 --lib-option rowwave_form=20 --col-tiles $t 2>&1 | grep summary | cut -c40-130)"
done
