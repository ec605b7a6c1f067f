#!/bin/bash
# RMAT-25 under the new schedule: persistent kernels forced / not
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do
echo "scale=25 default $(python bench.py --scale 25 --steps 20 --warmup 5 --no-extra --cpu-scale 0 2>&1 | grep summary | cut -c40-130)"
echo "scale=25 persistent forced $(python bench.py --scale 25 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option wave16_form=18 --lib-option rowwave_form=20 2>&1 | grep summary | cut -c40-130)"
done
echo "scale=25 persistent forced tiles=5 $(python bench.py --scale 25 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option wave16_form=18 --lib-option rowwave_form=20 --col-tiles 5 2>&1 | grep summary | cut -c40-130)"
echo "scale=24 persistent forced tiles=3 $(python bench.py --scale 24 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option wave16_form=18 --lib-option rowwave_form=20 --col-tiles 3 2>&1 | grep summary | cut -c40-130)"
