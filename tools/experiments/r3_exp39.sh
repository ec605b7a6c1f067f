#!/bin/bash
# keep mode: the untiled short-row pass through the plain row-block kernel (leaves LDS to the auxiliary stream's giant kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1 0 1; do
  echo "untiled_pass_plain=$v $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option untiled_pass_plain=$v 2>&1 | grep summary | cut -c40-150)"
done
