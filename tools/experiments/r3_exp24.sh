#!/bin/bash
# where the long wave rows end (rows of more edges get a wave each on the auxiliary stream) under today's kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e24; mkdir -p $out
for lm in 384 512 768 1024 1536; do
  echo "long_mid=$lm $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option long_mid=$lm 2>&1 | grep summary | cut -c1-150)"
done
