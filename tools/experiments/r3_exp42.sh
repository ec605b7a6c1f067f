#!/bin/bash
# keep mode: LDS left to the auxiliary stream's giant kernels by the persistent 16-row kernel (hot set 22528 / 16384 / 12288 entries)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for f in 2 7 6 2 6; do
  echo "wave16_form=$f $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option wave16_form=$f 2>&1 | grep summary | cut -c40-150)"
done
