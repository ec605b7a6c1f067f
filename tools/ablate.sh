#!/bin/bash
# ablation sweep of the multiply+reduce kernels (debug flags make results INVALID; timing only)
# usage: tools/ablate.sh "22 26" "0 1 2 3"
for sc in $1; do
  for f in $2; do
    python bench.py --scale $sc --steps 10 --warmup 2 --cpu-scale 0 --debug-flags $f 2>&1 >/dev/null | grep summary
  done
done
