#!/bin/bash
# round 5, experiment 34: the short rows behind the sweep through the persistent row-block kernel (LDS hot set of the busiest x entries) instead of the plain one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for o in "" "untiled_pass_plain=0" "untiled_pass_plain=0,persist_per_cu=1" "" "untiled_pass_plain=0"; do
  echo "== GRAPHMAT_OPTIONS=$o"
  GRAPHMAT_OPTIONS=$o timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra 2>&1 | grep -E "summary" | cut -c1-150
done
