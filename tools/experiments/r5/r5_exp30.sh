#!/bin/bash
# round 5, experiment 30: which kernels the unchanged SSSP / BFS spend their time in (RMAT-22, default = exact ordered folds)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e30; mkdir -p $out
python tools/app_at_scale.py 22 2>&1 | grep "==" | cut -c1-200
for app in SSSP BFS; do
  rocprofv3 --kernel-trace --stats -d $out/$app -o t -- build/ref_apps/$app /tmp/rmat22.bin.mtx 1 > $out/$app.log 2>&1
  f=$(find $out/$app -name "*kernel_stats.csv" | head -1)
  echo "== $app"; head -12 $f | cut -c1-200
done
