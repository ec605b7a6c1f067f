#!/bin/bash
# round 5, experiment 32: kernels of the unchanged BFS / SSSP at RMAT-26 after the giant rows' speculation; ordered tests with the guess for declared-ordered programs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e32; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 600 python tools/app_at_scale.py 26 2>&1 | grep "==" | cut -c1-200
for app in BFS SSSP; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/$app -o t -- build/ref_apps/$app /tmp/rmat26.bin.mtx 1 > $out/$app.log 2>&1
  echo "== $app"; timeout 120 python tools/prof_summary.py $out/$app/t_results.db 2>&1 | grep -v "gm::k_\|rocprim\|rocclr" | head -14 | cut -c1-200
  rm -f $out/$app/t_results.db
done
