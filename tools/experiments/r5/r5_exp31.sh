#!/bin/bash
# round 5, experiment 31: giant rows of ANY undeclared function: associativity speculated, proven chunk by chunk
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e31; mkdir -p $out
timeout 300 build/apps/speculated_float_sum 2>&1 | grep -v Completed | tail -16
timeout 600 python tools/app_at_scale.py 22 2>&1 | grep "==" | tee $out/apps22.txt | cut -c1-230
timeout 900 python tools/app_at_scale.py 26 2>&1 | grep "==" | tee $out/apps26.txt | cut -c1-230
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
