#!/bin/bash
# round 5, experiment 29: giant rows of undeclared float sums: speculated (exact replay) and proven chunk by chunk with the program's own function
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e29; mkdir -p $out
timeout 300 build/apps/speculated_float_sum 2>&1 | tail -12
python tools/app_at_scale.py 22 2>&1 | grep "==" | tee $out/apps22.txt | cut -c1-230
python tools/app_at_scale.py 26 2>&1 | grep "==" | tee $out/apps26.txt | cut -c1-230
GRAPHMAT_OPTIONS=ordered_giant_two_pass=1 python tools/app_at_scale.py 22 2>&1 | grep "== unchanged PageRank" | cut -c1-230
timeout 1500 python -m pytest tests -m gpu -x -q -k "dropin or unchanged or apps" 2>&1 | tail -5
