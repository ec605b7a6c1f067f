#!/bin/bash
# round 5, experiment 28: ordered folds of the one-wave-per-row kernel out of LDS (dense chunks): parity of the ordered programs + the unchanged apps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e28; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q -k "ordered or dropin or unchanged or parity or oracle" 2>&1 | tail -5
python tools/app_at_scale.py 22 2>&1 | grep "==" | tee $out/apps22.txt | cut -c1-230
python tools/app_at_scale.py 26 2>&1 | grep "==" | tee $out/apps26.txt | cut -c1-230
