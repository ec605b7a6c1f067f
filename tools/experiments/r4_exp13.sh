#!/bin/bash
# round 4, experiment 13: the ordered giant-row fold out of the lanes' registers (option 2) against the LDS form (1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e13; mkdir -p $out
python tools/app_at_scale.py 22 2>&1 | grep "==" | head -1 | cut -c1-250
for v in 1 2 0; do echo "ordered_giant_two_pass=$v"; GRAPHMAT_OPTIONS="ordered_giant_two_pass=$v" build/ref_apps/PageRank /tmp/rmat22.bin.mtx 2>&1 | grep -E "PR Time|Completed 54"; done
GRAPHMAT_OPTIONS="ordered_giant_two_pass=2" timeout 900 python -m pytest tests/test_dropin_apps.py -x -q -m gpu -k "untraited or reference" 2>&1 | tail -2
GRAPHMAT_OPTIONS="bogus=1,wave16_form=99" build/ref_apps/PageRank /tmp/rmat22.bin.mtx 2>&1 | grep -E "ignoring|PR Time"
