#!/bin/bash
# round 4, experiment 9: head share of the two-stage schedule (with the rows it leaves to the tail), power-of-two slice lookup
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e9; mkdir -p $out
run() { echo "== $*"; GRAPHMAT_VERBOSE=1 timeout 300 python tools/shard_emulation.py --staged --shards 0 "$@" 2>&1 | grep -E "wall clock|two-stage schedule, head|shard 0 of" | sort -u | cut -c1-300; }
{
run
run --lib-option two_stage_head_permille=900
run --lib-option two_stage_head_permille=930
run --lib-option two_stage_head_permille=950
run --lib-option two_stage_head_permille=970
} > $out/shard_head_share.txt 2>&1
cat $out/shard_head_share.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "multi or dropin" > $out/pytest_subset.txt 2>&1; tail -3 $out/pytest_subset.txt
python tools/app_at_scale.py 22 2>&1 | grep "==" | head -1
