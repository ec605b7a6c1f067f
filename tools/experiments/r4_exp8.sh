#!/bin/bash
# round 4, experiment 8: scalar-load ordered giant fold (unchanged apps), head share of the two-stage schedule on a shard of 8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e8; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu -k "dropin or apps or giant or ordered or multi" > $out/pytest_subset.txt 2>&1; tail -4 $out/pytest_subset.txt
{
echo "# unchanged reference apps (build/ref_apps) on RMAT-22: exact-by-default (ordered fold: no trait, no probe) vs GRAPHMAT_TRUST_PROBE=1"
python tools/app_at_scale.py 22 2>&1 | grep "=="
echo "# GRAPHMAT_TRUST_PROBE=1"
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 22 2>&1 | grep "=="
} > $out/r04_unchanged_apps.txt
cat $out/r04_unchanged_apps.txt
rocprofv3 --kernel-trace --stats -d $out -o prapp -- build/ref_apps/PageRank /tmp/rmat22.bin.mtx > $out/prapp.out 2> $out/prapp.err
python tools/prof_summary.py $out/prapp_results.db | head -8 | cut -c1-200 | tee $out/r04_unchanged_pagerank_kernel_stats.md
rm -f $out/*.db
run() { echo "== $*"; timeout 300 python tools/shard_emulation.py --staged --shards 0 "$@" 2>&1 | grep -E "wall clock" | cut -c1-300; }
{
run
run --lib-option two_stage_head_permille=750
run --lib-option two_stage_head_permille=800
run --lib-option two_stage_head_permille=850
run --lib-option two_stage_head_permille=900
run --lib-option two_stage_head_permille=500
} > $out/shard_head_share.txt 2>&1
cat $out/shard_head_share.txt
