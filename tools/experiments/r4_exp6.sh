#!/bin/bash
# round 4, experiment 6: a shard of 8 under other graph-build parameters (row-class thresholds), plain loop and two-stage
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e6; mkdir -p $out
run() { echo "== $*"; timeout 300 python tools/shard_emulation.py --staged --shards 0 "$@" 2>&1 | grep -E "shard 0|wall clock" | cut -c1-300; }
{
run
run --lib-option short_row=32
run --lib-option short_row=96
run --lib-option short_row=128
run --lib-option giant_row=8192
run --lib-option giant_row=16384
run --lib-option giant_row=65536
run --lib-option long_mid=512
run --lib-option long_mid=2048
run --lib-option long_mid=4096
run --lib-option giant_maps=0
} > $out/shard_sweep.txt 2>&1
cat $out/shard_sweep.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "dropin or apps or giant or ordered or multi or parity" > $out/pytest_subset.txt 2>&1; tail -4 $out/pytest_subset.txt
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
