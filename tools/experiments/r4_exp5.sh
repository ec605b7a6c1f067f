#!/bin/bash
# round 4, experiment 5: GPU suite, unchanged apps, the round's profile set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e5; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -6 $out/pytest_gpu.txt
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
bash tools/final_profiles_r4.sh 2>&1 | tail -40
timeout 900 python tools/shard_emulation.py --staged --shards 0 1 > $out/r04_shard_emulation.txt 2>&1; cat $out/r04_shard_emulation.txt | cut -c1-330
