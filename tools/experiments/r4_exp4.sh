#!/bin/bash
# round 4, experiment 4: the refactored engine (options on the graph handle, Run class) through the whole GPU suite; unchanged
# apps with the pipelined ordered giant-row fold; SGD matrix-core form compared properly
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e4; mkdir -p $out
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
python tools/prof_summary.py $out/prapp_results.db | head -12 | cut -c1-200 | tee $out/r04_unchanged_pagerank_kernel_stats.md
rm -f $out/*.db
python tools/sgd_bench.py --users 2000000 --items 200000 --compare-mfma 2>&1 | grep "^SGD" | tee $out/sgd.txt
timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/bench26.json 2> $out/bench26.err; grep summary $out/bench26.err | cut -c1-200
timeout 600 python bench.py --scale 22 --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/bench22.json 2> $out/bench22.err; grep summary $out/bench22.err | cut -c1-200
