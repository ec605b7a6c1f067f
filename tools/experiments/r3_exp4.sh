#!/bin/bash
# round 3: whole GPU test suite, unchanged apps with and without the probe, default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e4; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -15 $out/pytest_gpu.txt
{
echo "# unchanged reference apps (build/ref_apps) on RMAT-22: exact-by-default (ordered fold: no trait, no probe) vs GRAPHMAT_TRUST_PROBE=1"
python tools/app_at_scale.py 22 2>&1 | grep "=="
echo "# GRAPHMAT_TRUST_PROBE=1"
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 22 2>&1 | grep "=="
} > $out/r03_unchanged_apps.txt
cat $out/r03_unchanged_apps.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -12 $out/bench_default.err
