#!/bin/bash
# round 4, closing run: whole GPU suite, smoke, the default bench line, the round's profile set for the final sources
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4final; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -4 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $out/bench_default.json 2> $out/bench_default.err; grep -E "summary|cpu_baseline:|extra" $out/bench_default.err | cut -c1-220
bash tools/final_profiles_r4.sh > $out/final_profiles.txt 2>&1; tail -3 $out/final_profiles.txt | cut -c1-160
{
echo "# unchanged reference apps (build/ref_apps) on RMAT-22: exact-by-default (ordered fold: no trait, no probe) vs GRAPHMAT_TRUST_PROBE=1"
python tools/app_at_scale.py 22 2>&1 | grep "=="
echo "# GRAPHMAT_TRUST_PROBE=1"
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 22 2>&1 | grep "=="
} > $out/r04_unchanged_apps.txt; cat $out/r04_unchanged_apps.txt | cut -c1-220
