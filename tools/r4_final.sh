#!/bin/bash
# round 4, closing run: whole GPU suite, smoke, the default bench line (timed), the round's profile set for the final sources
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4final; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -4 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
/usr/bin/time -v python bench.py > $out/bench_default.json 2> $out/bench_default.err; grep -E "Elapsed|summary|cpu_baseline:|extra" $out/bench_default.err | cut -c1-220
bash tools/final_profiles_r4.sh > $out/final_profiles.txt 2>&1; tail -5 $out/final_profiles.txt | cut -c1-200
