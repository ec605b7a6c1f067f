#!/bin/bash
# round 6, closing run: whole GPU suite, smoke, the default bench line, the round's profile set for the final sources
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r6final; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
timeout 2700 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -4 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $out/bench_default.json 2> $out/bench_default.err; grep -E "summary|cpu_baseline:|extra" $out/bench_default.err | cut -c1-220
bash tools/final_profiles_r6.sh > $out/final_profiles.txt 2>&1; tail -30 $out/final_profiles.txt | cut -c1-200
