#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4bench; mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
grep -E "summary|cpu_baseline|extra|N=" $out/bench_default.err | cut -c1-220
