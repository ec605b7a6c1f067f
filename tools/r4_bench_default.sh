#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4bench; mkdir -p $out
S=$(date +%s.%N); python bench.py > $out/bench_default.json 2> $out/bench_default.err; E=$(date +%s.%N)
echo "default bench.py wall: $(echo "$E - $S" | bc) s"
grep -E "summary|cpu_baseline:|extra|N=" $out/bench_default.err | cut -c1-220
