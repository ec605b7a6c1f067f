#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e31; mkdir -p $out
timeout 3000 python -m pytest tests -q -m gpu > $out/gputests.txt 2>&1; tail -n 12 $out/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err; grep -E "summary|extra|cpu_baseline:" $out/bench_default.err | cut -c1-200; head -c 600 $out/bench_default.json; echo
