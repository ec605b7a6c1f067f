#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e22; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; tail -n 3 $out/gputests.txt
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
for i in 1 2; do $B > $out/prod_$i.json 2> $out/prod_$i.err; echo "product $i: $(grep summary $out/prod_$i.err | cut -c1-150)"; done
$B --debug-flags 16384 > $out/plain16.json 2> $out/plain16.err; echo "general wave16 form: $(grep summary $out/plain16.err | cut -c1-150)"
$B --scale 22 > $out/s22.json 2> $out/s22.err; echo "scale 22: $(grep summary $out/s22.err | cut -c1-150)"
$B --scale 24 > $out/s24.json 2> $out/s24.err; echo "scale 24: $(grep summary $out/s24.err | cut -c1-150)"
$B --scale 25 > $out/s25.json 2> $out/s25.err; echo "scale 25: $(grep summary $out/s25.err | cut -c1-150)"
