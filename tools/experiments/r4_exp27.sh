#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e27; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gputests.txt 2>&1; tail -n 15 $out/gputests.txt
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
for sc in 26 25 24 27; do $B --scale $sc > $out/s$sc.json 2> $out/s$sc.err; echo "scale $sc: $(grep -E 'summary' $out/s$sc.err | cut -c1-150)"; done
$B --scale 25 --lib-option sweep_slices=0 > $out/s25_off.json 2> $out/s25_off.err; echo "scale 25 no sweep: $(grep -E 'summary' $out/s25_off.err | cut -c1-150)"
$B --scale 27 --lib-option sweep_slices=0 > $out/s27_off.json 2> $out/s27_off.err; echo "scale 27 no sweep: $(grep -E 'summary' $out/s27_off.err | cut -c1-150)"
timeout 900 python bench.py --cpu-scale 0 > $out/bench_extra.json 2> $out/bench_extra.err; grep -E "extra|summary" $out/bench_extra.err | cut -c1-200
