#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e30; mkdir -p $out
timeout 3000 python -m pytest tests -q -m gpu > $out/gputests.txt 2>&1; tail -n 25 $out/gputests.txt
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
for sc in 26 25 27; do $B --scale $sc > $out/s$sc.json 2> $out/s$sc.err; echo "scale $sc: $(grep -E 'summary' $out/s$sc.err | cut -c1-150)"; done
for sc in 25 27; do $B --scale $sc --lib-option sweep_form=1 > $out/s${sc}_f1.json 2> $out/s${sc}_f1.err; echo "scale $sc form 1: $(grep -E 'summary' $out/s${sc}_f1.err | cut -c1-150)"; done
