#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e35; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | tail -n 3
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s26b --scale 26
run s26_f0 --scale 26 --lib-option sweep_form=0
run s26_f1 --scale 26 --lib-option sweep_form=1
run s25 --scale 25
run s27 --scale 27
