#!/bin/bash
# round 5, experiment 4: first run of the SELL sweep inside the library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e4; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sweep" 2>&1 | tail -n 25
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-200)"; tail -n 3 $out/$name.err | cut -c1-300; }
run s26 --scale 26
run s26_f1 --scale 26 --lib-option sweep_form=1
run s26_f2 --scale 26 --lib-option sweep_form=2
run s22 --scale 22
run s24 --scale 24
