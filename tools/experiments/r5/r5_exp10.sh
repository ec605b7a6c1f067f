#!/bin/bash
# round 5, experiment 10: the sweep with cross-slice prefetch in the bench; slice counts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e10; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sweep" 2>&1 | tail -n 3
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s26_t128 --scale 26 --lib-option sweep_slices=128
run s26_t64 --scale 26 --lib-option sweep_slices=64
run s26b --scale 26
run s25 --scale 25
run s24 --scale 24
run s24_t16 --scale 24 --lib-option sweep_slices=16
run s24_notile --scale 24 --col-tiles 1
run s27 --scale 27
