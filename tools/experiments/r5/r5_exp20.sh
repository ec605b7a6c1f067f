#!/bin/bash
# round 5, experiment 20: the two-batches-deep sweep in the library: parity, bench at five scales
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e20; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | tail -n 3
timeout 900 build/sweep_lib_bench 26 4 > $out/t26.txt 2>&1; grep -v "phase" $out/t26.txt | tail -n 16
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s26b --scale 26
run s25 --scale 25
run s24 --scale 24
run s27 --scale 27
