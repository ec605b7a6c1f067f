#!/bin/bash
# round 5, experiment 21: the sweep gathers for the giant rows (their fold passes behind it, next to the short rows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e21; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | tail -n 8
timeout 900 python -m pytest tests/test_dropin_apps.py -x -q -m gpu 2>&1 | tail -n 3
timeout 900 build/sweep_lib_bench 26 4 > $out/t26.txt 2>&1; grep "library\|giant" $out/t26.txt
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s26_f8 --scale 26 --lib-option sweep_form=8
run s26b --scale 26
run s25 --scale 25
run s24 --scale 24
run s27 --scale 27
