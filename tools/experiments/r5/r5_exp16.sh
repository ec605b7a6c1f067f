#!/bin/bash
# round 5, experiment 16: the long rows' stage filled without a barrier, staging shares dealt against the waves' group rows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e16; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sweep" 2>&1 | tail -n 3
runt() { name=$1; shift; timeout 600 build/sweep_lib_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $@"; grep -v "^  \.\.\.\|against" $out/$name.txt | tail -n 4; }
runt t26 26 3
runt t26_f100 26 3 sweep_fold_share=100
runt t26_f40 26 3 sweep_fold_share=40
runt t24 24 3
runt t25 25 3
runt t27 27 3
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s25 --scale 25
run s24 --scale 24
run s27 --scale 27
