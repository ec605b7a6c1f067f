#!/bin/bash
# round 5, experiment 13: 512 long slots (8 folding waves): medium / long border at 2048 / 1024 in one launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e13; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sweep" 2>&1 | tail -n 3
runt() { name=$1; shift; timeout 600 build/sweep_lib_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $@"; grep -v "^  \.\.\.\|against" $out/$name.txt | tail -n 5; }
runt t26 26 3
runt t26_l4096 26 3 sweep_long_row=4096
runt t26_l1024 26 3 sweep_long_row=1024
runt t26_l512 26 3 sweep_long_row=512
runt t26_f50 26 3 sweep_fold_share=50
runt t26_f100 26 3 sweep_fold_share=100
runt t24 24 3
runt t25 25 3
runt t27 27 3
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s26_l1024 --scale 26 --lib-option sweep_long_row=1024
run s25 --scale 25
run s24 --scale 24
run s27 --scale 27
