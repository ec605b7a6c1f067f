#!/bin/bash
# round 5, experiment 14: as many folding waves as the long rows need; calibration of the medium / long border rule
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e14; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sweep" 2>&1 | tail -n 3
runt() { name=$1; shift; timeout 600 build/sweep_lib_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $@ :: $(grep 'border' $out/$name.txt) :: $(grep "library's form" $out/$name.txt | cut -c60-90)"; }
runt t26 26 3
runt t26_f50 26 3 sweep_fold_share=50
runt t26_f100 26 3 sweep_fold_share=100
for L in 4096 2048 1024; do runt t25_l$L 25 3 sweep_long_row=$L; done
for L in 4096 2048 1024 512; do runt t24_l$L 24 3 sweep_long_row=$L; done
runt t25 25 3
runt t24 24 3
runt t27 27 3
runt t27_l2048 27 3 sweep_long_row=2048
