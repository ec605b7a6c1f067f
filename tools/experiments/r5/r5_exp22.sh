#!/bin/bash
# round 5, experiment 22: slice counts and the folding waves' share with the two-batches-deep kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e22; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
runt() { name=$1; shift; timeout 600 build/sweep_lib_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $(grep "library's form" $out/$name.txt | cut -c60-90) | giants too: $(grep "giant rows as well" $out/$name.txt | cut -c60-82) | 5 rows: $(grep "batches of 5" $out/$name.txt | cut -c60-82) | 7 rows: $(grep "batches of 7" $out/$name.txt | cut -c60-82)"; }
runt t26 26 4
runt t26_s64 26 4 sweep_slices=64
runt t26_s80 26 4 sweep_slices=80
runt t26_s112 26 4 sweep_slices=112
runt t26_s128 26 4 sweep_slices=128
runt t26_f50 26 4 sweep_fold_share=50
runt t26_f60 26 4 sweep_fold_share=60
runt t26_f85 26 4 sweep_fold_share=85
runt t26_f100 26 4 sweep_fold_share=100
runt t25_s32 25 4 sweep_slices=32
runt t25 25 4
runt t25_s64 25 4 sweep_slices=64
runt t25_s80 25 4 sweep_slices=80
