#!/bin/bash
# round 5, experiment 19: batch sizes and a pipeline two batches deep in k_spmv_sell (tools/sweep_lib_bench.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e19; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 build/sweep_lib_bench_ub 26 4 > $out/t26.txt 2>&1; grep -v "phase\|waves:" $out/t26.txt | tail -n 22
timeout 900 build/sweep_lib_bench_ub 25 4 > $out/t25.txt 2>&1; grep -v "phase\|waves:" $out/t25.txt | grep "library\|deep\|batches" 
