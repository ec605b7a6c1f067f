#!/bin/bash
# round 5, experiment 11: where a wave of k_spmv_sell spends its time (phase clocks)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e11; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
run() { name=$1; shift; timeout 600 build/sweep_lib_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $@"; cat $out/$name.txt | tail -n 12; }
run s26 26 3
run s24 24 3
