#!/bin/bash
# round 5, experiment 9 (cross-slice prefetch): what bounds k_spmv_sell on the library's own structure (tools/sweep_lib_bench.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e9; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
run() { name=$1; shift; timeout 600 build/sweep_lib_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $@"; cat $out/$name.txt | tail -n 12; }
run s26 26 5
run s26_t64 26 5 sweep_slices=64
run s24 24 5
