#!/bin/bash
# round 5, experiment 2: SELL sweep of the medium rows + long rows staged through LDS (tools/sell_bench.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e2; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
run() { name=$1; shift; timeout 600 build/sell_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $@"; grep -v "differ (bit" $out/$name.txt | tail -n 14; grep "differ (bit" $out/$name.txt | grep -v " 0 of" | head -3; }
#            scale T reps row_hi short row_mid fold_waves fold_share
run t64      26 64  5 32768 0 4096 2 100
run t64_s50  26 64  5 32768 0 4096 2 50
run t64_s0   26 64  5 32768 0 4096 2 0
run t128     26 128 5 32768 0 4096 2 50
run t96      26 96  5 32768 0 4096 2 50
run t48      26 48  5 32768 0 4096 2 50
run t64_m2k  26 64  5 32768 0 2048 2 50
run t64_m8k  26 64  5 32768 0 8192 2 50
