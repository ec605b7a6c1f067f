#!/bin/bash
# round 5, experiment 3: long rows staged without workgroup barriers (an LDS counter; only the folding waves wait)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e3; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
run() { name=$1; shift; timeout 600 build/sell_bench "$@" > $out/$name.txt 2>&1; echo "== $name: $@"; grep -v "differ (bit" $out/$name.txt | tail -n 14; grep "differ (bit" $out/$name.txt | grep -v " 0 of" | head -3; }
run t64      26 64  5 32768 0 4096 2 50
run t80      26 80  5 32768 0 4096 2 50
run t96      26 96  5 32768 0 4096 2 50
run t128     26 128 5 32768 0 4096 2 50
run s25_t64  25 64  5 32768 0 4096 2 50
run s25_t32  25 32  5 32768 0 4096 2 50
run s24_t16  24 16  5 32768 0 4096 2 50
run s24_t32  24 32  5 32768 0 4096 2 50
