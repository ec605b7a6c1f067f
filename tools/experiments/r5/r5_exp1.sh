#!/bin/bash
# round 5, experiment 1: the SELL-layout sweep prototype (tools/sell_bench.hip) against round 4's sweep (tools/sweep_bench.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e1; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 build/sell_bench 26 64 5 4096 1 > $out/sell_26_64_4096.txt 2>&1; tail -n 40 $out/sell_26_64_4096.txt
timeout 900 build/sell_bench 26 64 5 32768 0 > $out/sell_26_64_32768.txt 2>&1; tail -n 20 $out/sell_26_64_32768.txt
timeout 900 build/sell_bench 26 128 5 32768 0 > $out/sell_26_128_32768.txt 2>&1; tail -n 20 $out/sell_26_128_32768.txt
timeout 900 build/sell_bench 26 32 5 32768 0 > $out/sell_26_32_32768.txt 2>&1; tail -n 20 $out/sell_26_32_32768.txt
