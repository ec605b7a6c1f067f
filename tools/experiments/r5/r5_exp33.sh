#!/bin/bash
# round 5, experiment 33: the sweep under a register cap (97 VGPRs, 22 spilled) with the LDS-free short-row kernel next to it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e33; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd:$R/build
timeout 900 build/sweep_lib_bench 26 4 > $out/t26.txt 2>&1; grep -i "library's form\|alone\|two streams\|launched first" $out/t26.txt | cut -c1-150
