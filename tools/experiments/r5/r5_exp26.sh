#!/bin/bash
# round 5, experiment 26: can a register-light, LDS-free kernel for the short rows run NEXT TO the sweep on the same CUs?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e26; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 build/sweep_lib_bench 26 3 > $out/t26.txt 2>&1; grep -A12 "^short rows" $out/t26.txt
