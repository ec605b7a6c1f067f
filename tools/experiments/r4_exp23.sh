#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e23; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd:$LD_LIBRARY_PATH
timeout 300 build/sweep_bench 22 16 3 > $out/sweep_22.txt 2>&1; cat $out/sweep_22.txt
for T in 8 32 64; do
  timeout 600 build/sweep_bench 26 $T 4 > $out/sweep_26_$T.txt 2>&1; cat $out/sweep_26_$T.txt
done
