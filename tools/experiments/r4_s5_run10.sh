#!/bin/bash
# closing session of round 4, GPU call 10: the sweep prototype with MORE, SMALLER workgroups (2-3 per CU, smaller hot sets, more waves per CU)
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/graphmat_amd:$LD_LIBRARY_PATH
for v in f g h; do timeout 300 build/sweep_bench_$v 26 64 5 1 0 2>&1 | tail -12; echo "rc $?"; done
