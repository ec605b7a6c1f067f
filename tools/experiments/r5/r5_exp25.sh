#!/bin/bash
# round 5, experiment 25: the cold parts of the first staging round's and the giant rows' gathers requested before the slice barriers
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e25; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
for sc in 26 25 24; do timeout 900 build/sweep_lib_bench $sc 4 > $out/t$sc.txt 2>&1; echo "== RMAT-$sc"; grep "library\|giant rows as\|early\|cold parts" $out/t$sc.txt | cut -c1-130; done
