#!/bin/bash
# round 5, experiment 26: the library's short-row kernel next to a sweep that leaves it LDS and registers (no long rows, pool cut by 36 KB)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e26; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
for sc in 26; do timeout 900 build/sweep_lib_bench $sc 4 > $out/t$sc.txt 2>&1; echo "== RMAT-$sc"; grep -i "library\|rowblock\|two streams\|without long" $out/t$sc.txt | cut -c1-150; done
