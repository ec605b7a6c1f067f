#!/bin/bash
# round 5, experiment 27: batches without end-of-rows tests (empty meta rows behind the stream) and a fold without a branch per padding test
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e27; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
for sc in 26 24; do timeout 900 build/sweep_lib_bench $sc 4 > $out/t$sc.txt 2>&1; echo "== RMAT-$sc"; grep -i "library\|before\|differ\|pad-row\|padding" $out/t$sc.txt | cut -c1-150; done

