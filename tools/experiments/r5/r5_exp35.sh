#!/bin/bash
# round 5, experiment 35: what the sweep would do on the rows of a shard (the in-edges of every K-th vertex of RMAT-26 against the whole message vector)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e35; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
for k in 8 4 2; do
  timeout 600 build/sweep_lib_bench 26 3 rows_of=$k > $out/rows_of_$k.txt 2>&1
  echo "== rows_of=$k"; grep -i "rows_of\|RMAT-26:\|border\|library's form\|giant rows as well\|k_spmv_rowblock alone\|short rows 1..64\|differ" $out/rows_of_$k.txt | cut -c1-230
done
