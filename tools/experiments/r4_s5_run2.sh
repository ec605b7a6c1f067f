#!/bin/bash
# closing session of round 4, GPU call 2: (1) the sweep prototype with every wave (or quarter workgroup) as its own "workgroup" with private
# rows and no hot set -- whole graph and shard 0 of 8; (2) the class boundaries around the library's sweep once more (tools/r4_exp36.sh:
# its results were lost with the container of the session that ran it)
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/graphmat_amd:$LD_LIBRARY_PATH
for v in b c d e; do
  timeout 300 build/sweep_bench_$v 26 64 5 1 0 >> $out/wave_private_sweep.txt 2>&1
  timeout 300 build/sweep_bench_$v 26 64 5 8 0 >> $out/wave_private_sweep.txt 2>&1
  timeout 300 build/sweep_bench_$v 26 32 5 8 0 >> $out/wave_private_sweep.txt 2>&1
done
cat $out/wave_private_sweep.txt
bash tools/r4_exp36.sh 2>&1 | tee $out/class_boundaries.txt
