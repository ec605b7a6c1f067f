#!/bin/bash
# closing session of round 4, GPU call 1: the GPU suite and the default bench at HEAD, then the row-stationary sweep of ONE SHARD's
# medium rows (tools/sweep_bench.hip with its shard arguments: every 8th row of the length ranking against the whole x)
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/graphmat_amd:$LD_LIBRARY_PATH
( timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.txt )
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
for T in 8 16 32 64; do
  timeout 300 build/sweep_bench 26 $T 5 8 0 >> $out/shard_sweep.txt 2>&1
done
timeout 300 build/sweep_bench 26 16 5 8 3 >> $out/shard_sweep.txt 2>&1
timeout 300 build/sweep_bench 26 64 5 1 0 >> $out/shard_sweep.txt 2>&1
tail -3 $out/pytest_gpu.txt; cat $out/bench_default.json | cut -c1-400; cat $out/shard_sweep.txt
