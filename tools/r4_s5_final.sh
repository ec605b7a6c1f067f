#!/bin/bash
# closing session of round 4, last GPU call: the whole GPU suite and smoke() on the final tree, the default bench line (with the traffic figure of
# the refreshed pmc_traffic.json), and two records: every non-giant wave row in the sweep at 3 tiles, and the shard emulation against the new single-GPU time
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
( timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_final.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu_final.txt ); tail -3 $out/pytest_gpu_final.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py > $out/bench_default_final.json 2> $out/bench_default_final.err; cut -c1-200 $out/bench_default_final.json
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
f=$out/sweep2_t3; timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --lib-option sweep_slices=2 > $f.json 2> $f.err; echo "sweep_slices=2 at 3 tiles: $(sm $f.err)"
timeout 900 python tools/shard_emulation.py --staged --shards 0 1 2>&1 | cut -c1-330 | tee $out/shard_emulation_final.txt
