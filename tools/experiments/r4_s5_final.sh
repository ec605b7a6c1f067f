#!/bin/bash
# closing session of round 4, last GPU call: smoke() and the default bench line on the final tree
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > $out/bench_default_final.json 2> $out/bench_default_final.err; cut -c1-200 $out/bench_default_final.json
