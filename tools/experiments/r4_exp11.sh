#!/bin/bash
# round 4, experiment 11: the DPP chain fold in the one-wave-per-row kernel: parity, then the bench at RMAT-26 / 22 / 25
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e11; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
for sc in 26 26 22 25 24; do timeout 600 python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/b$sc.json 2> $out/b$sc.err; grep summary $out/b$sc.err | cut -c1-150; done
timeout 600 python tools/shard_emulation.py --staged --shards 0 2>&1 | grep -E "wall clock|shard 0 of" | cut -c1-300
