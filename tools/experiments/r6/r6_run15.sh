#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 1 7 --iters 20 --staged
python tools/shard_emulation.py --scale 26 --nshards 4 --shards 0 --iters 20 --staged
python tools/shard_emulation.py --scale 26 --nshards 2 --shards 0 --iters 20 --staged
echo "== round 5's sharded path on these sources (sweep_slices=0): shard 0 of 8"
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 20 --staged --lib-option sweep_slices=0
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/shard_emulation_final.txt
cat gpurun_out/r6/shard_emulation_final.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r6/gpu_tests_full.log 2>&1; tail -3 gpurun_out/r6/gpu_tests_full.log
