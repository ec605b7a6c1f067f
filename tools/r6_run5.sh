#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=24
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=24 --lib-option giant_maps=0
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=24 --lib-option giant_row=32768
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/shard_giant_counters.txt
cat gpurun_out/r6/shard_giant_counters.txt
