#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1; tail -3 gpurun_out/r6/build.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6/parity.log 2>&1; tail -5 gpurun_out/r6/parity.log
{
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 1 7 --iters 10
python tools/shard_emulation.py --scale 26 --nshards 4 --shards 0 --iters 10
python tools/shard_emulation.py --scale 26 --nshards 2 --shards 0 --iters 10
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/shard_emulation_replay_maps.txt
cat gpurun_out/r6/shard_emulation_replay_maps.txt
bash tools/sweep.sh 26 "--no-extra" 2>&1 | grep -v amdgpu > gpurun_out/r6/single_after_replay.txt; cat gpurun_out/r6/single_after_replay.txt
