#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
for sl in 16 24 32; do
  echo "== 4 shards, sweep_slices=$sl"
  python tools/shard_emulation.py --scale 26 --nshards 4 --shards 0 --iters 10 --lib-option sweep_slices=$sl
done
for sl in 24 32 48; do
  echo "== 2 shards, sweep_slices=$sl"
  python tools/shard_emulation.py --scale 26 --nshards 2 --shards 0 --iters 10 --lib-option sweep_slices=$sl
done
echo "== round 5's path (sweep_slices=0), 4 and 2 shards"
python tools/shard_emulation.py --scale 26 --nshards 4 --shards 0 --iters 10 --lib-option sweep_slices=0
python tools/shard_emulation.py --scale 26 --nshards 2 --shards 0 --iters 10 --lib-option sweep_slices=0
echo "== 8 shards, 24 slices, only the very longest rows giant"
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=24 --lib-option giant_row=262144
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=24 --lib-option giant_row=131072
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/shard_slices_sweep2.txt
cat gpurun_out/r6/shard_slices_sweep2.txt
