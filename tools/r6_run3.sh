#!/bin/bash
# round 6: slice count / giant-row form of the sharded sweep (shard 0 of 8, 4, 2 of RMAT-26, compute only)
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
for sl in 16 24 32 48 64; do for form in 0 8; do
  echo "== 8 shards, sweep_slices=$sl sweep_form=$form"
  python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=$sl --lib-option sweep_form=$form
done; done
for sl in 32 48 64; do
  echo "== 4 shards, sweep_slices=$sl sweep_form=8"
  python tools/shard_emulation.py --scale 26 --nshards 4 --shards 0 --iters 10 --lib-option sweep_slices=$sl --lib-option sweep_form=8
done
for sl in 48 64 80; do
  echo "== 2 shards, sweep_slices=$sl sweep_form=8"
  python tools/shard_emulation.py --scale 26 --nshards 2 --shards 0 --iters 10 --lib-option sweep_slices=$sl --lib-option sweep_form=8
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/shard_slices_sweep.txt
cat gpurun_out/r6/shard_slices_sweep.txt
