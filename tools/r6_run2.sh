#!/bin/bash
# round 6: shard emulation with the sharded sweep + multi-rank checks over the native exchange (shm stand-in)
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
SHM=$(python -c "from tests.support import build as b; print(b.build())")
for w in 2 3; do
  GM_BACKEND=gloo GM_SCALE=15 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + w)) tools/multi_sweep_check.py > gpurun_out/r6/sweep_multi_$w.log 2>&1
  echo "world $w callback rc=$?"; grep "SWEEP_MULTI\|^rank" gpurun_out/r6/sweep_multi_$w.log | head
  GRAPHMAT_RCCL_LIBRARY=$SHM GM_EXCHANGE=native GM_BACKEND=gloo GM_SCALE=16 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29600 + w)) tools/multi_sweep_check.py > gpurun_out/r6/sweep_multi_native_$w.log 2>&1
  echo "world $w native rc=$?"; grep "SWEEP_MULTI\|^rank" gpurun_out/r6/sweep_multi_native_$w.log | head
done
{
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 1 7 --iters 10
python tools/shard_emulation.py --scale 26 --nshards 4 --shards 0 --iters 10
python tools/shard_emulation.py --scale 26 --nshards 2 --shards 0 --iters 10
echo "== sweep_slices=0 (round 5's sharded path)"
python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=0
} > gpurun_out/r6/shard_emulation_rmat26.txt 2>&1
cat gpurun_out/r6/shard_emulation_rmat26.txt | grep -v amdgpu.ids
