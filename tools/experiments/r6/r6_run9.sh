#!/bin/bash
# round 6: the sharded swept schedule -- multi-rank checks, then bench.py with 8 ranks on the shared-memory stand-in for librccl (one GPU:
# timings meaningless, what counts is that the path runs: distributed build, native exchange cross-checked against the callback, both forms)
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
SHM=$(python -c "from tests.support import build as b; print(b.build())")
for w in 2 3; do
  GM_BACKEND=gloo GM_SCALE=15 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + w)) tools/multi_sweep_check.py > gpurun_out/r6/sweep_multi_$w.log 2>&1
  echo "world $w callback rc=$?"; grep "SWEEP_MULTI\|^rank\|Error\|error" gpurun_out/r6/sweep_multi_$w.log | head
  GRAPHMAT_RCCL_LIBRARY=$SHM GM_EXCHANGE=native GM_BACKEND=gloo GM_SCALE=16 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29600 + w)) tools/multi_sweep_check.py > gpurun_out/r6/sweep_multi_native_$w.log 2>&1
  echo "world $w native rc=$?"; grep "SWEEP_MULTI\|^rank\|Error\|error" gpurun_out/r6/sweep_multi_native_$w.log | head
done
GM_BENCH_BACKEND=gloo GRAPHMAT_RCCL_LIBRARY=$SHM timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29717 \
  bench.py --gpus 8 --scale 20 --steps 5 --warmup 2 --cpu-scale 0 --col-tiles 3 > gpurun_out/r6/bench_8ranks_standin_rmat20.log 2>&1
echo "bench 8 ranks rc=$?"; grep "^\[bench\]\|^{" gpurun_out/r6/bench_8ranks_standin_rmat20.log | cut -c1-600 | tail -20
