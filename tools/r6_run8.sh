#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
SHM=$(python -c "from tests.support import build as b; print(b.build())")
for w in 2 3; do
  GM_BACKEND=gloo GM_SCALE=15 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + w)) tools/multi_sweep_check.py > gpurun_out/r6/sweep_multi_$w.log 2>&1
  echo "world $w callback rc=$?"; grep "SWEEP_MULTI\|^rank\|Error\|error" gpurun_out/r6/sweep_multi_$w.log | head
  GRAPHMAT_RCCL_LIBRARY=$SHM GM_EXCHANGE=native GM_BACKEND=gloo GM_SCALE=16 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29600 + w)) tools/multi_sweep_check.py > gpurun_out/r6/sweep_multi_native_$w.log 2>&1
  echo "world $w native rc=$?"; grep "SWEEP_MULTI\|^rank\|Error\|error" gpurun_out/r6/sweep_multi_native_$w.log | head
done
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r6/multi_tests.log 2>&1; tail -3 gpurun_out/r6/multi_tests.log
