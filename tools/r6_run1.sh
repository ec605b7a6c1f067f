#!/bin/bash
# round 6, first GPU run: the sharded sweep against the oracle (2 and 3 ranks over gloo on one GPU)
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
for w in 2 3; do
  GM_BACKEND=gloo GM_SCALE=15 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + w)) tools/multi_sweep_check.py > gpurun_out/r6/sweep_multi_$w.log 2>&1
  echo "world $w rc=$?"; tail -5 gpurun_out/r6/sweep_multi_$w.log
done
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sweep or tile_structure" > gpurun_out/r6/tiles.log 2>&1; tail -3 gpurun_out/r6/tiles.log
