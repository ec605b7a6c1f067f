#!/bin/bash
# bench.py with NP ranks sharing ONE GPU, collectives over gloo: exercises the sharded data path
# (two-stage overlapped exchange and, with --no-overlap, the plain one) where only a 1-GPU box is
# available.  Timings are meaningless (the ranks time-slice one device); look at the self-check
# and at config.exchange in the JSON line.   NP=3 SCALE=22 bash tools/multi_bench_gloo.sh
export GM_BENCH_BACKEND=gloo
for extra in "" "--no-overlap"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NP:-2} --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus ${NP:-2} --scale ${SCALE:-22} --steps 10 --warmup 2 $extra 2>&1 | grep "summary\|rror\|disagrees\|\"exchange\"" | cut -c1-400
done
