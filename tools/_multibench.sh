export GM_BENCH_BACKEND=gloo
GRAPHMAT_VERBOSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --scale ${SCALE:-24} --steps 3 --warmup 1 2>&1 | grep "two-stage" | sort | uniq -c
