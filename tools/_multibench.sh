export GM_BENCH_BACKEND=gloo
for extra in "" "--no-overlap"; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NP:-2} --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus ${NP:-2} --scale ${SCALE:-22} --steps 10 --warmup 2 $extra 2>&1 | grep -v "^$" | grep "summary\|Error\|error\|overlapped\|disagrees" | cut -c1-330
done
