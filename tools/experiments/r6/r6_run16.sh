# round 6, last session: a shard's short rows through its sweep (stream groups on gm_sweep_t.nsub > 1 structures) -- per-shard cost on one GPU, with and without
mkdir -p gpurun_out/r6
{
for n in 8 4 2; do
  python tools/shard_emulation.py --nshards $n --shards 0 --iters 20 --staged 2>&1 | grep -v amdgpu.ids
done
python tools/shard_emulation.py --nshards 8 --shards 1 7 --iters 20 2>&1 | grep -v amdgpu.ids
echo "== the same with the row-block kernel for the short rows (sweep_form bit 7): shard 0 of 8 / 2"
python tools/shard_emulation.py --nshards 8 --shards 0 --iters 20 --lib-option sweep_form=128 2>&1 | grep -v amdgpu.ids
python tools/shard_emulation.py --nshards 2 --shards 0 --iters 20 --lib-option sweep_form=128 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6/shard_stream.txt 2>&1
cut -c1-250 gpurun_out/r6/shard_stream.txt
