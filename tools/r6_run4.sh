#!/bin/bash
# round 6: kernel trace of shard 0 of 8 with the sharded sweep (24 slices) + giant-row threshold variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/r6; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
rocprofv3 --kernel-trace --stats -d $out -o kt_shard -- python tools/shard_emulation.py --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=24 > $out/kt_shard.log 2> $out/kt_shard.err
python tools/prof_summary.py $out/kt_shard_results.db > $out/kt_shard0_of_8_slices24.md
rm -f $out/*.db
cat $out/kt_shard0_of_8_slices24.md | head -30
{
for gr in 16384 32768 65536; do for sl in 20 24 28; do
  echo "== 8 shards, sweep_slices=$sl giant_row=$gr"
  python tools/shard_emulation.py --scale 26 --nshards 8 --shards 0 --iters 10 --lib-option sweep_slices=$sl --lib-option giant_row=$gr
done; done
} 2>&1 | grep -v amdgpu.ids > $out/shard_giant_threshold.txt
cat $out/shard_giant_threshold.txt
