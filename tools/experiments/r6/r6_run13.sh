#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/r6; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
rocprofv3 --kernel-trace -d $out -o kt_shard -- python tools/shard_emulation.py --nshards 8 --shards 0 --iters 10 > $out/kt_shard.log 2> $out/kt_shard.err
python tools/per_dispatch.py $out/kt_shard_results.db k_giant_replay > $out/replay_dispatches_shard.txt
rm -f $out/*.db
cat $out/replay_dispatches_shard.txt
rocprofv3 --kernel-trace -d $out -o kt_single -- python bench.py --scale 26 --steps 10 --warmup 2 --cpu-scale 0 --no-extra > $out/kt_single.log 2> $out/kt_single.err
python tools/per_dispatch.py $out/kt_single_results.db k_giant_replay > $out/replay_dispatches_single.txt
rm -f $out/*.db
cat $out/replay_dispatches_single.txt
