#!/bin/bash
# round 4, experiment 10: persistent kernels on a shard now that the slice lookup of the hot set is a shift and a mask
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e10; mkdir -p $out
run() { echo "== $*"; timeout 300 python tools/shard_emulation.py --staged --shards 0 "$@" 2>&1 | grep -E "wall clock|shard 0 of" | cut -c1-300; }
{
run
run --lib-option wave16_form=18
run --lib-option rowwave_form=20
run --lib-option wave16_form=18 --lib-option rowwave_form=20
run --lib-option wave16_form=18 --lib-option rowwave_form=20 --lib-option persist_per_cu=1
} > $out/shard_persistent.txt 2>&1
cat $out/shard_persistent.txt
timeout 900 python bench.py --no-extra --cpu-scale 0 --steps 10 > /dev/null 2> $out/b.err; grep summary $out/b.err | cut -c1-120
timeout 900 python bench.py --cpu-scale 0 --steps 10 > $out/bench_extras.json 2> $out/b2.err; grep "extra sgd" $out/b2.err | cut -c1-200
