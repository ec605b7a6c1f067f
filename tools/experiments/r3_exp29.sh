#!/bin/bash
# kernel timeline of one shard of 8 (plain loop, do-nothing exchange): where its 1.09 ms go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e29; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o sh -- python tools/shard_emulation.py --shards 0 --iters 6 > $out/sh.log 2> $out/sh.err
python tools/prof_timeline.py $out/sh_results.db --match "k_spmv|k_giant|k_apply|k_send" --last 40 > $out/timeline.md
rm -f $out/*.db
cat $out/timeline.md | cut -c1-150
