#!/bin/bash
# L2 behaviour and kernel times of one shard of N against the whole graph (run through gpurun): is a shard's multiply
# slower per edge than the single-GPU one, and why?   usage: tools/shard_pmc.sh <nshards> <shard>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/shardpmc; mkdir -p $out
N=${1:-8}; S=${2:-1}
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $out -o pmc_shard -- python tools/shard_emulation.py --nshards $N --shards $S --iters 3 > $out/pmc_shard.log 2> $out/pmc_shard.err
python tools/prof_summary.py $out/pmc_shard_results.db | grep -E "counter|k_spmv|k_giant|k_apply|k_send" | grep -v Degree > $out/pmc_shard${S}_of_$N.md
rocprofv3 --kernel-trace --stats -d $out -o kt_shard -- python tools/shard_emulation.py --nshards $N --shards $S --iters 5 > $out/kt_shard.log 2> $out/kt_shard.err
python tools/prof_summary.py $out/kt_shard_results.db > $out/kt_shard${S}_of_$N.md
rm -f $out/*.db
