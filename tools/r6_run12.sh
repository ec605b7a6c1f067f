#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/r6; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "giant or long or replay or pagerank" > $out/parity_giant.log 2>&1; tail -2 $out/parity_giant.log
rocprofv3 --kernel-trace --stats -d $out -o kt_shard -- python tools/shard_emulation.py --nshards 8 --shards 0 --iters 10 > $out/kt_shard.log 2> $out/kt_shard.err
python tools/prof_summary.py $out/kt_shard_results.db > $out/kt_shard0_of_8_replay.md
rm -f $out/*.db
grep "giant\|sell\|rowblock" $out/kt_shard0_of_8_replay.md
grep -v amdgpu $out/kt_shard.log
bash tools/sweep.sh 26 "--no-extra" 2>&1 | grep -v amdgpu
