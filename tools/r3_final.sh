#!/bin/bash
# round 3 closing run: whole GPU suite, then the profile set + default bench line (tools/refresh_profiles.sh), shard emulation
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3final; mkdir -p $out
timeout 1700 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -6 $out/pytest_gpu.txt
bash tools/refresh_profiles.sh r03 > $out/refresh.log 2>&1
tail -5 $out/refresh.log | cut -c1-300
tail -4 gpurun_out/final_default/bench.err | cut -c1-300
python tools/shard_emulation.py --staged --shards 0 1 2>&1 | grep -v amdgpu > $out/shards.txt; cat $out/shards.txt | cut -c1-260
bash tools/iteration_timeline.sh > /dev/null 2>&1; cp gpurun_out/itertl/timeline.md $out/r03_iteration_timeline_scale26.md
cp profiles/pmc_traffic.json $out/pmc_traffic.json; cp profiles/r03_* $out/ 2>/dev/null
