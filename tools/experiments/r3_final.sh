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
cp profiles/pmc_traffic.json $out/pmc_traffic.json; cp profiles/r03_* $out/ 2>/dev/null
bash tools/iteration_timeline.sh > /dev/null 2>&1; cp gpurun_out/itertl/timeline.md $out/r03_iteration_timeline_scale26.md
bash tools/bfs_timeline.sh > /dev/null 2>&1
{ echo "# BFS RMAT-26, rocprofv3 kernel trace of tools/bfs_bench.py: the kernels of the last traversal (source 777), level by level"; echo "# (start / duration / gap in us; fills are the per-level clears; the host decides push / pull between levels)"; grep -n "k_frontier_stats" gpurun_out/bfstl/bfs_timeline.md | tail -1 | cut -d: -f1 > /tmp/bfs_first_line; head -2 gpurun_out/bfstl/bfs_timeline.md; tail -n +$(cat /tmp/bfs_first_line) gpurun_out/bfstl/bfs_timeline.md | grep -v "at::native\|rocprim" | head -90; } > $out/r03_bfs_timeline_scale26.md
