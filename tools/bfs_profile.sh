#!/bin/bash
# Per-dispatch kernel times of tools/bfs_bench.py under rocprofv3 (run through gpurun):
#   BFS_ARGS="--debug-flags 64" bash tools/bfs_profile.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace -d /tmp/bfsprof -o bfs -- python $R/tools/bfs_bench.py --scale ${SCALE:-26} $BFS_ARGS > /tmp/bfsprof.log 2>&1
db=$(ls /tmp/bfsprof/*/*.db /tmp/bfsprof/*.db 2>/dev/null | head -1)
python $R/tools/per_dispatch.py $db k_ | grep -v "k_rmat\|k_make\|k_deal\|k_degree\|k_rank\|k_count\|k_or\|k_seg\|k_row\|k_build\|k_hist" | tail -150
