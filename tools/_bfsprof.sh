cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/bfsprof -o bfs -- python /root/repo/tools/bfs_bench.py --scale 26 $BFS_ARGS > /tmp/bfsprof.log 2>&1
db=$(ls /tmp/bfsprof/*/*.db /tmp/bfsprof/*.db 2>/dev/null | head -1)
python /root/repo/tools/per_dispatch.py $db k_ | grep -v "k_rmat\|k_make\|k_deal\|k_degree\|k_rank\|k_count\|k_or\|k_seg\|k_row\|k_build\|k_mark\|k_hist" | tail -150
