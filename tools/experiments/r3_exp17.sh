#!/bin/bash
# BFS: quarter-wave rows in the grouped bottom-up kernel -- parity, then timings and the per-level kernel timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e17; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x -k "bfs or BFS or sssp or SSSP or last or dropin or apps or topo" > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
python tools/bfs_bench.py --scale 26 2>&1 | grep "^BFS" | cut -c1-250
bash tools/bfs_timeline.sh > /dev/null 2>&1
cp gpurun_out/bfstl/bfs_timeline.md $out/bfs_timeline.md
tail -60 $out/bfs_timeline.md | cut -c1-150
