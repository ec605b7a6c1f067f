#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/r6; mkdir -p $out
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
python tools/app_at_scale.py 26 > /dev/null 2>&1
rocprofv3 --kernel-trace -d $out -o bfsg -- build/ref_apps/BFS /tmp/rmat26.bin.mtx 1 > $out/bfsg.log 2> $out/bfsg.err
python tools/prof_timeline.py $out/bfsg_results.db --match "k_" --last 80 > $out/bfs_guided_timeline.md
rm -f $out/*.db
cut -c1-150 $out/bfs_guided_timeline.md
