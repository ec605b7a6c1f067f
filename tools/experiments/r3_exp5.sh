#!/bin/bash
# whole GPU suite (no -x), kernel stats with and without giant chunk maps, BFS per-level trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e5; mkdir -p $out
timeout 1700 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -25 $out/pytest_gpu.txt
for gm in 1 0; do
  rocprofv3 --kernel-trace --stats -d $out -o kt$gm -- python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --lib-option giant_maps=$gm > $out/bench_gm$gm.json 2> $out/bench_gm$gm.err
  python tools/prof_summary.py $out/kt${gm}_results.db | grep -E "k_spmv|k_giant|k_send|k_apply" | grep -v Degree > $out/kt_gm$gm.md
  rm -f $out/kt${gm}_results.db
  grep summary $out/bench_gm$gm.err; cat $out/kt_gm$gm.md
done
GRAPHMAT_VERBOSE=1 python tools/bfs_bench.py --scale 26 > $out/bfs_verbose.txt 2>&1
grep -E "active set|iteration|BFS scale|loop done" $out/bfs_verbose.txt | tail -60
