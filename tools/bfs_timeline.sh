cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/bfstl; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o bfs -- python tools/bfs_bench.py --scale 26 > $out/bfs.log 2> $out/bfs.err
python tools/prof_timeline.py $out/bfs_results.db --match "^(?!.*(at::native|rocprim|copyBuffer))" --last 150 > $out/bfs_timeline.md
rm -f $out/*.db
tail -n 4 $out/bfs.log
