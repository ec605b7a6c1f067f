# per-kernel times of the bench iteration at RMAT-26 (short rows through the sweep)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6/streamprof; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o kt -- python $R/bench.py --scale ${1:-26} --steps 10 --warmup 2 --cpu-scale 0 --no-extra $2 > $out/bench.json 2> $out/err.txt
PROF_ROWS=60 python $R/tools/prof_summary.py $(find $out -name "*.db" | head -1) > $out/stats.md 2>&1
find $out -name "*.db" -delete
grep -E "sell|short_fold|rowblock|giant|apply" $out/stats.md | cut -c1-200 | head -20
