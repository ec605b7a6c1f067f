mkdir -p gpurun_out/r6/ssspprof; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/app_at_scale.py 23 > /dev/null 2>&1
for mode in auto one; do
  if [ $mode = one ]; then export GRAPHMAT_COL_TILES=1; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6/ssspprof/$mode -o kt -- $R/build/ref_apps/SSSP /tmp/rmat23.bin.mtx 1 > $R/gpurun_out/r6/ssspprof/$mode.log 2>&1
  python $R/tools/prof_summary.py $(find $R/gpurun_out/r6/ssspprof/$mode -name "*.db" | head -1) > $R/gpurun_out/r6/ssspprof/$mode.md 2>&1
  find $R/gpurun_out/r6/ssspprof/$mode -name "*.db" -delete
done
head -40 $R/gpurun_out/r6/ssspprof/auto.md | cut -c1-200; echo ====; head -40 $R/gpurun_out/r6/ssspprof/one.md | cut -c1-200
