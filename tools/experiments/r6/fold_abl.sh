# what the short rows' fold kernel spends its time on: GM_FOLD_ABL bit 0 = no product loads, bit 1 = no row folds, bit 2 = return after the products are in LDS
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for abl in ${ABLS:-0 2 4}; do
  out=$R/gpurun_out/r6/foldabl_$abl; mkdir -p $out
  GM_FOLD_ABL=$abl rocprofv3 --kernel-trace --stats -d $out -o kt -- python $R/bench.py --scale 26 --steps 6 --warmup 1 --cpu-scale 0 --no-extra > $out/bench.json 2> $out/err.txt
  python $R/tools/prof_summary.py $(find $out -name "*.db" | head -1) > $out/stats.md 2>&1
  find $out -name "*.db" -delete
  echo "abl=$abl $(grep -E "short_fold" $out/stats.md | cut -c1-140)"
done
