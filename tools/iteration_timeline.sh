cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/itertl; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o it -- python bench.py --scale 26 --steps 3 --warmup 1 --cpu-scale 0 --no-extra --no-timing > /dev/null 2> $out/it.err
python tools/prof_timeline.py $out/it_results.db --match "k_spmv|k_giant|k_apply|k_send|k_short" --last ${TL_LAST:-44} > $out/timeline.md
rm -f $out/*.db
cat $out/timeline.md
