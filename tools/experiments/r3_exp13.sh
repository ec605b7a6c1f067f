#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e13; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o kt -- python bench.py --scale 26 --steps 10 --warmup 2 --cpu-scale 0 --no-extra > /dev/null 2> $out/kt.err
python tools/prof_summary.py $out/kt_results.db | grep -E "k_spmv|k_giant|k_short|k_send|k_apply" | grep -v Degree | cut -c1-200
python tools/prof_timeline.py $out/kt_results.db --match "k_short|k_spmv|k_giant" --last 60 | cut -c1-150
rm -f $out/*.db
