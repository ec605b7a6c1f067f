#!/bin/bash
# PMC comparison of the untiled and the column-tiled multiply (run through gpurun): L2 hits/misses per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
out=$R/gpurun_out/tilepmc; mkdir -p $out
for t in "$@"; do
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $out -o pmc_t$t -- python bench.py --scale 26 --steps 3 --warmup 1 --cpu-scale 0 --no-timing --col-tiles $t > /dev/null 2> $out/pmc_t$t.err
  python tools/prof_summary.py $out/pmc_t${t}_results.db | grep -E "counter|k_spmv|k_giant" | grep -v Degree > $out/pmc_tiles$t.md
  rocprofv3 --kernel-trace --stats -d $out -o kt_t$t -- python bench.py --scale 26 --steps 5 --warmup 1 --cpu-scale 0 --col-tiles $t > /dev/null 2> $out/kt_t$t.err
  python tools/prof_summary.py $out/kt_t${t}_results.db > $out/kt_tiles$t.md
  rm -f $out/*.db
done
