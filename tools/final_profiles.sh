#!/bin/bash
# Round summary profiles for the exact bench commands (run through gpurun):
#   kernel-trace stats, then PMC passes (each in its own rocprofv3 run) for HBM traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; tag=$1; shift
for sc in "$@"; do
  out=$R/gpurun_out/final_$sc; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out -o kt -- python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 > $out/bench_under_rocprof.json 2> $out/kt.err
  python tools/prof_summary.py $out/kt_results.db > $out/${tag}_scale${sc}_kernel_stats.md
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set -d $out -o pmc_$n -- python bench.py --scale $sc --steps 5 --warmup 1 --cpu-scale 0 --no-timing > /dev/null 2> $out/pmc_$n.err
    python tools/prof_summary.py $out/pmc_${n}_results.db | grep -E "counter|k_spmv|k_giant|k_send|k_apply" | grep -v Degree > $out/${tag}_scale${sc}_pmc_$n.md
  done
  python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 > $out/bench.json 2> $out/bench.err
  grep summary $out/bench.err; rm -f $out/*.db
done
# the other BASELINE configurations (text summaries)
out=$R/gpurun_out/final_misc; mkdir -p $out
python tools/bfs_bench.py --scale 26 2>&1 | grep "^BFS" > $out/bfs_scale26.txt
python tools/sgd_bench.py --users 200000 --items 20000 2>&1 | grep "^SGD" > $out/sgd.txt
python tools/sgd_bench.py 2>&1 | grep "^SGD" >> $out/sgd.txt
python tools/sgd_bench.py --users 10000000 --items 1000000 --iters 3 2>&1 | grep "^SGD" >> $out/sgd.txt
python tools/app_at_scale.py 20 2>&1 | grep "==" > $out/apps.txt
python tools/app_at_scale.py 22 2>&1 | grep "==" >> $out/apps.txt
