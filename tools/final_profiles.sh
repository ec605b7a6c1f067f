#!/bin/bash
# Round summary profiles for the exact bench commands (run through gpurun):
#   kernel-trace stats, then PMC passes (each in its own rocprofv3 run) for HBM traffic
# usage: bash tools/final_profiles.sh <tag> <scale> [<scale> ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; tag=$1; shift
for sc in "$@"; do
  out=$R/gpurun_out/final_$sc; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out -o kt -- python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/bench_under_rocprof.json 2> $out/kt.err
  python tools/prof_summary.py $out/kt_results.db > $out/${tag}_scale${sc}_kernel_stats.md
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set -d $out -o pmc_$n -- python bench.py --scale $sc --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra > /dev/null 2> $out/pmc_$n.err
    python tools/prof_summary.py $out/pmc_${n}_results.db | grep -E "counter|k_spmv|k_giant|k_send|k_apply" | grep -v Degree > $out/${tag}_scale${sc}_pmc_$n.md
  done
  python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/bench.json 2> $out/bench.err
  grep summary $out/bench.err; rm -f $out/*.db
done
# the other BASELINE configurations (text summaries)
out=$R/gpurun_out/final_misc; mkdir -p $out
python tools/bfs_bench.py --scale 26 2>&1 | grep "^BFS" > $out/bfs_scale26.txt
python tools/sgd_bench.py --users 200000 --items 20000 2>&1 | grep "^SGD" > $out/sgd.txt
python tools/sgd_bench.py 2>&1 | grep "^SGD" >> $out/sgd.txt
python tools/sgd_bench.py --users 10000000 --items 1000000 --iters 3 2>&1 | grep "^SGD" >> $out/sgd.txt
# SGD K=128: instruction mix / MFMA evidence (one PMC pass, kernel-trace only)
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU_MFMA[A-Z0-9_]*\|SQ_INSTS_MFMA\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_VALU\b\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_LDS\|SQ_INSTS_VMEM_RD\|GRBM_GUI_ACTIVE" | sort -u > $out/counters_available.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $out -o sgd_pmc -- python tools/sgd_bench.py --users 2000000 --items 200000 --iters 2 > /dev/null 2> $out/sgd_pmc.err
python tools/prof_summary.py $out/sgd_pmc_results.db 2>/dev/null | grep -E "counter|k_sgd" > $out/${tag}_sgd_k128_pmc.md
rocprofv3 --kernel-trace --stats -d $out -o sgd_kt -- python tools/sgd_bench.py --users 2000000 --items 200000 --iters 3 > /dev/null 2> $out/sgd_kt.err
python tools/prof_summary.py $out/sgd_kt_results.db 2>/dev/null | head -12 > $out/${tag}_sgd_k128_kernel_stats.md
python tools/app_at_scale.py 20 2>&1 | grep "==" > $out/apps.txt
python tools/app_at_scale.py 22 2>&1 | grep "==" >> $out/apps.txt
rm -f $out/*.db
