#!/bin/bash
# Round-5 summary profiles for the exact bench commands (run through gpurun): kernel-trace stats and timeline, PMC passes (each in
# its own rocprofv3 run, kernel-trace only) for memory-side traffic, the calibration of those counters on known access patterns
# (tools/pmc_calibrate.hip), the TCP passes of the multiply kernels, the sweep's phase clocks, and the unchanged reference apps.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; tag=r05
export LD_LIBRARY_PATH=$R/graphmat_amd
for sc in 26 22; do
  out=$R/gpurun_out/final_$sc; mkdir -p $out
  timeout 900 rocprofv3 --kernel-trace --stats -d $out -o kt -- python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/${tag}_scale${sc}_bench.json 2> $out/kt.err
  python tools/prof_summary.py $out/kt_results.db > $out/${tag}_scale${sc}_kernel_stats.md
  [ $sc = 26 ] && python tools/prof_timeline.py $out/kt_results.db --match "k_spmv|k_giant|k_apply|k_send" --last 18 > $out/${tag}_iteration_timeline_scale26.md 2>/dev/null
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    timeout 900 rocprofv3 --kernel-trace --pmc $set -d $out -o pmc_$n -- python bench.py --scale $sc --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra > /dev/null 2> $out/pmc_$n.err
    python tools/prof_summary.py $out/pmc_${n}_results.db | grep -E "counter|k_spmv|k_giant|k_send|k_apply" | grep -v Degree > $out/${tag}_scale${sc}_pmc_$n.md
  done
  rm -f $out/*.db
done
out=$R/gpurun_out/final_26
# the counters on known access patterns
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d $out -o cal_$n -- build/pmc_calibrate > $out/cal_$n.txt 2> $out/cal_$n.err
  python tools/prof_summary.py $out/cal_${n}_results.db | grep -E "counter|k_cal" > $out/${tag}_pmc_calibration_$n.md
  rm -f $out/cal_${n}_results.db
done
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d $out -o tcp$i -- python bench.py --scale 26 --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra > /dev/null 2> $out/tcp$i.err
  python tools/prof_summary.py $out/tcp${i}_results.db | grep -E "counter|k_spmv|k_giant" | grep -v Degree > $out/${tag}_tcp$i.md
  rm -f $out/tcp${i}_results.db
done
timeout 600 build/sweep_lib_bench 26 3 > $out/${tag}_sweep_lib_bench_rmat26.txt 2>&1
{
echo "# unchanged reference apps (build/ref_apps), exact by default (no trait, no probe, no environment variables)"
python tools/app_at_scale.py 22 2>&1 | grep "=="
python tools/app_at_scale.py 26 2>&1 | grep "=="
echo "# GRAPHMAT_TRUST_PROBE=1"
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 22 2>&1 | grep "=="
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 26 2>&1 | grep "=="
} > $out/${tag}_unchanged_apps.txt
cat $out/${tag}_tcp*.md | grep -v "^| kernel" | cut -c1-170
cat $out/${tag}_pmc_calibration_*.md | cut -c1-170
cat $out/${tag}_unchanged_apps.txt | cut -c1-220
