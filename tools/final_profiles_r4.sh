#!/bin/bash
# Round-4 summary profiles for the exact bench commands (run through gpurun): kernel-trace stats, PMC passes (each in its own
# rocprofv3 run) for memory-side traffic, and the TCP / TA passes behind profiles/r04_tcp_counters_scale26.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; tag=r04
for sc in 26 22; do
  out=$R/gpurun_out/final_$sc; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out -o kt -- python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/bench_under_rocprof.json 2> $out/kt.err
  python tools/prof_summary.py $out/kt_results.db > $out/${tag}_scale${sc}_kernel_stats.md
  [ $sc = 26 ] && python tools/prof_timeline.py $out/kt_results.db > $out/${tag}_iteration_timeline_scale26.md 2>/dev/null
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set -d $out -o pmc_$n -- python bench.py --scale $sc --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra > /dev/null 2> $out/pmc_$n.err
    python tools/prof_summary.py $out/pmc_${n}_results.db | grep -E "counter|k_spmv|k_giant|k_send|k_apply" | grep -v Degree > $out/${tag}_scale${sc}_pmc_$n.md
  done
  rm -f $out/*.db
done
out=$R/gpurun_out/final_26
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out -o tcp$i -- python bench.py --scale 26 --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra > /dev/null 2> $out/tcp$i.err
  python tools/prof_summary.py $out/tcp${i}_results.db | grep -E "counter|k_spmv|k_giant" | grep -v Degree > $out/${tag}_tcp$i.md
  rm -f $out/tcp${i}_results.db
done
cat $out/${tag}_tcp*.md | grep -v "^| kernel" | cut -c1-170
