#!/bin/bash
# Round-6 summary profiles for the exact bench commands (run through gpurun): kernel-trace stats and timeline, PMC passes (each in its own
# rocprofv3 run, kernel-trace only) for memory-side traffic, the calibration of those counters on known access patterns
# (tools/pmc_calibrate.hip), the BFS timeline, one shard of 8 (kernel stats), and the unchanged reference apps.
cd /tmp && export TMPDIR=/tmp PROF_ROWS=40
R=$GRAFT_REPO_ROOT; cd $R; tag=r06
export LD_LIBRARY_PATH=$R/graphmat_amd
mkdir -p build
[ -x build/pmc_calibrate ] || hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o build/pmc_calibrate
for sc in 26 22; do
  out=$R/gpurun_out/final_$sc; mkdir -p $out
  timeout 900 rocprofv3 --kernel-trace --stats -d $out -o kt -- python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 --no-extra > $out/${tag}_scale${sc}_bench.json 2> $out/kt.err
  python tools/prof_summary.py $out/kt_results.db > $out/${tag}_scale${sc}_kernel_stats.md
  [ $sc = 26 ] && python tools/prof_timeline.py $out/kt_results.db --match "k_spmv|k_short|k_giant|k_apply|k_send" --last 18 > $out/${tag}_iteration_timeline_scale26.md 2>/dev/null
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    timeout 900 rocprofv3 --kernel-trace --pmc $set -d $out -o pmc_$n -- python bench.py --scale $sc --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra > /dev/null 2> $out/pmc_$n.err
    python tools/prof_summary.py $out/pmc_${n}_results.db | grep -E "counter|k_spmv|k_short|k_giant|k_send|k_apply" | grep -v Degree > $out/${tag}_scale${sc}_pmc_$n.md
  done
  rm -f $out/*.db
done
out=$R/gpurun_out/final_26
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d $out -o cal_$n -- build/pmc_calibrate > $out/cal_$n.txt 2> $out/cal_$n.err
  python tools/prof_summary.py $out/cal_${n}_results.db | grep -E "counter|k_cal" > $out/${tag}_pmc_calibration_$n.md
  rm -f $out/cal_${n}_results.db
done
# one shard of 8 (shard 0: it owns the hub row), compute only
timeout 900 rocprofv3 --kernel-trace --stats -d $out -o kt_shard -- python tools/shard_emulation.py --nshards 8 --shards 0 --iters 20 > $out/kt_shard.log 2> $out/kt_shard.err
python tools/prof_summary.py $out/kt_shard_results.db | grep -E "kernel|---|k_spmv|k_short|k_giant|k_apply|k_send" | grep -v Degree > $out/${tag}_shard0_of_8_kernel_stats.md
python tools/prof_timeline.py $out/kt_shard_results.db --match "k_spmv|k_short|k_giant|k_apply|k_send" --last 16 > $out/${tag}_shard0_of_8_iteration_timeline.md 2>/dev/null
rm -f $out/kt_shard_results.db
# BFS RMAT-26, level by level
timeout 900 rocprofv3 --kernel-trace -d $out -o bfs -- python tools/bfs_bench.py --scale 26 > $out/bfs.log 2> $out/bfs.err
python tools/prof_timeline.py $out/bfs_results.db --match "^(?!.*(at::native|rocprim|copyBuffer))" --last 150 | grep -v "at::native" > $out/${tag}_bfs_timeline_scale26.md
tail -n 4 $out/bfs.log > $out/${tag}_bfs_bench_scale26.txt
rm -f $out/bfs_results.db
{
echo "# unchanged reference apps (build/ref_apps), exact by default (no trait, no probe, no environment variables)"
python tools/app_at_scale.py 22 2>&1 | grep "=="
python tools/app_at_scale.py 26 2>&1 | grep "=="
} > $out/${tag}_unchanged_apps.txt
cat $out/${tag}_pmc_calibration_*.md | cut -c1-170
cat $out/${tag}_unchanged_apps.txt | cut -c1-220
cat $out/${tag}_shard0_of_8_kernel_stats.md | cut -c1-200
