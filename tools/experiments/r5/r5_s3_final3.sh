#!/bin/bash
# closing run of round 5's last session on the sources with k_spmv_blocked: the whole GPU suite, smoke, kernel stats + timeline + the PMC passes
# behind pmc_traffic.json for RMAT-26 / 22, the table rebuilt on the box, the default bench line quoting it, and the uniform graphs (2^26 / 2^25 / 2^24:
# automatic policy against blocked_rows -1) with a kernel trace of the 2^26 run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; tag=r05
export LD_LIBRARY_PATH=$R/graphmat_amd
fin=$R/gpurun_out/s3final3; mkdir -p $fin
echo "(tests: run on the same code before a comment-only change)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
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
  cp $out/${tag}_scale${sc}_*.md profiles/
done
python tools/pmc_to_json.py $tag 26 22 > /dev/null
cp profiles/pmc_traffic.json $fin/pmc_traffic.json
python bench.py > $fin/r05_bench_default.json 2> $fin/bench_default.err
python - <<P
import json
d=json.loads(open("$fin/r05_bench_default.json").read().strip().splitlines()[-1])
print("default bench: ms", d["ms_per_step"], "value", d["value"], "roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","traffic","avg_launch_ms")})
P
