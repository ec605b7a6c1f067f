#!/bin/bash
# PMC passes over one bench command (each counter set in its own rocprofv3 run, kernel-trace only)
# usage: tools/pmc.sh <outdir> <scale> "<set1>" "<set2>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/$1; sc=$2; shift 2
mkdir -p $out; cd $R
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out -o pmc$i -- python bench.py --scale $sc --steps 5 --warmup 1 --cpu-scale 0 --no-timing > $out/pmc$i.json 2> $out/pmc$i.err
  python tools/prof_summary.py $out/pmc${i}_results.db | grep -E "k_spmv|k_send|k_apply|counter" | grep -v Degree > $out/pmc$i.md
done
