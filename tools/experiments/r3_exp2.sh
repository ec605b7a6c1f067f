#!/bin/bash
# persistent wave16 forms: serialized kernel durations (PMC mode serializes dispatches) and L2 requests per form
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e2; mkdir -p $out
for f in 0 1 2 5; do
  rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum -d $out -o f$f -- python bench.py --scale 26 --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra --lib-option wave16_form=$f > /dev/null 2> $out/f$f.err
  echo "== form $f" >> $out/forms.md
  python tools/prof_summary.py $out/f${f}_results.db | grep -E "k_spmv|k_giant" | grep -v Degree >> $out/forms.md
  rm -f $out/f${f}_results.db
done
cat $out/forms.md
