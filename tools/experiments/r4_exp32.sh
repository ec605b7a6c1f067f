#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e32; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
for f in 4 6 10 2 8; do run form$f --lib-option sweep_form=$f; done
rocprofv3 --kernel-trace --output-format csv --stats -d $out/prof -o p -- $B --no-timing > $out/prof.log 2>&1
python - $out/prof <<'PY'
import sys, glob, csv
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r['TotalDurationNs']))
        for r in rows[:24]:
            if 'PageRank' in r['Name']:
                print('  %-56s calls %5s total %9.3f ms avg %8.1f us' % (r['Name'][:56], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
