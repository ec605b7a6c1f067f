#!/bin/bash
# kernel changes of the late round: product timing, parity, A/B against the previous form, kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e18; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
$B > $out/prod.json 2> $out/prod.err
echo "product: $(grep summary $out/prod.err | cut -c1-170)"
$B --debug-flags 16384 > $out/prod_nostream.json 2> $out/prod_nostream.err
echo "product, group-by-group wave16: $(grep summary $out/prod_nostream.err | cut -c1-170)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 3
rocprofv3 --kernel-trace --output-format csv --stats -d $out/prof_prod -o p -- $B --no-timing > $out/prof_prod.log 2>&1
python - $out/prof_prod <<'PY'
import sys, glob, csv
for d in sys.argv[1:]:
    print(d)
    for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r['TotalDurationNs']))
        for r in rows[:20]:
            if 'spmv' in r['Name'] or 'giant' in r['Name'] or 'apply' in r['Name']:
                print('  %-56s calls %5s total %9.3f ms avg %8.1f us' % (r['Name'][:56], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
