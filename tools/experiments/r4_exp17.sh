#!/bin/bash
# the multiply with EVERY gather replaced by an LDS read (ablation build, ablate_cold_from=1 / ablate_cold_short=1): the part of
# the iteration that is not the vector-memory path, by tile count
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e17; mkdir -p $out
export GRAPHMAT_HIP_LIBRARY=$R/build/ablation/libgraphmat_hip.so
B="timeout 600 python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra"
for t in 8 1 4 12; do
  $B --col-tiles $t --lib-option ablate_cold_from=1 --lib-option ablate_cold_short=1 > $out/allcold_$t.json 2> $out/allcold_$t.err
  echo "all gathers from LDS, $t tiles: $(grep summary $out/allcold_$t.err | cut -c1-170)"
done
$B --lib-option ablate_cold_from=1 --lib-option ablate_cold_short=1 --debug-flags 1 > $out/allcold_nofold.json 2> $out/allcold_nofold.err
echo "all gathers from LDS, no folds, 8 tiles: $(grep summary $out/allcold_nofold.err | cut -c1-170)"
rocprofv3 --kernel-trace --output-format csv --stats -d $out/prof -o p -- $B --lib-option ablate_cold_from=1 --lib-option ablate_cold_short=1 --no-timing > $out/prof.log 2>&1
python - $out/prof <<'PY'
import sys, glob, csv
d = sys.argv[1]
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:16]:
        if 'spmv' in r['Name'] or 'giant' in r['Name'] or 'apply' in r['Name']:
            print('%-60s calls %5s total %9.3f ms avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
