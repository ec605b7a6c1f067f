#!/bin/bash
# the non-gather part ("skeleton") of the TILED multiply at RMAT-26: ablation build, gathers and / or folds skipped
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e16; mkdir -p $out
export GRAPHMAT_HIP_LIBRARY=$R/build/ablation/libgraphmat_hip.so
B="timeout 600 python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra"
for f in 0 1 2 3; do
  $B --debug-flags $f > $out/flags_$f.json 2> $out/flags_$f.err
  echo "flags $f: $(grep summary $out/flags_$f.err | cut -c1-160)"
done
for t in 1 4 12; do
  $B --debug-flags 3 --col-tiles $t > $out/flags_3_tiles_$t.json 2> $out/flags_3_tiles_$t.err
  echo "flags 3 tiles $t: $(grep summary $out/flags_3_tiles_$t.err | cut -c1-160)"
done
for f in 0 3; do
  rocprofv3 --kernel-trace --stats -d $out/prof_$f -o p -- $B --debug-flags $f --no-timing > $out/prof_$f.log 2>&1
  python - $out/prof_$f <<'PY'
import sys, glob, csv
d = sys.argv[1]
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:12]:
        print('%-60s calls %5s total %9.3f ms avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
done
