#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e33; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
for f in 6 5; do
rocprofv3 --kernel-trace --output-format csv --stats -d $out/prof$f -o p -- $B --no-timing --lib-option sweep_form=$f > $out/prof$f.log 2>&1
echo "form $f"
python - $out/prof$f <<'PY'
import sys, glob, csv
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r['TotalDurationNs']))
        for r in rows[:24]:
            if 'PageRank' in r['Name']:
                print('  %-56s calls %5s total %9.3f ms avg %8.1f us' % (r['Name'][:56], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
    # timeline of the last iteration
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'PageRank' in r['Kernel_Name']]
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        last = [i for i, r in enumerate(rows) if 'k_spmv_sweep' in r['Kernel_Name']][-2]
        t0 = int(rows[last - 1]['Start_Timestamp'])
        for r in rows[last - 1:last + 45]:
            print('    %8.1f %8.1f  %s q%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][19:50], r.get('Queue_Id', '')))
PY
done
