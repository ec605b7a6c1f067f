#!/bin/bash
# round 4, experiment 7: kernel timeline of a shard of 8 under the two-stage schedule and the plain loop (do-nothing exchange)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e7; mkdir -p $out
rocprofv3 --kernel-trace -d $out -o shard -- python tools/shard_emulation.py --staged --shards 0 --iters 6 > $out/shard.txt 2> $out/shard.err
cat $out/shard.txt | cut -c1-250
python tools/prof_timeline.py $out/shard_results.db --last 400 > $out/timeline_all.md
rm -f $out/*.db
python - <<'PY'
import re
rows=[l for l in open('/root/repo/gpurun_out/r4e7/timeline_all.md') if l.startswith('| ') and not l.startswith('| start') and not l.startswith('|---')]
print(len(rows))
for l in rows[-400:]:
    c=[x.strip() for x in l.strip().strip('|').split('|')]
    name=re.sub(r"<.*","",c[3].strip('`'))
    print("%10.1f %8.1f %8.1f %s"%(float(c[0]),float(c[1]),float(c[2]),name))
PY
