#!/bin/bash
# round 5, experiment 27: whole GPU suite after the engine clean-up; unchanged apps with the ordered chain beside the whole multiply
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e27; mkdir -p $out
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 5
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
{
echo "# unchanged reference apps (build/ref_apps), exact by default (no trait, no probe, no environment variables)"
python tools/app_at_scale.py 22 2>&1 | grep "=="
python tools/app_at_scale.py 26 2>&1 | grep "=="
echo "# GRAPHMAT_TRUST_PROBE=1"
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 22 2>&1 | grep "=="
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 26 2>&1 | grep "=="
} > $out/r05_unchanged_apps.txt; cat $out/r05_unchanged_apps.txt | cut -c1-230
