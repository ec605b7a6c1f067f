#!/bin/bash
# round 5, experiment 23: whole GPU suite, smoke, bench at five scales on the current sources
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e23; mkdir -p $out
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s25 --scale 25
run s24 --scale 24
run s27 --scale 27
run s22 --scale 22
