#!/bin/bash
# round 5, experiment 15: the whole GPU suite on the current sources + bench at four scales
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e15; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 15
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s25 --scale 25
run s24 --scale 24
run s27 --scale 27
run s22 --scale 22
