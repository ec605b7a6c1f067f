#!/bin/bash
# round 5, experiment 7: the sweep's stream describes itself (meta rows), staging rounds issue all their loads at once
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e7; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sweep or structure" 2>&1 | tail -n 5
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s26 --scale 26
run s26b --scale 26
run s25 --scale 25
run s24 --scale 24
run s27 --scale 27
