#!/bin/bash
# the swept multiply: hot-set size (LDS left for the other streams' workgroups) x placement of the untiled pass x giant stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e25; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --lib-option sweep_slices=1"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary|Error|error|differ' $out/$name.err | cut -c1-150 | head -3)"; }
for f in 0 1 2 4 5 6; do run form$f --lib-option sweep_form=$f; done
for f in 2 6; do run form${f}_gs0 --lib-option sweep_form=$f --lib-option giant_stream=0; done
