#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e34; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
for f in 4 5 6 10 2 9 1; do run form$f --lib-option sweep_form=$f; done
timeout 600 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 3
