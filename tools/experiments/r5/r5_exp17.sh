#!/bin/bash
# round 5, experiment 17: does the sweep pay below 25 MiB of live messages (RMAT-22 / 23)?  unchanged apps at RMAT-22 / 26
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e17; mkdir -p $out
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run s22 --scale 22
run s22_t2 --scale 22 --col-tiles 2
run s22_t2_s8 --scale 22 --col-tiles 2 --lib-option sweep_slices=8
run s22_t2_s32 --scale 22 --col-tiles 2 --lib-option sweep_slices=32
run s23 --scale 23
run s23_t2 --scale 23 --col-tiles 2
run s23_t2_s32 --scale 23 --col-tiles 2 --lib-option sweep_slices=32
run s26 --scale 26
{
echo "# unchanged reference apps (build/ref_apps), exact by default (no trait, no probe, no environment variables)"
python tools/app_at_scale.py 22 2>&1 | grep "=="
python tools/app_at_scale.py 26 2>&1 | grep "=="
} > $out/unchanged_apps.txt; cat $out/unchanged_apps.txt | cut -c1-260
