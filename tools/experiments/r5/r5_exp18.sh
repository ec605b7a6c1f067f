#!/bin/bash
# round 5, experiment 18: does the automatic policy (slices, border, sweep) hold off RMAT seed 1?  + the default bench line with extras
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e18; mkdir -p $out
B="timeout 900 python bench.py --steps 10 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-150)"; }
run seed1 --scale 26
run seed2 --scale 26 --seed 2
run seed2_t64 --scale 26 --seed 2 --lib-option sweep_slices=64
run seed2_nosweep --scale 26 --seed 2 --lib-option sweep_slices=0
run seed3 --scale 26 --seed 3
run seed3_l2048 --scale 26 --seed 3 --lib-option sweep_long_row=2048
run uniform --scale 26 --graph uniform
run uniform_nosweep --scale 26 --graph uniform --lib-option sweep_slices=0
timeout 1500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; grep -E "summary|cpu_baseline:|extra" $out/bench_default.err | cut -c1-220
