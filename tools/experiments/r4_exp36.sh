#!/bin/bash
# class boundaries around the sweep: which rows the untiled short-row pass, the sweep and the one-wave-per-row passes get
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e36; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep -E 'summary|Error|rror' $out/$name.err | cut -c1-150 | head -2)"; }
run base
run plain0 --lib-option untiled_pass_plain=0
run own8192 --lib-option own_wave_row=8192
run own2048 --lib-option own_wave_row=2048
run own16384 --lib-option own_wave_row=16384
run short48 --short-row 48
run short32 --short-row 32
run short96 --short-row 96
