#!/bin/bash
# closing session of round 4, GPU call 6: the new automatic tile counts (3 at RMAT-26) -- what they pick at every scale, the secondary knobs
# re-measured around them, then the whole GPU suite (its full-scale test compares RMAT-26 automatic tiles with the untiled bits)
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
tiles() { python -c "import json,sys; print(json.loads(open('$1').read().strip().splitlines()[-1])['config']['col_tiles'])"; }
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(sm $out/$name.err) col_tiles=$(tiles $out/$name.json)"; }
for sc in 22 23 24 25 26 27; do run auto_s$sc --scale $sc; done
run auto_s26_seed2 --scale 26 --seed 2
for f in 0 1 5 6 8; do run k_form$f --scale 26 --lib-option sweep_form=$f; done
for gs in 0 2; do run k_gs$gs --scale 26 --lib-option giant_stream=$gs; done
for o in 2048 8192 16384; do run k_own$o --scale 26 --lib-option own_wave_row=$o; done
run k_plain0 --scale 26 --lib-option untiled_pass_plain=0
run k_short96 --scale 26 --short-row 96
run k_s24_form0 --scale 24 --lib-option sweep_form=0
run k_s24_form8 --scale 24 --lib-option sweep_form=8
run k_s24_gs0 --scale 24 --lib-option giant_stream=0
( timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu6.txt 2>&1; echo "pytest rc $?" >> $out/pytest_gpu6.txt ); tail -4 $out/pytest_gpu6.txt
