#!/bin/bash
# closing session of round 4, GPU call 7: knobs fitted before the sweep existed, re-measured at the new tile counts (RMAT-26, 3 tiles)
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(sm $out/$name.err) $(grep -E 'Error|rror' $out/$name.err | head -1 | cut -c1-120)"; }
run base
run giant16384 --giant-row 16384
run giant65536 --giant-row 65536
run rankby1 --rank-by 1
run rankby2 --rank-by 2
run bal0 --lib-option tile_balance=0
run bal_pow70 --lib-option tile_balance=100070
run bal_pow130 --lib-option tile_balance=100130
run bal_add2 --lib-option tile_balance=3
run longmid2048 --lib-option long_mid=2048
run t3_own6144 --lib-option own_wave_row=6144
