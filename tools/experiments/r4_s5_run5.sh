#!/bin/bash
# closing session of round 4, GPU call 5: (1) parity of the new sweep forms (the tiles' one-wave-per-row rows folded 16 to a wave on the auxiliary
# stream); (2) more tile counts (fewer tiles win since the sweep: call 4); (3) the new forms timed at 3 and 8 tiles
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | tail -5
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(sm $out/$name.err)"; }
run tc_s26_t2 --scale 26 --col-tiles 2
for t in 3 4 5; do run tc_s27_t$t --scale 27 --col-tiles $t; done
for t in 1 2 3; do run tc_s23_t$t --scale 23 --col-tiles $t; done
for t in 1 2; do run tc_s22_t$t --scale 22 --col-tiles $t; done
for t in 2 3 8; do
  run f20_s26_t$t --scale 26 --col-tiles $t --lib-option sweep_form=20
  run f36_s26_t$t --scale 26 --col-tiles $t --lib-option sweep_form=36
done
run f20_s25_t3 --scale 25 --col-tiles 3 --lib-option sweep_form=20
run f20_s24_t2 --scale 24 --col-tiles 2 --lib-option sweep_form=20
run f20_s27_t6 --scale 27 --col-tiles 6 --lib-option sweep_form=20
