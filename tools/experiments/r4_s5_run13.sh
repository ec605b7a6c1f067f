#!/bin/bash
# closing session of round 4, GPU call 13: unequal tiles -- the last tile (whose passes run beside the sweep) gets a larger share of the slices
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(sm $out/$name.err) $(grep -E 'Error|rror' $out/$name.err | head -1 | cut -c1-120)"; }
run base $B
for pm in 333 420 500 580 660 250; do GRAPHMAT_LAST_TILE_PERMILLE=$pm run last$pm $B; done
for pm in 400 550; do GRAPHMAT_LAST_TILE_PERMILLE=$pm run t4_last$pm $B --col-tiles 4; done
GRAPHMAT_LAST_TILE_PERMILLE=600 run t2_last600 $B --col-tiles 2
GRAPHMAT_LAST_TILE_PERMILLE=500 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiles or tiled" 2>&1 | tail -2
