#!/bin/bash
# closing session of round 4, GPU call 4: the tile count once more, now that the medium rows are swept (tiles only cut the one-wave-per-row
# and giant rows; slices = tiles * k <= 64): seeds 1-3 at scale 26, and scales 25 / 27
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
for seed in 1 2 3; do
  for t in 3 4 5 6 7 8 9; do f=$out/tc_seed${seed}_tiles$t; timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --seed $seed --col-tiles $t > $f.json 2> $f.err; echo "scale 26 seed $seed tiles $t: $(sm $f.err)"; done
done
for t in 2 3 4 5 6; do f=$out/tc_s25_tiles$t; timeout 600 python bench.py --scale 25 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --col-tiles $t > $f.json 2> $f.err; echo "scale 25 seed 1 tiles $t: $(sm $f.err)"; done
for t in 6 8 10 12 15; do f=$out/tc_s27_tiles$t; timeout 600 python bench.py --scale 27 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --col-tiles $t > $f.json 2> $f.err; echo "scale 27 seed 1 tiles $t: $(sm $f.err)"; done
for t in 1 2 3; do f=$out/tc_s24_tiles$t; timeout 600 python bench.py --scale 24 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --col-tiles $t > $f.json 2> $f.err; echo "scale 24 seed 1 tiles $t: $(sm $f.err)"; done
