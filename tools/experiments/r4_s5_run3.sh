#!/bin/bash
# closing session of round 4, GPU call 3: the automatic tile / sweep policy on other inputs, re-measured under the sources that have the
# row-stationary sweep (profiles/r04_policy_other_inputs.md was measured before it existed)
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra"
sm() { grep summary $1 | sed 's/send=.*//' ; }
tiles() { python -c "import json,sys; print(json.loads(open('$1').read().strip().splitlines()[-1])['config']['col_tiles'])"; }
for seed in 2 3; do
  f=$out/pol_seed${seed}_auto; $B --seed $seed > $f.json 2> $f.err; echo "seed $seed automatic: $(sm $f.err) col_tiles=$(tiles $f.json)"
  f=$out/pol_seed${seed}_nosweep; $B --seed $seed --lib-option sweep_slices=0 > $f.json 2> $f.err; echo "seed $seed sweep off: $(sm $f.err) col_tiles=$(tiles $f.json)"
  for t in 6 10; do f=$out/pol_seed${seed}_tiles$t; $B --seed $seed --col-tiles $t > $f.json 2> $f.err; echo "seed $seed tiles $t: $(sm $f.err)"; done
done
f=$out/pol_uniform_auto; $B --graph uniform > $f.json 2> $f.err; echo "uniform automatic: $(sm $f.err) col_tiles=$(tiles $f.json)"
f=$out/pol_uniform_nosweep; $B --graph uniform --lib-option sweep_slices=0 > $f.json 2> $f.err; echo "uniform sweep off: $(sm $f.err) col_tiles=$(tiles $f.json)"
for sc in 24 25 27; do f=$out/pol_scale$sc; timeout 900 python bench.py --scale $sc --steps 10 --warmup 3 --cpu-scale 0 --no-extra > $f.json 2> $f.err; echo "RMAT-$sc seed 1 automatic: $(sm $f.err) col_tiles=$(tiles $f.json)"; done
