#!/bin/bash
# closing session of round 4, GPU call 8: the DPP chain fold in the one-wave-per-row kernel (float sums) against the broadcast form
# (debug flag 32768 = DBG_READLANE_FOLD), parity first
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -x -q -m gpu -k "not fullscale" 2>&1 | tail -4
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(sm $out/$name.err) $(grep -E 'Error|rror' $out/$name.err | head -1 | cut -c1-120)"; }
for sc in 26 24 22 25 27; do
  run dpp_s$sc --scale $sc
  run readlane_s$sc --scale $sc --debug-flags 32768
done
run dpp_s26_b --scale 26
run readlane_s26_b --scale 26 --debug-flags 32768
