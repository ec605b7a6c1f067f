#!/bin/bash
# closing session of round 4, GPU call 9: row-block records (gm_csr_t.blk_desc: one 16-byte load instead of blk_seg -> seg_row -> rowptr) --
# parity subset, then the iteration at RMAT-22..27 and a shard of 8 (before: 0.44 / 0.82 / 1.33 / 2.49 / 4.84 / 11.15 ms; shard 0 plain loop 1.087)
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py tests/test_dropin_apps.py -x -q -m gpu -k "not fullscale" 2>&1 | tail -4
sm() { grep summary $1 | sed 's/send=.*//' | sed 's/.*ms.step/ms\/step/'; }
B="timeout 600 python bench.py --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(sm $out/$name.err) $(grep -E 'Error|rror' $out/$name.err | head -1 | cut -c1-120)"; }
for sc in 26 22 23 24 25 27 26; do run desc_s$sc --scale $sc; done
timeout 900 python tools/shard_emulation.py --staged --shards 0 2>&1 | cut -c1-330
