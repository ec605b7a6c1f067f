#!/bin/bash
# giant piece maps (history hints): parity tests, single-GPU bench with/without, shard emulation with/without
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e6; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py tests/test_gpu_multi.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -8 $out/pytest.txt
for gm in 1 0; do
  python bench.py --scale 26 --steps 20 --warmup 5 --cpu-scale 0 --no-extra --lib-option giant_maps=$gm 2>&1 >/dev/null | grep summary | sed "s/^/gm=$gm /"
done
python tools/shard_emulation.py --staged --shards 0 1 > $out/shards_gm1.txt 2>&1; cat $out/shards_gm1.txt
