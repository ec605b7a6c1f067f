#!/bin/bash
# one launch per tile for row-blocks + 16-row groups: parity, then A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e43; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py tests/test_dropin_apps.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
for v in 1 0 1 0; do
  echo "merge_tile_kernels=$v $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option merge_tile_kernels=$v 2>&1 | grep summary | cut -c40-150)"
done
for sc in 25 27; do for v in 1 0; do
  echo "scale=$sc merge=$v $(python bench.py --scale $sc --steps 10 --warmup 3 --no-extra --cpu-scale 0 --lib-option merge_tile_kernels=$v 2>&1 | grep summary | cut -c40-150)"
done; done
