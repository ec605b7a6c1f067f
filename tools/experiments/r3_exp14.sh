#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e14; mkdir -p $out
for v in 1 0 1 0; do
  python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --lib-option long_first_on_aux=$v 2>&1 >/dev/null | grep summary | sed "s/^/long_first=$v /" | cut -c1-200
done
for sc in 25 27; do for v in 1 0; do
  python bench.py --scale $sc --steps 10 --warmup 3 --cpu-scale 0 --no-extra --lib-option long_first_on_aux=$v 2>&1 >/dev/null | grep summary | sed "s/^/long_first=$v /" | cut -c1-200
done; done
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
