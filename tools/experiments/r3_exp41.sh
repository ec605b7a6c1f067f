#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e41; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
for sc in 25 27; do
  for v in 1 0; do
    echo "scale=$sc untiled_pass_plain=$v $(python bench.py --scale $sc --steps 10 --warmup 3 --no-extra --cpu-scale 0 --lib-option untiled_pass_plain=$v 2>&1 | grep summary | cut -c40-150)"
  done
done
