#!/bin/bash
# column tiles with row classes fixed per row (auxiliary stream joined once per iteration): parity with the mode forced, then a sweep
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e33; mkdir -p $out
GRAPHMAT_OWN_WAVE_ROW=512 timeout 1700 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py tests/test_dropin_apps.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -6 $out/pytest.txt | cut -c1-200
for v in 0 1024 2048 4096 8192; do
  echo "own_wave_row=$v $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option own_wave_row=$v 2>&1 | grep summary | cut -c1-170)"
done
