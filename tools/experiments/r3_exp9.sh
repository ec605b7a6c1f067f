#!/bin/bash
# every row tiled (tile_min_row 0) with the wave-level row-block kernel; tile counts
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e9; mkdir -p $out
for t in 2 3 4 6; do
  python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra --col-tiles $t --tile-min-row 0 2>&1 >/dev/null | grep summary | sed "s/^/alltiled tiles=$t /" | cut -c1-170
done 2>&1 | tee $out/alltiled.txt
for t in 5 6 7; do
  python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra --col-tiles $t 2>&1 >/dev/null | grep summary | sed "s/^/tiles=$t /" | cut -c1-170
done 2>&1 | tee -a $out/alltiled.txt
for pc in 1 2; do
  python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra --lib-option rowwave_form=3 --lib-option persist_per_cu=$pc 2>&1 >/dev/null | grep summary | sed "s/^/rowwave=3 per_cu=$pc /" | cut -c1-170
done 2>&1 | tee -a $out/alltiled.txt
timeout 1200 python -m pytest tests/test_dropin_apps.py tests/test_gpu_multirank_apps.py tests/test_gpu_multi.py tests/test_gpu_tiles.py -q -m gpu > $out/pytest.txt 2>&1
tail -6 $out/pytest.txt
