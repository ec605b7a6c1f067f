#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e12; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_dropin_apps.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
for v in 3 2; do
  python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --lib-option short_streams=$v 2>&1 >/dev/null | grep summary | sed "s/^/short_streams=$v /" | cut -c1-200
done
for t in 8 10; do
  python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra --col-tiles $t 2>&1 >/dev/null | grep summary | sed "s/^/tiles=$t /" | cut -c1-200
done
python bench.py --scale 27 --steps 10 --warmup 3 --cpu-scale 0 --no-extra 2>&1 >/dev/null | grep summary | cut -c1-200
python bench.py --scale 25 --steps 10 --warmup 3 --cpu-scale 0 --no-extra 2>&1 >/dev/null | grep summary | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fullscale_pagerank" > $out/pytest2.txt 2>&1
tail -3 $out/pytest2.txt
