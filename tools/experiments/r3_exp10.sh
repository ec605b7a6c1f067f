#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e10; mkdir -p $out
python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra 2>&1 >/dev/null | grep summary | cut -c1-200
python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra --lib-option rowwave_form=0 2>&1 >/dev/null | grep summary | sed "s/^/rowwave=0 /" | cut -c1-200
python bench.py --scale 27 --steps 10 --warmup 3 --cpu-scale 0 --no-extra 2>&1 >/dev/null | grep summary | cut -c1-200
timeout 1700 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -8 $out/pytest_gpu.txt
