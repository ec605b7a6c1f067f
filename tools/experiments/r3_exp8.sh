#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e8; mkdir -p $out
for sc in 22 24 25 27; do
  for f in "0 0" "2 4" "0 4" "2 0"; do
    set -- $f
    python bench.py --scale $sc --steps 20 --warmup 3 --cpu-scale 0 --no-extra --lib-option wave16_form=$1 --lib-option rowwave_form=$2 2>&1 >/dev/null | grep summary | sed "s/^/wave16=$1 rowwave=$2 /" | cut -c1-190
  done
done 2>&1 | tee $out/forms_by_scale.txt
timeout 1700 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -6 $out/pytest_gpu.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -9 $out/bench_default.err | cut -c1-400
