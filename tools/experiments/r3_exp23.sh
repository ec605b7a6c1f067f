#!/bin/bash
# fused apply + send: parity (whole GPU suite), then the default bench line with and without
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e23; mkdir -p $out
timeout 1700 python -m pytest tests -q -m gpu -x > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
for f in 1 0 1 0; do
  python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option fuse_apply_send=$f 2> $out/b$f.err > $out/b$f.json
  echo "fuse=$f $(grep summary $out/b$f.err | cut -c1-200)"
done
python bench.py --scale 22 --steps 20 --warmup 5 --no-extra --cpu-scale 0 2>&1 | grep summary | cut -c1-200
python bench.py --scale 22 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option fuse_apply_send=0 2>&1 | grep summary | cut -c1-200
