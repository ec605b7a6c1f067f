#!/bin/bash
# double-buffered wave16 groups: parity first, then the default bench line three times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e16; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
for i in 1 2 3; do
  python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 2> $out/b$i.err > $out/b$i.json
  grep summary $out/b$i.err | cut -c1-260
done
python bench.py --scale 25 --steps 20 --warmup 5 --no-extra --cpu-scale 0 2>&1 | grep summary | cut -c1-200
python bench.py --scale 22 --steps 20 --warmup 5 --no-extra --cpu-scale 0 2>&1 | grep summary | cut -c1-200
