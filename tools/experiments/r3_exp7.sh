#!/bin/bash
# rowwave forms: parity + sweep; giant maps test; BFS with bits_push
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e7; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "giant_rows or long_row or rmat or config2" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
for f in 0 1 2 3 4; do
  python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra --lib-option rowwave_form=$f 2>&1 >/dev/null | grep summary | sed "s/^/rowwave=$f /" | cut -c1-200
done
for f in 1 4; do
  python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra --lib-option rowwave_form=$f --lib-option wave16_form=2 2>&1 >/dev/null | grep summary | sed "s/^/rowwave=$f wave16=2 /" | cut -c1-200
  python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra --col-tiles 8 --lib-option rowwave_form=$f 2>&1 >/dev/null | grep summary | sed "s/^/rowwave=$f tiles=8 /" | cut -c1-200
done
python tools/bfs_bench.py --scale 26 2>&1 | grep "^BFS" | cut -c1-200
