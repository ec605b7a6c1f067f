#!/bin/bash
# column tiles on shards: multi-rank parity (forced tiles), then what a shard of 8 costs with 1 / 3 / 4 / 6 tiles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e21; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multirank_apps.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -25 $out/pytest.txt | cut -c1-250
python tools/shard_emulation.py --staged --shards 0 --col-tiles 1 3 4 6 2>&1 | grep -v amdgpu > $out/shards.txt; cat $out/shards.txt | cut -c1-330
