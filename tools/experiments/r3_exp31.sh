#!/bin/bash
# two-stage schedule vs programs that change in do_every_iteration; fused apply+send in the sharded plain loop
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e31; mkdir -p $out
timeout 1700 python -m pytest tests/test_gpu_multirank_apps.py tests/test_gpu_multi.py tests/test_dropin_apps.py -q -m gpu -x > $out/pytest.txt 2>&1
tail -15 $out/pytest.txt | cut -c1-250
python tools/shard_emulation.py --staged --shards 0 1 2>&1 | grep -v amdgpu | cut -c1-300
