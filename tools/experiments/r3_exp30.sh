#!/bin/bash
# a shard's long wave rows on the auxiliary stream or on the main one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in 1 0 1 0; do
  echo "shard_long_rows_on_aux=$v"; python tools/shard_emulation.py --staged --shards 0 --lib-option shard_long_rows_on_aux=$v 2>&1 | grep -v amdgpu | cut -c1-300
done
python tools/shard_emulation.py --staged --shards 1 7 2>&1 | grep -v amdgpu | cut -c1-300
