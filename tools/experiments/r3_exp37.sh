#!/bin/bash
# a shard's one-wave-per-row rows split between the streams
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in 0 100 250 400 600; do
  echo "shard_long_split=$v"; python tools/shard_emulation.py --staged --shards 0 --lib-option shard_long_split=$v 2>&1 | grep -v amdgpu | cut -c1-300
done
