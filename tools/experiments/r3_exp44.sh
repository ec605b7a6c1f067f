#!/bin/bash
# keep mode: a share of the untiled pass's row-blocks on the auxiliary stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e44; mkdir -p $out
for v in 0 200 350 500 0 350; do
  echo "untiled_share=$v $(python bench.py --scale 26 --steps 20 --warmup 5 --no-extra --cpu-scale 0 --lib-option untiled_share=$v 2>&1 | grep summary | cut -c40-150)"
done
