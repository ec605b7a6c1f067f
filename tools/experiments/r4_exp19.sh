#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e19; mkdir -p $out
timeout 900 python tools/work_shape_probe.py --scale 26 > $out/shape26.txt 2>&1
cat $out/shape26.txt | cut -c1-330
