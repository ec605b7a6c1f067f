#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e14; mkdir -p $out
timeout 300 build/sgather_bench > $out/sgather.txt 2>&1
cat $out/sgather.txt
