#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e20; mkdir -p $out
export GRAPHMAT_HIP_LIBRARY=$R/build/ablation/libgraphmat_hip.so
echo "== product schedule (two streams), streaming wave16" > $out/times.txt
timeout 600 python tools/wave_times_probe.py --scale 26 >> $out/times.txt 2>&1
echo "== group-by-group lean wave16 (debug_flags 16384)" >> $out/times.txt
timeout 600 python tools/wave_times_probe.py --scale 26 --lib-option debug_flags=16384 >> $out/times.txt 2>&1
echo "== everything on one stream (debug_flags 16 = no overlap)" >> $out/times.txt
timeout 600 python tools/wave_times_probe.py --scale 26 --lib-option debug_flags=16 >> $out/times.txt 2>&1
grep -v amdgpu.ids $out/times.txt
