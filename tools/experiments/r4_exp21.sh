#!/bin/bash
# tile lists sorted by piece length: product timing A/B (sorted / row order) x (streaming / group-by-group wave16), wave time stamps, parity
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e21; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 20 --warmup 3 --cpu-scale 0 --no-extra"
run() { name=$1; shift; $B "$@" > $out/$name.json 2> $out/$name.err; echo "$name: $(grep summary $out/$name.err | cut -c1-150)"; }
run sorted_stream
run sorted_groups --debug-flags 16384
run roworder_stream --lib-option sort_tile_lists=0
run roworder_groups --lib-option sort_tile_lists=0 --debug-flags 16384
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | tail -n 3
export GRAPHMAT_HIP_LIBRARY=$R/build/ablation/libgraphmat_hip.so
timeout 600 python tools/wave_times_probe.py --scale 26 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/wave_times_probe.py --scale 26 --lib-option debug_flags=16384 2>&1 | grep -v amdgpu.ids
