#!/bin/bash
# round 5, experiment 24: rest of the GPU suite; part of the hot set requested before the slice barrier
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e24; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 900 build/sweep_lib_bench 26 4 > $out/t26_base.txt 2>&1; grep "library\|giant rows as" $out/t26_base.txt
timeout 900 build/sweep_lib_bench_h 26 4 > $out/t26_hreg12.txt 2>&1; grep "library\|giant rows as\|differ" $out/t26_hreg12.txt; grep "waves:" $out/t26_hreg12.txt | cut -c1-250
timeout 900 build/sweep_lib_bench_h 25 4 > $out/t25_hreg12.txt 2>&1; grep "library\|giant rows as" $out/t25_hreg12.txt
timeout 900 build/sweep_lib_bench 25 4 > $out/t25_base.txt 2>&1; grep "library\|giant rows as" $out/t25_base.txt
timeout 2700 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_tiles.py 2>&1 | tail -n 6
