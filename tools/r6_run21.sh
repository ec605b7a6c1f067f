#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "sparse_message or 768 or sweep" > gpurun_out/r6/sparse_sweep_tests.log 2>&1; tail -15 gpurun_out/r6/sparse_sweep_tests.log
timeout 900 python -m pytest tests/test_dropin_apps.py -x -q -m gpu -k "untraited" > gpurun_out/r6/untraited.log 2>&1; tail -5 gpurun_out/r6/untraited.log
{
echo "# unchanged reference apps, sparse x through the sweep (default)"
python tools/app_at_scale.py 26 2>&1 | grep "=="
echo "# GRAPHMAT_OPTIONS=sweep_form=32 (sparse x keeps the tile-less pull kernels)"
GRAPHMAT_OPTIONS=sweep_form=32 python tools/app_at_scale.py 26 2>&1 | grep "SSSP"
python tools/app_at_scale.py 24 2>&1 | grep "SSSP"
GRAPHMAT_OPTIONS=sweep_form=32 python tools/app_at_scale.py 24 2>&1 | grep "SSSP"
} > gpurun_out/r6/unchanged_apps_sparse_sweep.txt
cut -c1-230 gpurun_out/r6/unchanged_apps_sparse_sweep.txt
