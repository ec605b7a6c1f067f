# closing session: edge values through the column-blocked stream -- its tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp49; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_tiles.py tests/test_dropin_apps.py -x -q -m gpu -k "blocked or sweep or edge_value" > $out/pytest.txt 2>&1; tail -12 $out/pytest.txt
