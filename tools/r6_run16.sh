#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
timeout 900 python -m pytest tests/test_dropin_apps.py -x -q -m gpu > gpurun_out/r6/dropin.log 2>&1; tail -5 gpurun_out/r6/dropin.log
{
echo "# unchanged reference apps with the guided pull (default on graphs of >= 2^27 edges)"
python tools/app_at_scale.py 26 2>&1 | grep "=="
echo "# GRAPHMAT_OPTIONS=guided_pull=0"
GRAPHMAT_OPTIONS=guided_pull=0 python tools/app_at_scale.py 26 2>&1 | grep "=="
} > gpurun_out/r6/unchanged_apps_guided.txt
cut -c1-230 gpurun_out/r6/unchanged_apps_guided.txt
