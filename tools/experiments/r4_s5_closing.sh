#!/bin/bash
# closing session of round 4: the whole GPU suite on the final sources, then the profile set for them (tools/final_profiles_r4.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s5
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s5/pytest_gpu_final.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s5/pytest_gpu_final.txt ); tail -3 gpurun_out/s5/pytest_gpu_final.txt
bash tools/final_profiles_r4.sh > gpurun_out/final_profiles.log 2>&1
tail -4 gpurun_out/final_profiles.log | cut -c1-200
