#!/bin/bash
# closing session of round 4: the profile set for the final sources (tools/final_profiles_r4.sh), then the default bench line
cd $GRAFT_REPO_ROOT
bash tools/final_profiles_r4.sh > gpurun_out/final_profiles.log 2>&1
tail -30 gpurun_out/final_profiles.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/final_26/bench_default.json 2> gpurun_out/final_26/bench_default.err
cut -c1-300 gpurun_out/final_26/bench_default.json
