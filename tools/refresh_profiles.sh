#!/bin/bash
# One gpurun call that leaves a consistent set of round profiles: the rocprofv3 passes (tools/final_profiles.sh), the
# traffic table rebuilt from them on the box, then the default bench line quoting that table.
#   usage (through gpurun): bash tools/refresh_profiles.sh <tag>
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; cd $R
bash tools/final_profiles.sh $tag 26 22
for sc in 26 22; do cp gpurun_out/final_$sc/${tag}_scale${sc}_*.md profiles/; cp gpurun_out/final_$sc/bench.json profiles/${tag}_scale${sc}_bench.json; done
python tools/pmc_to_json.py $tag 26 22 > /dev/null
mkdir -p gpurun_out/final_default
cp profiles/pmc_traffic.json gpurun_out/final_default/pmc_traffic.json
python bench.py > gpurun_out/final_default/bench.json 2> gpurun_out/final_default/bench.err
