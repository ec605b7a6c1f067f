#!/bin/bash
# round 6: the PageRank line at every scale of the README (bench.py --steps 10 --warmup 2 --cpu-scale 0 --no-extra)
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
for sc in 20 22 23 24 25 26 27; do bash tools/sweep.sh $sc "--no-extra"; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/scales.txt
grep "==\|summary" gpurun_out/r6/scales.txt | sed 's/\[bench\] summary //' | cut -c1-140
