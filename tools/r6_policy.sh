#!/bin/bash
# round 6: does the automatic layout policy (tiles, slices, medium / long border) hold off the metric's input?  RMAT seeds 2 and 3, the
# uniform graph, and RMAT with SCRAMBLED vertex ids (degree rank and native ranges decorrelated)
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
bash tools/sweep.sh 26 "--no-extra" "--no-extra --seed 2" "--no-extra --seed 3" \
  "--no-extra --graph rmat-scrambled" "--no-extra --graph rmat-scrambled --lib-option sweep_slices=64" "--no-extra --graph rmat-scrambled --lib-option sweep_slices=128" \
  "--no-extra --graph rmat-scrambled --col-tiles 2" "--no-extra --graph rmat-scrambled --col-tiles 5" "--no-extra --graph rmat-scrambled --lib-option sweep_slices=0" \
  "--no-extra --graph uniform"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/policy_other_inputs.txt
cat gpurun_out/r6/policy_other_inputs.txt
