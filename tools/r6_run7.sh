#!/bin/bash
# round 6: single-GPU baseline on these sources + the sweep's fold share (the waves that also fold long rows)
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
bash tools/sweep.sh 26 "--no-extra" "--no-extra --lib-option sweep_fold_share=70" "--no-extra --lib-option sweep_fold_share=85" "--no-extra --lib-option sweep_fold_share=100" "--no-extra --lib-option sweep_border_factor=2" "--no-extra --lib-option sweep_border_factor=8"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/single_fold_share.txt
cat gpurun_out/r6/single_fold_share.txt
