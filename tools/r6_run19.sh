#!/bin/bash
mkdir -p gpurun_out/r6
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r6/build.log 2>&1
{
bash tools/sweep.sh 26 "--no-extra" "--no-extra --lib-option sweep_form=16" "--no-extra" "--no-extra --lib-option sweep_form=16"
bash tools/sweep.sh 25 "--no-extra" "--no-extra --lib-option sweep_form=16"
bash tools/sweep.sh 24 "--no-extra" "--no-extra --lib-option sweep_form=16"
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/giant_gather_apart.txt
grep "==\|summary" gpurun_out/r6/giant_gather_apart.txt | sed 's/\[bench\] summary //' | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k sweep 2>&1 | tail -2
