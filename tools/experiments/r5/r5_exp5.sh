#!/bin/bash
# round 5, experiment 5: parity suites on the SELL sweep + kernel timeline of an RMAT-26 iteration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r5e5; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu 2>&1 | tail -n 12
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 12
rocprofv3 --kernel-trace -d $out -o it -- python bench.py --scale 26 --steps 3 --warmup 1 --cpu-scale 0 --no-extra --no-timing > /dev/null 2> $out/it.err
python tools/prof_timeline.py $out/it_results.db --match "k_spmv|k_giant|k_apply|k_send" --last 16 > $out/timeline.md
rm -f $out/*.db
cat $out/timeline.md
