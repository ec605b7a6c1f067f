#!/bin/bash
# round 4, experiment 3: full GPU suite; where the unchanged PageRank.cpp spends its time; shard emulation baseline; policy on
# other inputs; SGD counters and the matrix-core form of its dot products
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e3; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -6 $out/pytest_gpu.txt
# unchanged PageRank.cpp on RMAT-22 under the kernel trace
python tools/app_at_scale.py 22 2>&1 | grep "==" > $out/apps22.txt; cat $out/apps22.txt
rocprofv3 --kernel-trace --stats -d $out -o prapp -- build/ref_apps/PageRank /tmp/rmat22.bin.mtx > $out/prapp.out 2> $out/prapp.err
python tools/prof_summary.py $out/prapp_results.db | head -16 | cut -c1-200
GRAPHMAT_VERBOSE=1 build/ref_apps/PageRank /tmp/rmat22.bin.mtx 2>&1 | grep -E "reduce strategy|iteration 5[0-3]|loop done|PR Time" | head
rm -f $out/*.db
# shards of 8
timeout 900 python tools/shard_emulation.py --staged --shards 0 1 > $out/shard_emulation.txt 2>&1; cat $out/shard_emulation.txt | cut -c1-330
# policy on other inputs
B="timeout 600 python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra"
sm() { grep summary $1 | sed 's/rowblock.*//' ; }
for seed in 2 3; do
  for t in -1 6 10; do f=$out/rmat_seed${seed}_tiles$t; $B --seed $seed --col-tiles $t > $f.json 2> $f.err; echo "seed $seed tiles $t: $(sm $f.err)"; done
done
for t in -1 1 4 8 16; do f=$out/uniform_tiles$t; $B --graph uniform --col-tiles $t > $f.json 2> $f.err; echo "uniform tiles $t: $(sm $f.err) col_tiles=$(python -c "import json,sys; print(json.loads(open('$f.json').read().strip().splitlines()[-1])['config']['col_tiles'])")"; done
f=$out/uniform_plainforms; $B --graph uniform --lib-option wave16_form=0 --lib-option rowwave_form=0 > $f.json 2> $f.err; echo "uniform plain kernel forms: $(sm $f.err)"
f=$out/uniform_native; $B --graph uniform --native-layout > $f.json 2> $f.err; echo "uniform native layout: $(sm $f.err)"
# SGD: the vector form and the matrix-core form of the dot products
python tools/sgd_bench.py --users 2000000 --items 200000 --compare-mfma 2>&1 | grep "^SGD" | tee $out/sgd.txt
python tools/sgd_bench.py --users 2000000 --items 200000 --lib-option sgd_mfma=1 2>&1 | grep "^SGD" | tee -a $out/sgd.txt
python tools/sgd_bench.py --users 10000000 --items 1000000 --iters 3 2>&1 | grep "^SGD" | tee -a $out/sgd.txt
python tools/sgd_bench.py --users 10000000 --items 1000000 --iters 3 --lib-option sgd_mfma=1 2>&1 | grep "^SGD" | tee -a $out/sgd.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $out -o sgdpmc_$n -- python tools/sgd_bench.py --users 10000000 --items 1000000 --iters 1 > /dev/null 2> $out/sgdpmc_$n.err
  python tools/prof_summary.py $out/sgdpmc_${n}_results.db 2>/dev/null | grep -E "counter|k_sgd_multiply" > $out/r04_sgd_k128_pmc_$n.md; cat $out/r04_sgd_k128_pmc_$n.md | cut -c1-160
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $out -o sgdmfma -- python tools/sgd_bench.py --users 2000000 --items 200000 --iters 2 --lib-option sgd_mfma=1 > /dev/null 2> $out/sgdmfma.err
python tools/prof_summary.py $out/sgdmfma_results.db 2>/dev/null | grep -E "counter|k_sgd_multiply" > $out/r04_sgd_k128_mfma_pmc.md; cat $out/r04_sgd_k128_mfma_pmc.md | cut -c1-160
rm -f $out/*.db
