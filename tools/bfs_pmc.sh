#!/bin/bash
# vector-memory-path counters of the BFS kernels, per dispatch (the bottom-up levels of tools/bfs_bench.py at RMAT-26)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/bfspmc; mkdir -p $out
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out -o pmc$i -- python tools/bfs_bench.py --scale 26 > /dev/null 2> $out/pmc$i.err
  for k in k_spmv_short_last k_spmv_wave_grouped k_apply; do python tools/pmc_per_dispatch.py $out/pmc${i}_results.db $k 100 | tail -5; done > $out/pmc$i.txt
  rm -f $out/pmc${i}_results.db
done
cat $out/pmc*.txt
