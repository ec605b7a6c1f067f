#!/bin/bash
# round 3, first GPU call: new parity tests, persistent wave16 sweep, TCP/TA counter passes of the round-2 kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r3e1; mkdir -p $out
python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "persistent_wave16" > $out/pytest_forms.txt 2>&1
tail -3 $out/pytest_forms.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config2 or tiled_equals_untiled" > $out/pytest_parity.txt 2>&1
tail -3 $out/pytest_parity.txt
for t in 6 8 10; do
  for f in 0 1 2 3 4 5; do
    echo "== tiles $t form $f" >> $out/sweep.txt
    python bench.py --scale 26 --steps 10 --warmup 2 --cpu-scale 0 --no-extra --col-tiles $t --lib-option wave16_form=$f 2>&1 >/dev/null | grep summary >> $out/sweep.txt
  done
done
cat $out/sweep.txt
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC)_[A-Z0-9_]+" | sort -u > $out/counters_tcp_ta.txt
wc -l $out/counters_tcp_ta.txt
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out -o pmc$i -- python bench.py --scale 26 --steps 5 --warmup 1 --cpu-scale 0 --no-timing --no-extra > /dev/null 2> $out/pmc$i.err
  python tools/prof_summary.py $out/pmc${i}_results.db | grep -E "counter|k_spmv|k_giant" | grep -v Degree > $out/pmc$i.md
  rm -f $out/pmc${i}_results.db
done
cat $out/pmc*.md | grep -v "^| kernel" | head -80
