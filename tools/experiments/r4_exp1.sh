#!/bin/bash
# round 4, experiment 1: what would a hot/cold split of the multiply buy?  (a) microbenchmarks of the two new passes,
# (b) the multiply with the gathers of cold columns removed (ablation build), untiled, for several hot-set limits
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e1; mkdir -p $out
timeout 600 ./build/coldpass_bench > $out/coldpass.txt 2>&1
cat $out/coldpass.txt
B="timeout 600 python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra"
ms() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['ms_per_step'], 'rowblock', r.get('rowblock_avg_ms'), 'wave', r.get('wave_avg_ms'), 'aux', r.get('aux_streams_avg_ms_overlapped'), 'apply', r.get('apply_avg_ms'))" $1; }
$B > $out/base.json 2> $out/base.err; ms $out/base.json
export GRAPHMAT_HIP_LIBRARY=$R/build/ablation/libgraphmat_hip.so
$B --col-tiles 1 > $out/untiled.json 2> $out/untiled.err; ms $out/untiled.json
for N in 262144 524288 1048576 2097152 4194304; do
  $B --col-tiles 1 --lib-option ablate_cold_from=$N > $out/cold_$N.json 2> $out/cold_$N.err; ms $out/cold_$N.json
done
$B --col-tiles 1 --lib-option ablate_cold_from=1048576 --lib-option wave16_form=0 --lib-option rowwave_form=0 > $out/cold_plain.json 2> $out/cold_plain.err; ms $out/cold_plain.json
$B --col-tiles 1 --lib-option ablate_cold_from=1048576 --lib-option rowwave_form=0 > $out/cold_plainrb.json 2> $out/cold_plainrb.err; ms $out/cold_plainrb.json
