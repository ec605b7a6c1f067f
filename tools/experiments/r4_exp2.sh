#!/bin/bash
# round 4, experiment 2: GPU test suite after the round's first changes; unchanged apps by default; the tiled multiply with the
# gathers of each tile's cold columns removed (ablation build); default bench with the new cpu_baseline probe
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e2; mkdir -p $out
B="timeout 600 python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --no-extra"
ms() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1].split('/')[-1], d['ms_per_step'], 'rowblock', r.get('rowblock_avg_ms'), 'wave', r.get('wave_avg_ms'), 'aux', r.get('aux_streams_avg_ms_overlapped'), 'apply', r.get('apply_avg_ms'))" $1; }
(
export GRAPHMAT_HIP_LIBRARY=$R/build/ablation/libgraphmat_hip.so
for cfg in "524288 65536" "262144 32768" "1048576 131072" "524288 0" "0 65536" "131072 16384"; do
  set -- $cfg
  f=$out/tiled_cold_$1_$2
  $B --lib-option ablate_cold_from=$1 --lib-option ablate_cold_short=$2 > $f.json 2> $f.err; ms $f.json; grep "ablation:" $f.json $f.err | cut -c1-200 | head -12
done
for T in 12 16; do
  f=$out/tiles${T}_cold
  $B --col-tiles $T --lib-option ablate_cold_from=262144 --lib-option ablate_cold_short=32768 > $f.json 2> $f.err; ms $f.json
done
)
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -15 $out/pytest_gpu.txt
{
echo "# unchanged reference apps (build/ref_apps) on RMAT-22: exact-by-default (ordered fold: no trait, no probe) vs GRAPHMAT_TRUST_PROBE=1"
python tools/app_at_scale.py 22 2>&1 | grep "=="
echo "# GRAPHMAT_TRUST_PROBE=1"
GRAPHMAT_TRUST_PROBE=1 python tools/app_at_scale.py 22 2>&1 | grep "=="
} > $out/r04_unchanged_apps.txt
cat $out/r04_unchanged_apps.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -30 $out/bench_default.err | cut -c1-260
