#!/bin/bash
# third stream for the giant rows' passes (engine option giant_stream) x the row length from which a row is a one-wave-per-row
# row in every tile (own_wave_row: shifts work from the main stream's 16-rows-per-wave kernel to the auxiliary stream)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/r4e15; mkdir -p $out
run() { # name, options...
  name=$1; shift
  opts=""; for o in "$@"; do opts="$opts --lib-option $o"; done
  timeout 600 python bench.py --cpu-scale 0 --no-extra --steps 20 --warmup 5 $opts > $out/$name.json 2> $out/$name.err
  echo "$name: $(grep -E 'summary' $out/$name.err | cut -c1-330)"
}
run gs1_ow2048 giant_stream=1 own_wave_row=2048
run gs1_ow1024 giant_stream=1 own_wave_row=1024
run gs1_ow512 giant_stream=1 own_wave_row=512
run gs1_ow256 giant_stream=1 own_wave_row=256
run gs0_ow2048 own_wave_row=2048
run gs0_ow1024 own_wave_row=1024
run gs1_ow1024_w0 giant_stream=1 own_wave_row=1024 wave16_form=0
run gs1_ow512_w0r0 giant_stream=1 own_wave_row=512 wave16_form=0 rowwave_form=0
