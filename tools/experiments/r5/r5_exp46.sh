# closing session: k_spmv_blocked on the uniform 2^26 graph with fewer, larger slices (and the window / batch forms at 64 slices)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp46; mkdir -p $out
for sl in 32 48 56 72; do
  timeout 900 python bench.py --graph uniform --scale 26 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --lib-option sweep_slices=$sl > $out/uniform_slices$sl.json 2> $out/uniform_slices$sl.err
  echo "sweep_slices $sl: $(grep summary $out/uniform_slices$sl.err | cut -c1-120)"
done
for form in 1 3 18 0; do
  timeout 900 python bench.py --graph uniform --scale 26 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --lib-option sweep_slices=64 --lib-option blocked_form=$form > $out/uniform_s64_form$form.json 2> $out/uniform_s64_form$form.err
  echo "64 slices, blocked_form $form: $(grep summary $out/uniform_s64_form$form.err | cut -c1-120)"
done
timeout 900 python bench.py --graph uniform --scale 25 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra > $out/uniform25.json 2> $out/uniform25.err; echo "2^25 default: $(grep summary $out/uniform25.err | cut -c1-120)"
timeout 900 python bench.py --graph uniform --scale 25 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --lib-option blocked_rows=-1 > $out/uniform25_off.json 2> $out/uniform25_off.err; echo "2^25 off: $(grep summary $out/uniform25_off.err | cut -c1-120)"
timeout 900 python bench.py --graph uniform --scale 25 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --lib-option sweep_slices=32 > $out/uniform25_s32.json 2> $out/uniform25_s32.err; echo "2^25 32 slices: $(grep summary $out/uniform25_s32.err | cut -c1-120)"
