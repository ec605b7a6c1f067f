# closing session: k_spmv_blocked in the library with the short rows dealt over the blocks in runs of 64: tests, then forms on the uniform 2^26 graph
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp45; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "blocked" > $out/pytest_blocked.txt 2>&1; tail -8 $out/pytest_blocked.txt
for form in 2 1 3 18 0; do
  timeout 900 python bench.py --graph uniform --scale 26 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --lib-option blocked_form=$form > $out/uniform_form$form.json 2> $out/uniform_form$form.err
  echo "blocked_form $form: $(grep summary $out/uniform_form$form.err | cut -c1-120)"
done
for sl in 64 96; do
  timeout 900 python bench.py --graph uniform --scale 26 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --lib-option sweep_slices=$sl > $out/uniform_slices$sl.json 2> $out/uniform_slices$sl.err
  echo "sweep_slices $sl: $(grep summary $out/uniform_slices$sl.err | cut -c1-120)"
done
