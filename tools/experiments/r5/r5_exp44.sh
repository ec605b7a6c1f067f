# closing session: the column-blocked stream of the short rows inside the library: its tests, then the uniform 2^26 graph and RMAT-26 through bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp44; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_tiles.py -x -q -m gpu -k "blocked" > $out/pytest_blocked.txt 2>&1; tail -15 $out/pytest_blocked.txt
for mode in 0 -1; do
  timeout 900 python bench.py --graph uniform --scale 26 --steps 5 --warmup 2 --cpu-scale 0 --cpu-scale2 0 --no-extra --lib-option blocked_rows=$mode > $out/uniform_blocked$mode.json 2> $out/uniform_blocked$mode.err
  grep "summary" $out/uniform_blocked$mode.err | cut -c1-300
done
timeout 900 python bench.py --scale 26 --steps 10 --warmup 3 --cpu-scale 0 --cpu-scale2 0 --no-extra > $out/rmat26.json 2> $out/rmat26.err; grep "summary" $out/rmat26.err | cut -c1-300
