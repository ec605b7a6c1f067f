cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/s3v1; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -4 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-400 $out/bench_default.json
bash tools/bfs_timeline.sh
