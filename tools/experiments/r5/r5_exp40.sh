# closing session: the short rows as a column-blocked wave-stationary stream (tools/blocked_bench.hip), RMAT and uniform
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp40; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 300 build/blocked_bench 20 0 16 3 > $out/rmat20.txt 2>&1; tail -25 $out/rmat20.txt
timeout 600 build/blocked_bench 26 0 128 5 > $out/rmat26_s128.txt 2>&1; cat $out/rmat26_s128.txt
timeout 600 build/blocked_bench 26 1 128 5 > $out/uniform26_s128.txt 2>&1; cat $out/uniform26_s128.txt
timeout 600 build/blocked_bench 26 1 256 5 > $out/uniform26_s256.txt 2>&1; cat $out/uniform26_s256.txt
