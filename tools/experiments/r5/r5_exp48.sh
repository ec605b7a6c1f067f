# closing session: blocked_bench, batches of 1 / 3 x 64 entries
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp48; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 600 build/blocked_bench 26 1 64 3 64 2 > $out/uniform26_s64.txt 2>&1; grep "prefetched" $out/uniform26_s64.txt | grep "window  2" | cut -c1-200
