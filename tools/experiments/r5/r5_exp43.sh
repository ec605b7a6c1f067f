# closing session: tools/blocked_bench.hip, the workgroup-stationary form with the workgroups of an XCD kept in step slice by slice
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp43; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 300 build/blocked_bench 20 1 16 3 64 2 > $out/uniform20.txt 2>&1; cat $out/uniform20.txt
for S in 128 64 96; do
timeout 600 build/blocked_bench 26 1 $S 3 64 2 > $out/uniform26_s$S.txt 2>&1; cat $out/uniform26_s$S.txt
done
timeout 600 build/blocked_bench 26 0 128 3 64 2 > $out/rmat26_s128.txt 2>&1; cat $out/rmat26_s128.txt
