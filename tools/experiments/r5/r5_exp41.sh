# closing session: tools/blocked_bench.hip with slices the size of the Infinity Cache's share instead of an L2's (8 / 16 / 32 slices)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp41; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
timeout 300 build/blocked_bench 20 0 16 3 > $out/rmat20.txt 2>&1; grep -c " 0 of" $out/rmat20.txt; grep differ $out/rmat20.txt | grep -v " 0 of" | head -3
for S in 8 16 32 128; do
timeout 600 build/blocked_bench 26 1 $S 3 > $out/uniform26_s$S.txt 2>&1; grep -v "U 2\|U 8" $out/uniform26_s$S.txt
done
for S in 8 32 128; do
timeout 600 build/blocked_bench 26 0 $S 3 > $out/rmat26_s$S.txt 2>&1; grep -v "U 2\|U 8" $out/rmat26_s$S.txt
done
