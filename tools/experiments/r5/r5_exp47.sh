# closing session: blocked_bench, two workgroups of 512 threads / 16384 rows per CU (out of phase with each other) against one of 1024 / 32768
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp47; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
for S in 64 96; do
timeout 600 build/blocked_bench 26 1 $S 3 64 2 > $out/w16_uniform26_s$S.txt 2>&1; grep "prefetched\|^workgroup-stat" $out/w16_uniform26_s$S.txt | cut -c1-200
timeout 600 build/blocked_bench_w8 26 1 $S 3 64 2 > $out/w8_uniform26_s$S.txt 2>&1; grep "prefetched\|^workgroup-stat" $out/w8_uniform26_s$S.txt | cut -c1-200
done
