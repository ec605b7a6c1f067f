# closing session: the reference's unchanged PageRank.cpp on a graph without skew (int edge values kept): the column-blocked stream with edge values
# (automatic) against GRAPHMAT_COL_TILES=1 (no slices: the row-blocks)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/exp51; mkdir -p $out
export LD_LIBRARY_PATH=$R/graphmat_amd
for sc in 24 25; do
  GRAPHMAT_VERBOSE=1 timeout 1500 python tools/app_at_scale.py $sc uniform > $out/uniform${sc}_auto.txt 2>&1
  grep "==\|column-blocked" $out/uniform${sc}_auto.txt | sort | uniq -c | cut -c1-330
  GRAPHMAT_COL_TILES=1 timeout 1500 python tools/app_at_scale.py $sc uniform > $out/uniform${sc}_tiles1.txt 2>&1
  grep "==" $out/uniform${sc}_tiles1.txt | cut -c1-330
done
