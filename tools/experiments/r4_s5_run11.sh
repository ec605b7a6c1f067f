#!/bin/bash
# closing session of round 4, GPU call 11: graphs that KEEP their edge values (the unchanged applications) are not swept; does the tile count
# fitted to the sweep hurt them?  unchanged PageRank.cpp at RMAT-24 / 25 with the old rule's count (automatic for such graphs) and the new one forced
cd $GRAFT_REPO_ROOT; out=gpurun_out/s5; mkdir -p $out
for sc in 24 25; do
  python tools/app_at_scale.py $sc 2>&1 | grep "== unchanged PageRank" | cut -c1-260
  for t in 1 2 3 5; do
    echo "GRAPHMAT_COL_TILES=$t: $(GRAPHMAT_COL_TILES=$t build/ref_apps/PageRank /tmp/rmat$sc.bin.mtx 2>&1 | grep -E 'PR Time|Completed' | tr '\n' ' ' | cut -c1-200)"
  done
  rm -f /tmp/rmat$sc.bin.mtx*
done
