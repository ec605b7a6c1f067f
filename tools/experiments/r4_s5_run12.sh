#!/bin/bash
# closing session of round 4, GPU call 12: unchanged PageRank.cpp (keeps edge values, ordered fold, not swept) at RMAT-26 by tile count
cd $GRAFT_REPO_ROOT
python tools/app_at_scale.py 26 2>&1 | grep "== unchanged PageRank" | cut -c1-260
for t in 1 3 8; do
  echo "GRAPHMAT_COL_TILES=$t: $(GRAPHMAT_COL_TILES=$t build/ref_apps/PageRank /tmp/rmat26.bin.mtx 2>&1 | grep -E 'PR Time|Completed' | tr '\n' ' ' | cut -c1-200)"
done
echo "GRAPHMAT_TRUST_PROBE=1 automatic: $(GRAPHMAT_TRUST_PROBE=1 build/ref_apps/PageRank /tmp/rmat26.bin.mtx 2>&1 | grep -E 'PR Time|Completed' | tr '\n' ' ' | cut -c1-200)"
echo "GRAPHMAT_TRUST_PROBE=1 GRAPHMAT_COL_TILES=1: $(GRAPHMAT_TRUST_PROBE=1 GRAPHMAT_COL_TILES=1 build/ref_apps/PageRank /tmp/rmat26.bin.mtx 2>&1 | grep -E 'PR Time|Completed' | tr '\n' ' ' | cut -c1-200)"
rm -f /tmp/rmat26.bin.mtx*
