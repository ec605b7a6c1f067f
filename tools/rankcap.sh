#!/bin/bash
# experiment: hybrid device order (hot vertices ranked by degree, the rest in native order)
for cap in 0 16 64 256 1024 4096; do
  echo "== rank_cap $cap"; python bench.py --scale ${1:-26} --steps 10 --warmup 3 --cpu-scale 0 --rank-cap $cap 2>&1 | grep -o '"ms_per_step": [0-9.]*\|summary.*'
done
