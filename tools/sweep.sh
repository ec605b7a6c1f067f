#!/bin/bash
# usage: tools/sweep.sh <scale> "<extra args 1>" "<extra args 2>" ...
sc=$1; shift
for a in "$@"; do
  echo "== scale $sc $a"
  python bench.py --scale $sc --steps 10 --warmup 2 --cpu-scale 0 $a 2>&1 >/dev/null | grep summary
done
