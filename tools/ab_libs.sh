#!/bin/bash
# A/B of kernel-library builds on one box: interleaved processes, SPT_LIB selects the build.  usage: bash tools/ab_libs.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for L in "$@"; do
    echo "== $L (round $round)"
    SPT_LIB=$R/$L timeout -k 10 120 python $R/tools/exp_pool_sizes.py 0:4 2>&1 | tail -1
  done
done
