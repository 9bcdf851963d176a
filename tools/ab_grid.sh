#!/bin/bash
# A/B of grid-kernel builds on one box: interleaved processes, SPT_LIB selects the build.
# usage: bash tools/ab_grid.sh "<bench_grid args>" variants/lib_a.so variants/lib_b.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=$1; shift
for round in 1 2; do
  for L in "$@"; do
    echo "== $L (round $round)"
    SPT_LIB=$R/$L timeout -k 10 200 python $R/tools/bench_grid.py $ARGS --nocheck 2>&1 | grep msamples | sed -e 's/.*"kernel_ms"/"kernel_ms"/' -e 's/, "bounces.*//'
  done
done
