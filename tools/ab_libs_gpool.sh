#!/bin/bash
# A/B of grid-pool kernel builds on one box (tools/build_gpool_variants.sh): interleaved processes, SPT_LIB selects the build.
# usage: bash tools/ab_libs_gpool.sh "<bench_grid args>" name1 name2 ...      (variants/lib_<name>.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=$1; shift
for round in 1 2; do
  for L in "$@"; do
    printf "%-14s r%d " $L $round
    SPT_LIB=$R/variants/lib_$L.so timeout -k 10 200 python $R/tools/bench_grid.py $ARGS --nocheck 2>&1 | grep msamples | sed -e 's/.*"kernel_ms"/"kernel_ms"/' -e 's/, "bounces.*//'
  done
done
