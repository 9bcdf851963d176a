#!/bin/bash
# A/B of the grid-pool kernel's pool geometry on one box: SPT_GPOOL="S,R,drain,min_batch" per process (tools/bench_grid.py, config 5).
# usage: bash tools/ab_gpool.sh "<bench_grid args>" "S,R,drain,minb" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=$1; shift
for G in "$@"; do
  echo "== SPT_GPOOL=$G"
  SPT_GPOOL=$G timeout -k 10 200 python $R/tools/bench_grid.py $ARGS --nocheck 2>&1 | grep msamples | sed -e 's/.*"kernel": /"kernel": /' -e 's/"image.*"kernel_ms"/"kernel_ms"/' -e 's/, "bounces_per_sample[^,]*,//' -e 's/"oracle_rows": 0, "bit_exact": true,//'
done
