#!/bin/bash
# One-call evidence collection on the GPU box: rocprofv3 kernel stats of the default bench command, the six PMC passes,
# and the BASELINE.json configs table.  usage: bash tools/prof_round.sh <tag>   (outputs under gpurun_out/<tag>_*)
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_stats_bench.json 2> $R/gpurun_out/${TAG}_stats.log
find $R/gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_bench_kernel_stats.csv \;
cd $R
bash tools/prof_pmc.sh ${TAG}_pmc > gpurun_out/${TAG}_pmc.out 2>&1
timeout -k 10 600 python tools/run_configs.py > gpurun_out/${TAG}_configs.log 2>&1
cp gpurun_out/configs.json gpurun_out/${TAG}_configs.json; cp gpurun_out/configs.md gpurun_out/${TAG}_configs.md
tail -3 gpurun_out/${TAG}_stats_bench.json; cat gpurun_out/${TAG}_bench_kernel_stats.csv | head -8; grep -A24 "poolkernel" gpurun_out/${TAG}_pmc_summary.txt | head -30; cat gpurun_out/${TAG}_configs.md
