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
# the extra configurations of bench.py (configs 3 and 5, the shipped triangle scene): kernel statistics + PMC passes of each
for X in config5 config3 mesh_256spp; do
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${X}_stats -- python $R/bench.py --only-extra $X > $R/gpurun_out/${TAG}_${X}.json 2> $R/gpurun_out/${TAG}_${X}_stats.log )
  find $R/gpurun_out/${TAG}_${X}_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_${X}_kernel_stats.csv \;
  SPT_PMC_ONLY="0 1 2 3 4 6 7" bash tools/prof_pmc_cmd.sh ${TAG}_${X}_pmc python bench.py --only-extra $X > gpurun_out/${TAG}_${X}_pmc.out 2>&1
  echo "== $X"; cat gpurun_out/${TAG}_${X}.json | cut -c1-400; head -4 gpurun_out/${TAG}_${X}_kernel_stats.csv | cut -c1-160
done
tail -3 gpurun_out/${TAG}_stats_bench.json; cat gpurun_out/${TAG}_bench_kernel_stats.csv | head -8; grep -A24 "poolkernel" gpurun_out/${TAG}_pmc_summary.txt | head -30; cat gpurun_out/${TAG}_configs.md
