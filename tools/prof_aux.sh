#!/bin/bash
# rocprofv3 kernel statistics of the auxiliary benches (triangle path, sphere hierarchy).  usage: bash tools/prof_aux.sh <tag>
set -u
TAG=${1:-aux}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_mesh -- python $R/tools/bench_mesh.py > $R/gpurun_out/${TAG}_mesh.log 2>&1
find $R/gpurun_out/${TAG}_mesh -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_mesh_kernel_stats.csv \;
echo "mesh done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_sbvh -- python $R/tools/bench_sphere_accel.py > $R/gpurun_out/${TAG}_sbvh.log 2>&1
find $R/gpurun_out/${TAG}_sbvh -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_sbvh_kernel_stats.csv \;
echo "sbvh done"
head -8 $R/gpurun_out/${TAG}_mesh_kernel_stats.csv | cut -c1-200; head -6 $R/gpurun_out/${TAG}_sbvh_kernel_stats.csv | cut -c1-200
