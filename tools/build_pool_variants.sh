#!/bin/bash
# Builds -D variants of the pool kernel (spt_pool.hip) into variants/ for A/B timing on one box (SPT_LIB=variants/lib_<name>.so).
# usage: bash tools/build_pool_variants.sh name1 "-DFLAG=1 ..." name2 "..." ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/optix-test-smallpt_amd/csrc
O=$R/variants
mkdir -p $O
make -C $C -s
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $flags --offload-arch=gfx950 -I$C -c $C/spt_pool.hip -o $O/p_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $C/spt_kernel.o $O/p_$name.o $C/spt_mesh.o $C/spt_grid.o $C/spt_gpool.o $C/spt_bvh.o $C/spt_gridb.o $C/spt_api.o -o $O/lib_$name.so
  echo built $name
done
