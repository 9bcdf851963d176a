#!/bin/bash
# Builds compiler-flag variants of the kernel library into gpurun_out/variants/ for A/B timing with SPT_LIB=...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/optix-test-smallpt_amd/csrc
O=$R/variants
mkdir -p $O
build() {  # name, extra flags
  name=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "$@" --offload-arch=gfx950 -c $C/spt_kernel.hip -o $O/k_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $O/k_$name.o $C/spt_api.o -o $O/lib_$name.so
  echo built $name
}
build base -fno-slp-vectorize
build slp
build maxilp -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp
build maxmem -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause
build o2 -O2 -fno-slp-vectorize
build noswp -fno-slp-vectorize -mllvm -amdgpu-schedule-metric-bias=100
