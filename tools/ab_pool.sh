#!/bin/bash
# A/B of pool-kernel builds on one box: interleaved processes, SPT_LIB selects the build (tools/build_grid_variants.sh shows how a
# variant library is linked).  Per build: the lone-chain latency probe, config 2 (5 steps) and the interactive frames of bench.py.
# usage: bash tools/ab_pool.sh variants/lib_a.so variants/lib_b.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for L in "$@"; do
    echo "== $L (round $round)"
    SPT_LIB=$R/$L timeout -k 10 120 python $R/tools/probe_chain_latency.py 2>&1 | grep -E "mirror|glass 1x1"
    SPT_LIB=$R/$L SPT_BENCH_NO_CPP=1 timeout -k 10 300 python - <<PY
import json, os, sys
sys.path.insert(0, "$R")
import torch, bench
import optix_test_smallpt_amd as pkg
r = pkg.Renderer(0); r.set_scene(pkg.cornell9())
t = torch.empty((768, 1024, 3), dtype=torch.float32, device="cuda")
ks = []
for i in range(6):
    r.render_rows_device(t, 1024, 768, 0, 768, 256, seed=0, normalise=True); st = r.sync(); ks.append(st["kernel_ms"])
print("config 2 kernel_ms", [round(k, 2) for k in ks[1:]], "Msamples/s", round(st["samples"] / min(ks[1:]) / 1e3, 1), "checksum", float(t.double().sum()))
i = bench.interactive(pkg, r, torch.device("cuda", 0), frames=300)
print("interactive", i["frames_per_s"], "fps,", i["frames_per_s_two_in_flight"], "two in flight, kernel_ms", i["kernel_ms"])
PY
  done
done
