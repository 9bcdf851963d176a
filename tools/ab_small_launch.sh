#!/bin/bash
# Interactive frames (bench.py's interactive block: Python loop and the C++ host with 1 / 2 / 4 / 8 frames in flight) against the number
# of pool-kernel workgroups per CU that short launches get (SPT_SMALL_LAUNCH_BLOCKS; the library's default is 2, 4 = as for long launches).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for B in 4 3 2; do
    SPT_SMALL_LAUNCH_BLOCKS=$B timeout -k 10 300 python - <<PY
import json, os, sys
sys.path.insert(0, "$R")
import torch, bench
import optix_test_smallpt_amd as pkg
r = pkg.Renderer(0); r.set_scene(pkg.cornell9())
i = bench.interactive(pkg, r, torch.device("cuda", 0), frames=400)
print("blocks $B:", {k: v for k, v in i.items() if "frames_per_s" in k}, "kernel_ms", i["kernel_ms"], flush=True)
PY
  done
done
