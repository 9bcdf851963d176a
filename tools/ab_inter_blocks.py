"""Interactive frames (1280x720, 4 spp) against the number of pool-kernel workgroups per CU (spt_set_tuning blocks_per_cu; default 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, bench
import optix_test_smallpt_amd as pkg
os.environ["SPT_BENCH_NO_CPP"] = "1"
r = pkg.Renderer(0); r.set_scene(pkg.cornell9())
for rnd in range(2):
    for per_cu in (0, 3, 2, 1):
        r.set_tuning(per_cu, 0)
        i = bench.interactive(pkg, r, torch.device("cuda", 0), frames=300)
        print(f"blocks_per_cu {per_cu or 4}: {i['frames_per_s']} fps, {i['frames_per_s_two_in_flight']} two in flight, kernel_ms {i['kernel_ms']}", flush=True)
