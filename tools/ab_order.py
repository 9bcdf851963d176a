"""Cost-ordered dispatch of the pool kernel (spt_api.cpp, tuning bit 13 switches it off): config 2 and the interactive frames with and
without it, in one process, alternating."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
import optix_test_smallpt_amd as pkg
from pool_check import pool_report

os.environ["SPT_BENCH_NO_CPP"] = "1"
r = pkg.Renderer(0)
r.set_scene(pkg.cornell9())
t = torch.empty((768, 1024, 3), dtype=torch.float32, device="cuda")
samps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for rnd in range(2):
    for label, variant in (("static order", 0x2000), ("cost order", 0)):
        r.set_tuning(0, variant)
        ks = []
        for i in range(6):
            r.render_rows_device(t, 1024, 768, 0, 768, samps, seed=0, normalise=True); st = r.sync(); ks.append(st["kernel_ms"])
        print(f"{label}: config 2 kernel_ms {[round(k, 2) for k in ks]} Msamples/s {st['samples'] / min(ks[1:]) / 1e3:.1f} checksum {float(t.double().sum())!r}", flush=True)
        print("   ", pool_report(r, st)[:215], flush=True)
        i = bench.interactive(pkg, r, torch.device("cuda", 0), frames=300)
        print(f"{label}: interactive {i['frames_per_s']} fps, {i['frames_per_s_two_in_flight']} two in flight, kernel_ms {i['kernel_ms']}", flush=True)
r.set_tuning(0, 0)
