"""How uneven are the contiguous row bands of bench.py's weak-scaling image (1024 x 768 N, N = 8) and of config 4
(4096^2 over 8 ranks)?  Each band rendered on this GPU, kernel time and bounces per sample."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg
import torch
from optix_test_smallpt_amd.distributed import row_band

r = pkg.Renderer(0)
r.set_scene(pkg.cornell9())
for name, w, h, samps in (("bench --gpus 8: 1024x6144", 1024, 6144, 32), ("config 4: 4096x4096", 4096, 4096, 8)):
    ts = []
    for rank in range(8):
        b, c = row_band(h, 8, rank)
        t = torch.empty((c, w, 3), dtype=torch.float32, device="cuda")
        r.render_rows_device(t, w, h, b, c, samps, seed=0, normalise=True)
        st = r.sync()
        ts.append(st["kernel_ms"])
        print(f"{name} rank {rank}: rows {b}..{b + c - 1} kernel {st['kernel_ms']:.2f} ms  bounces/sample {st['bounces'] / st['samples']:.3f}", flush=True)
    print(f"{name}: max/mean = {max(ts) / (sum(ts) / 8):.3f}  (weak-scaling efficiency bound {sum(ts) / 8 / max(ts):.3f})", flush=True)
