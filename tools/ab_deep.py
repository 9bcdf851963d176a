"""Pool kernel, deep-slot batches (spt_pool.hip SPT_POOL_DEEP): config 2 with a NEW seed every launch (static dispatch order: what bench.py
times) and the viewer's serial frames, for the library SPT_LIB names.  usage: ab_deep.py [samps per cell]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import optix_test_smallpt_amd as pkg
r = pkg.Renderer(0)
r.set_scene(pkg.cornell9())
t = torch.empty((768, 1024, 3), dtype=torch.float32, device="cuda")
samps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ks = []
for i in range(8):
    r.render_rows_device(t, 1024, 768, 0, 768, samps, seed=1000 + i, normalise=True); st = r.sync(); ks.append(st["kernel_ms"])
k = ks[2:]
line = f"{4 * samps} spp, new seed every launch: kernel_ms {[round(v, 2) for v in ks]} mean(3..8) {sum(k) / len(k):.2f} -> {st['samples'] / (sum(k) / len(k)) / 1e3:.0f} Msamples/s checksum {float(t.double().sum())!r}"
cam = pkg.pinhole_camera()
f = torch.empty((720, 1280, 3), dtype=torch.float32, device="cuda")
import time
r.render_rows_device(f, 1280, 720, 0, 720, 1, seed=0, camera=cam); r.sync()
t0 = time.perf_counter()
for i in range(200):
    r.render_rows_device(f, 1280, 720, 0, 720, 1, seed=i, camera=cam); r.sync()
fps = 200 / (time.perf_counter() - t0)
print(line, f"| viewer frames (1280x720, 4 spp, serial): {fps:.0f} frames/s", flush=True)
