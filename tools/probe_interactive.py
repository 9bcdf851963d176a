"""Where does an interactive frame (1280x720, 4 spp, pinhole camera inside the Cornell box) spend its time?"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg
import torch

r = pkg.Renderer(0)
r.set_scene(pkg.cornell9())
w, h = 1280, 720
cam = pkg.pinhole_camera(vx=(1, 0, 0), vz=(0, 0, -1), org=(50, 45, 168), near=1.0)
prog = pkg.ProgressiveRenderer(r, w, h, 1, camera=cam)
for _ in range(5):
    prog.step()
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    prog.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
st = r.sync()
print(f"step(): {dt * 1e3:.3f} ms/frame, kernel {st['kernel_ms']:.3f} finalize {st['finalize_ms']:.4f} bounces/sample {st['bounces'] / st['samples']:.2f} "
      f"grid {st['grid_blocks']} kernel={r.last_kernel()}", flush=True)
frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
for k in range(n):
    r.render_rows_device(frame, w, h, 0, h, 1, seed=k, normalise=False, camera=cam, stream=stream)
torch.cuda.synchronize()
print(f"render_rows_device only, no sync per frame: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame", flush=True)
t0 = time.perf_counter()
for k in range(n):
    r.render_rows_device(frame, w, h, 0, h, 1, seed=k, normalise=False, camera=cam, stream=stream)
    r.sync()
print(f"render_rows_device + sync: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame", flush=True)
for per_cu in (1, 2, 3):
    r.set_tuning(per_cu, 0)
    r.render_rows_device(frame, w, h, 0, h, 1, seed=0, normalise=False, camera=cam, stream=stream); st = r.sync()
    print(f"blocks/CU {per_cu}: kernel {st['kernel_ms']:.3f} ms", flush=True)
r.set_tuning(0, 0x400)
r.render_rows_device(frame, w, h, 0, h, 1, seed=0, normalise=False, camera=cam, stream=stream); st = r.sync()
print(f"megakernel: kernel {st['kernel_ms']:.3f} ms", flush=True)
