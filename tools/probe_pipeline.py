"""Do two progressive frames in flight (two contexts / streams) overlap on the GPU?  Run under rocprofv3 --kernel-trace."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg
import torch

r = pkg.Renderer(0)
r.set_scene(pkg.cornell9())
cam = pkg.pinhole_camera(vx=(1, 0, 0), vz=(0, 0, -1), org=(50, 45, 168), near=1.0)
for pipeline in (1, 2):
    prog = pkg.ProgressiveRenderer(r, 1280, 720, 1, camera=cam, pipeline=pipeline)
    for _ in range(5):
        prog.step()
    prog.flush()
    t0 = time.perf_counter()
    for _ in range(60):
        prog.step()
    prog.flush()
    print(f"pipeline {pipeline}: {(time.perf_counter() - t0) / 60 * 1e3:.3f} ms/frame", flush=True)
    prog.close()
