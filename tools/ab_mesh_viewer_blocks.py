"""The viewer's loop on the reference's shipped mesh scene through the hierarchy (1280x720, 4 spp per frame): frames/s against the
mesh kernel's workgroups per CU (spt_set_tuning blocks_per_cu; default 4), one frame at a time and two in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import optix_test_smallpt_amd as pkg

meshes = [pkg.make_sphere_trimesh((50, 40.8, 81.6), 10.0), pkg.make_sphere_trimesh((50, 681.6 - .27, 81.6), 600.0)]
mats = [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((1, 1, 1), (0, 0, 0), pkg.DIFF)]
cam = pkg.smallpt_camera(1280, 720)
for rnd in range(2):
    for per_cu in (0, 3, 2, 1, 8):
        for pipeline in (1, 2):
            r = pkg.Renderer(0)
            r.set_mesh_accel(pkg.ACCEL_BVH)
            r.set_meshes(meshes, mats)
            r.set_tuning(per_cu, 0)
            prog = pkg.ProgressiveRenderer(r, 1280, 720, 1, camera=cam, pipeline=pipeline)
            for _ in range(10):
                prog.step()
            prog.flush() if pipeline > 1 else torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                prog.step()
            prog.flush() if pipeline > 1 else torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 300
            print(f"blocks_per_cu {per_cu or 4}, {pipeline} in flight: {1 / dt:.1f} frames/s", flush=True)
            if pipeline > 1:
                prog.close()
            r.close()
