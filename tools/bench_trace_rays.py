"""The Intersector seam (spt_trace_rays: host buffers in and out, smallpt.cpp:460-470; spt_trace_rays_device: device buffers) on the
reference's shipped mesh scene: rays/s by ray count, exhaustive loop and hierarchy; the device variant is checked against the host one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import optix_test_smallpt_amd as pkg

r = pkg.Renderer(0)
meshes = [pkg.make_sphere_trimesh((50, 40.8, 81.6), 10.0), pkg.make_sphere_trimesh((50, 681.6 - .27, 81.6), 600.0)]
mats = [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((1, 1, 1), (0, 0, 0), pkg.DIFF)]
r.set_meshes(meshes, mats)
rs = np.random.RandomState(1)
for accel, name in ((pkg.ACCEL_EXHAUSTIVE, "exhaustive"), (pkg.ACCEL_BVH, "hierarchy")):
    r.set_mesh_accel(accel)
    for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22):
        if accel == pkg.ACCEL_EXHAUSTIVE and n > (1 << 20):
            continue
        o = np.tile(np.array([50, 45, 160], dtype=np.float32), (n, 1)) + rs.randn(n, 3).astype(np.float32)
        d = rs.randn(n, 3).astype(np.float32); d[:, 2] = -np.abs(d[:, 2]) - 1; d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.ascontiguousarray(np.concatenate([o, d], axis=1))
        hits = r.trace_rays(rays)
        t0 = time.perf_counter(); reps = 3
        for _ in range(reps):
            hits = r.trace_rays(rays)
        dt = (time.perf_counter() - t0) / reps
        rt = torch.from_numpy(rays).cuda()
        ht = r.trace_rays_device(rt); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ht = r.trace_rays_device(rt, ht)
        torch.cuda.synchronize()
        dtd = (time.perf_counter() - t0) / reps
        same = np.array_equal(ht.cpu().numpy().view(np.uint32), hits.view(np.uint32).reshape(n, 11))
        print(f"{name}: {n} rays: host buffers {n / dt / 1e6:.1f} Mrays/s ({dt * 1e3:.2f} ms), device buffers {n / dtd / 1e6:.1f} Mrays/s ({dtd * 1e3:.3f} ms), identical {same}, hit rate {float((hits['dist'] < 1e30).mean()):.2f}", flush=True)
