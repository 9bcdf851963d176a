"""Measured rows for the triangle path (f4): the scene the reference ships (`spheres[]`, smallpt.cpp:31-34: r=10 red diffuse +
r=600 light, each tessellated into 4096 triangles by the Sphere constructor, scene.h:91), 256x256, 4 spp -- the
"as shipped" probe of BASELINE.md section 2 (6.62 s on 6 CPU threads for one normal-visualisation bounce) -- and main()'s
SingleTriangleScene at 1280x720.  Reports kernel time, rays and ray-triangle tests per second, the FP32 rate of the
triIntersect arithmetic (52 flop per call as the reference writes it, scene.cpp:52-70) and oracle parity on two rows."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import optix_test_smallpt_amd as pkg  # noqa: E402
import oracle_binding as orc  # noqa: E402

PEAK = 157.3


def run(name, meshes, mats, w, h, samps, camera, rows, accel=0):
    r = pkg.Renderer(0)
    r.set_mesh_accel(accel)
    t0 = time.perf_counter()
    r.set_meshes(meshes, mats)
    setup_ms = (time.perf_counter() - t0) * 1e3
    ntri = sum(m.triangle_count for m in meshes)
    r.render(w, h, samps, seed=0, normalise=True, camera=camera)
    best = None
    for _ in range(3):
        img, st = r.render(w, h, samps, seed=0, normalise=True, camera=camera)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    st = best
    exact = True
    for row in rows:
        ref, _ = orc.render_meshes(meshes, mats, w, h, samps, seed=0, normalise=True, row_begin=row, row_count=1, camera=camera)
        exact &= bool(np.array_equal(img[row:row + 1], ref))
    tests = st["bounces"] * ntri                 # what the exhaustive loop evaluates; with the hierarchy: the work it replaces
    out = {"config": name, "accel": "bvh" if accel else "exhaustive", "set_meshes_ms": round(setup_ms, 2), "triangles": ntri, "image": f"{w}x{h}", "spp": 4 * samps, "kernel_ms": round(st["kernel_ms"], 3),
           "msamples_s": round(st["samples"] / st["kernel_ms"] / 1e3, 2), "mrays_s": round(st["bounces"] / st["kernel_ms"] / 1e3, 2),
           "bounces_per_sample": round(st["bounces"] / st["samples"], 3),
           "gtests_s": round(tests / st["kernel_ms"] / 1e6, 1), "tflops_triIntersect": round(tests * 52 / st["kernel_ms"] / 1e9, 2),
           "pct_of_157.3": round(100 * tests * 52 / st["kernel_ms"] / 1e9 / PEAK, 2), "oracle_rows": len(rows), "bit_exact": exact}
    print(json.dumps(out), flush=True)
    r.close()
    return out


def viewer_loop(name, meshes, mats, frames=200, pinhole=False, accel=None):
    import torch
    r = pkg.Renderer(0)
    r.set_mesh_accel(pkg.ACCEL_BVH if accel is None else accel)
    r.set_meshes(meshes, mats)
    # pinhole: the interactive driver's Camera (smallpt.cpp:607-641) at the smallpt camera's position and direction -- one origin for
    # the whole frame, so the rays of depth 0 test the list of planes through it instead of walking the plane tree
    cam = pkg.pinhole_camera(vz=(0, -0.042573, -0.999093), org=(50, 52, 295.6)) if pinhole else pkg.smallpt_camera(1280, 720)
    prog = pkg.ProgressiveRenderer(r, 1280, 720, 1, camera=cam)
    for _ in range(10):
        prog.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        prog.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    out = {"config": name, "accel": {None: "bvh", 0: "exhaustive", 1: "bvh", 3: "bvh-fast"}[accel], "camera": "pinhole" if pinhole else "smallpt", "triangles": sum(m.triangle_count for m in meshes), "image": "1280x720", "spp": 4,
           "frames_per_s": round(1.0 / dt, 1), "ms_per_frame": round(dt * 1e3, 3), "accum_nonzero": bool(float(prog.accum.abs().sum()) > 0)}
    print(json.dumps(out), flush=True)
    r.close()
    return out


def main():
    rows = []
    # the reference's live global table: Sphere(10, (50,40.8,81.6), 0, (.75,.25,.25), DIFF), Sphere(600, (50,681.6-.27,81.6), (1,1,1), 0, DIFF)
    meshes = [pkg.make_sphere_trimesh((50, 40.8, 81.6), 10.0), pkg.make_sphere_trimesh((50, 681.6 - .27, 81.6), 600.0)]
    mats = [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((1, 1, 1), (0, 0, 0), pkg.DIFF)]
    rows.append(run("f4: the reference's shipped scene, 2 tessellated spheres, cpuRender camera, 4 spp", meshes, mats, 256, 256, 1, None, [100, 200]))
    rows.append(run("f4: same scene, 256 spp (lanes regenerate paths: the steady-state rate)", meshes, mats, 256, 256, 64, None, [128]))
    # the same through the hierarchy (SPT_ACCEL_BVH: the OptiX Prime model's role, smallpt.cpp:475-603), and at the viewer's size
    rows.append(run("f4: shipped scene, 4 spp, hierarchy", meshes, mats, 256, 256, 1, None, [100, 200], accel=1))
    rows.append(run("f4: shipped scene, 256 spp, hierarchy", meshes, mats, 256, 256, 64, None, [128], accel=1))
    rows.append(run("f4: shipped scene at the viewer's 1280x720, 4 spp, hierarchy", meshes, mats, 1280, 720, 1, None, [360], accel=1))
    big = [pkg.make_sphere_trimesh((50, 40.8, 81.6), 10.0, 256), pkg.make_sphere_trimesh((50, 681.6 - .27, 81.6), 600.0, 256)]
    rows.append(run("f4: the two spheres at subdivision 256 (2 x 262144 triangles), 1280x720, 4 spp, hierarchy (oracle rows skipped: "
                    "0.5 M triangles per ray on the CPU)", big, mats, 1280, 720, 1, None, [], accel=1))
    rows.append(viewer_loop("f3 + f4: the viewer's render loop (ProgressiveRenderer, smallpt.cpp:895-942) on the shipped mesh scene, 1280x720, "
                            "1 sample per jitter cell per frame, hierarchy", meshes, mats))
    for accel, frames in ((pkg.ACCEL_BVH, 200), (pkg.ACCEL_BVH_FAST, 200), (pkg.ACCEL_EXHAUSTIVE, 20)):
        rows.append(viewer_loop("f3 + f4: the same loop with the interactive driver's pinhole Camera (one origin per frame)", meshes, mats, frames=frames, pinhole=True, accel=accel))
    meshes, mats = pkg.single_triangle_scene()
    rows.append(run("f4: SingleTriangleScene of main(), viewer camera", meshes, mats, 1280, 720, 1, pkg.pinhole_camera(), [300, 500]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "mesh_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
