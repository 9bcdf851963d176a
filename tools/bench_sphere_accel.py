"""Config 5 (1024 spheres, 1024x768, 1024 spp) and larger tables: the exhaustive megakernel, the sphere hierarchy and the uniform
grid (spt_set_sphere_accel; the grid is the default), with oracle rows for parity."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

r = pkg.Renderer(0)
r.set_watchdog(120.0)
rows = []
for name, sc, w, h, samps, check_rows in (("config 5: 1024 spheres", pkg.random_spheres(1024, 1024), 1024, 768, 256, [100, 500]),
                                          ("4096 spheres", pkg.random_spheres(4096, 7), 1024, 768, 64, [300]),
                                          ("300 clustered spheres (sizes 0.03 .. 6, overlapping, some concentric)", None, 1024, 768, 64, [300]),
                                          ("16384 spheres (records beyond one CU's LDS: the default mode falls to the hierarchy)", pkg.random_spheres(16384, 5), 1024, 768, 16, [300])):
    if sc is None and name.startswith("300"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_sphere_accel import _cluster_scene
        sc = _cluster_scene(pkg, 300, 1)
    for accel in (pkg.ACCEL_EXHAUSTIVE, pkg.ACCEL_BVH, pkg.ACCEL_GRID):
        if accel == pkg.ACCEL_EXHAUSTIVE and len(sc) > 4096:        # SPT_MAX_SPHERES: the exhaustive kernels stage the table in LDS
            continue
        r.set_sphere_accel(accel)
        r.set_scene(sc)
        img, st = r.render(w, h, samps, seed=0, normalise=True)
        img, st = r.render(w, h, samps, seed=0, normalise=True)
        exact = True
        for row in check_rows:
            ref, _ = orc.render(sc, w, h, samps, seed=0, normalise=True, row_begin=row, row_count=1)
            exact &= bool(np.array_equal(img[row:row + 1], ref))
        out = {"scene": name, "accel": {0: "exhaustive", 1: "bvh", 2: "grid"}[accel], "kernel": r.last_kernel(), "image": f"{w}x{h}", "spp": 4 * samps,
               "kernel_ms": round(st["kernel_ms"], 2), "msamples_s": round(st["samples"] / st["kernel_ms"] / 1e3, 1),
               "bounces_per_sample": round(st["bounces"] / st["samples"], 4), "oracle_rows": len(check_rows), "bit_exact": exact}
        print(json.dumps(out), flush=True)
        rows.append(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
r.set_sphere_accel(pkg.ACCEL_GRID)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "sphere_accel.json"), "w"), indent=1)
