"""A bounce-heavy mesh scene (tests/test_meshes.py _mesh_scene: a 1000-unit floor ball, a light, diffuse / mirror / glass balls, the
single triangle; 6.6 bounces per sample) through the three closest-hit modes: what the exact hierarchy costs when most rays are NOT
camera rays (they walk the plane tree)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import optix_test_smallpt_amd as pkg  # noqa: E402
from test_meshes import _mesh_scene  # noqa: E402

meshes, mats = _mesh_scene(pkg)
if len(sys.argv) > 1 and sys.argv[1] == "diffuse":             # the same geometry, every surface diffuse (colour .75): bounce rays without deep chains
    mats = [(m[0], (.75, .75, .75) if sum(m[0]) == 0 else m[1], pkg.DIFF) for m in mats]
w, h, samps = 256, 192, 16
ref = None
for name, accel in (("exhaustive", pkg.ACCEL_EXHAUSTIVE), ("bvh", pkg.ACCEL_BVH), ("bvh_fast", pkg.ACCEL_BVH_FAST), ("auto (the default)", pkg.ACCEL_AUTO)):
    with pkg.Renderer(0) as r:
        r.set_mesh_accel(accel)
        r.set_meshes(meshes, mats)
        r.render(w, h, samps, seed=1)
        ks = []
        for _ in range(3):
            img, st = r.render(w, h, samps, seed=1)
            ks.append(st["kernel_ms"])
        if ref is None:
            ref = img
        print(json.dumps({"accel": name, "triangles": sum(m.triangle_count for m in meshes), "image": f"{w}x{h}", "spp": 4 * samps, "kernel_ms": round(min(ks), 3), "ran": r.last_kernel(),
                          "bounces_per_sample": round(st["bounces"] / st["samples"], 3), "mrays_s": round(st["bounces"] / min(ks) / 1e3, 1),
                          "identical_to_exhaustive": bool(np.array_equal(img, ref))}), flush=True)
