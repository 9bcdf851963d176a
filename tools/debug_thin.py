import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import test_meshes as tm
rs = np.random.RandomState(11)
S = pkg.make_sphere_trimesh
scenes = {"shipped": [S((-1, 0, -4), 1.0), S((1.5, 0, -5), 1.0)], "cornell-like": tm._mesh_scene(pkg)[0]}
r = pkg.Renderer(0)
for name, meshes in scenes.items():
    mats = [((0, 0, 0), (.5, .5, .5), pkg.DIFF)] * len(meshes)
    rays = tm._adversarial_rays(meshes, rs, 150000 if name == "shipped" else 60000)
    r.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE); r.set_meshes(meshes, mats); ref = r.trace_rays(rays)
    r.set_mesh_accel(pkg.ACCEL_BVH); got = r.trace_rays(rays)
    bad = np.unique(np.nonzero(got.view(np.uint8).reshape(len(rays), -1) != ref.view(np.uint8).reshape(len(rays), -1))[0])
    tri = np.concatenate([m.positions[m.indices.reshape(-1, 3)] for m in meshes]).astype(np.float64)
    first = np.cumsum([0] + [m.triangle_count for m in meshes])
    print(name, "rays", len(rays), "bad", len(bad))
    for i in bad[:12]:
        t = tri[first[ref["instId"][i]] + ref["triId"][i]]
        e1, e2 = t[1] - t[0], t[2] - t[0]
        n = np.cross(e1, e2); L2 = max(e1 @ e1, e2 @ e2, (e2 - e1) @ (e2 - e1))
        print(" ray", rays[i], "\n  ref", ref["dist"][i], ref["instId"][i], ref["triId"][i], "thin" if np.linalg.norm(n) <= L2 / 1024 else "regular", "|n|/L2 %.2e" % (np.linalg.norm(n) / L2),
              "\n  got", got["dist"][i], got["instId"][i], got["triId"][i], "\n  tri", t.tolist())
