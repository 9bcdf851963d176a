"""How often does the reference's triIntersect (scene.cpp:52-70: no determinant test) report a hit that is rounding noise?
Random rays against one tessellated sphere (4096 triangles) through the CPU oracle's brute force; a hit is "noise" when the
ray is geometrically (float64) farther than 1 % of the radius from the triangle it claims to have hit.  CPU only."""
import os
import sys
import types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as orc

nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
pos, nor, idx = orc.make_sphere_trimesh((0, 0, 0), 1.0, 32)
mesh = types.SimpleNamespace(positions=pos, normals=nor, indices=idx)
rs = np.random.RandomState(5)
o = rs.uniform(-3, 3, (nrays, 3)).astype(np.float32)
d = rs.randn(nrays, 3); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
hits = orc.trace_rays([mesh], np.concatenate([o, d], axis=1))
hit = hits["dist"] < 1e19
tri = idx[hits["triId"][hit]]
v0, v1, v2 = (pos[tri[:, k]].astype(np.float64) for k in range(3))
oo, dd = o[hit].astype(np.float64), d[hit].astype(np.float64)
n = np.cross(v1 - v0, v2 - v0)
n /= np.linalg.norm(n, axis=1, keepdims=True)
det = np.einsum("ij,ij->i", dd, n)
t = np.einsum("ij,ij->i", v0 - oo, n) / np.where(det == 0, 1e-300, det)
p = oo + dd * t[:, None]                                   # where the ray really crosses the triangle's plane
cen = (v0 + v1 + v2) / 3
far = np.linalg.norm(p - cen, axis=1) > 0.5                # edges are ~0.1 long: half a radius away is no hit of this triangle
print(f"{nrays} rays, {int(hit.sum())} hits, of them {int(far.sum())} with the true plane crossing > 0.5 from the triangle "
      f"(|cos(ray, plane normal)| of those: {np.sort(np.abs(det[far]))[:8]})")
print("distances reported for those:", hits["dist"][hit][far][:8])
