"""Fuzz of the exact triangle hierarchy on the GPU: random mesh scenes (tessellated spheres of random subdivision and size, triangle
soups, coplanar soups with slivers / collinear / zero-edge triangles, single quads, everything far from the origin or tiny), random and
adversarial rays (tests/test_meshes.py: aimed at vertices / edges / centroids, along edges, axis-parallel, in a triangle's plane anywhere
in it and tilted / lifted out of it, across the supporting lines of long edges), spt_trace_rays through SPT_ACCEL_BVH against
SPT_ACCEL_EXHAUSTIVE, byte for byte.  FUZZ_SECONDS (default 120), FUZZ_SEED.  Prints one line per 20 scenes and a summary."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import optix_test_smallpt_amd as pkg  # noqa: E402
from test_meshes import _adversarial_rays, _degenerate_rays  # noqa: E402


def soup(rs, n, flat, spread, scale):
    c = rs.uniform(-spread, spread, (n, 1, 3))
    v = c + rs.uniform(-1, 1, (n, 3, 3)) * scale * 10.0 ** rs.uniform(-1, 1, (n, 1, 1))
    if flat:
        v[:, :, 1] = 3.0
        k = rs.rand(n)
        v[k < 0.15, 2] = 0.5 * (v[k < 0.15, 0] + v[k < 0.15, 1]) + rs.uniform(-1e-5, 1e-5, ((k < 0.15).sum(), 3)) * [1, 0, 1]   # slivers
        v[(k > 0.15) & (k < 0.2), 1] = v[(k > 0.15) & (k < 0.2), 0]                                                              # an edge of length zero
        v[(k > 0.2) & (k < 0.25), 2] = 2 * v[(k > 0.2) & (k < 0.25), 1] - v[(k > 0.2) & (k < 0.25), 0]                           # collinear
    pos = v.reshape(-1, 3).astype(np.float32)
    nor = np.tile(np.array([0, 1, 0], dtype=np.float32), (len(pos), 1))
    return pkg.TriMesh(pos, nor, np.arange(3 * n, dtype=np.uint32).reshape(n, 3))


def scene(rs):
    kind = rs.randint(6)
    off = rs.choice([0.0, 0.0, 1e3, 3e4]) * rs.uniform(-1, 1, 3)
    S = pkg.make_sphere_trimesh
    if kind == 0:
        return [S(tuple(off + rs.uniform(-5, 5, 3)), float(10.0 ** rs.uniform(-1, 1.5)), int(rs.choice([4, 8, 16, 24, 32, 48]))) for _ in range(rs.randint(1, 4))]
    if kind == 1:
        return [soup(rs, int(rs.choice([1, 5, 60, 800, 3000])), False, 10.0, 1.0)]
    if kind == 2:
        return [soup(rs, int(rs.choice([3, 40, 900])), True, 10.0, 1.0), S((0, 3, 0), 2.0, 8)]
    if kind == 3:
        return [S(tuple(off), 1e-3, 8), S(tuple(off + np.array([0, -1e3 - 2, -6])), 1e3, 16)]
    if kind == 4:
        h = float(10.0 ** rs.uniform(0, 2))
        q = np.array([[-h, 0, -h], [h, 0, -h], [h, 0, h], [-h, 0, h]], dtype=np.float32) + off.astype(np.float32)
        quad = pkg.TriMesh(q, np.tile(np.array([0, 1, 0], dtype=np.float32), (4, 1)), np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32))
        return [quad, S(tuple(off + np.array([0, 0.3 * h, 0])), 0.25 * h, int(rs.choice([8, 20])))]
    return [S(tuple(off + rs.uniform(-2, 2, 3)), 1.0, 64)]


def main():
    seconds = float(os.environ.get("FUZZ_SECONDS", "120"))
    rs = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "1")))
    t0 = time.time()
    scenes = rays_total = bad_total = hits = 0
    with pkg.Renderer(0) as r:
        while time.time() - t0 < seconds:
            meshes = scene(rs)
            mats = [((0, 0, 0), (.5, .5, .5), pkg.DIFF)] * len(meshes)
            rays = np.concatenate([_adversarial_rays(meshes, rs, 20000), _degenerate_rays(meshes, rs, 1500)])
            rays = rays[np.isfinite(rays).all(axis=1)]
            r.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
            r.set_meshes(meshes, mats)
            ref = r.trace_rays(rays)
            r.set_mesh_accel(pkg.ACCEL_BVH)
            got = r.trace_rays(rays)
            bad = np.unique(np.nonzero(got.view(np.uint8).reshape(len(rays), -1) != ref.view(np.uint8).reshape(len(rays), -1))[0])
            if len(bad):
                np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_mesh_bad_rays_{scenes}.npy"), rays[bad[:64]])
                print(f"scene {scenes}: {len(bad)} of {len(rays)} rays differ; first: {rays[bad[0]].tolist()} exhaustive {ref[bad[0]]} hierarchy {got[bad[0]]}", flush=True)
            scenes += 1; rays_total += len(rays); bad_total += len(bad); hits += int((ref["dist"] < 1e20).sum())
            if scenes % 20 == 0:
                print(f"{scenes} scenes, {rays_total} rays, {hits} hits, {bad_total} differ, {time.time() - t0:.0f} s", flush=True)
    print(f"mesh hierarchy fuzz: {scenes} scenes, {rays_total} rays ({hits} hits), {bad_total} differ -> {'FAILED' if bad_total else 'ok'}")
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
