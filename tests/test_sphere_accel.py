"""spt_set_sphere_accel: sphere tables above the pool kernel's 24 through a hierarchy that is exhaustive-equivalent by construction
(every box is inflated per ray by twice the rounding-error bound of intersectAnalytic).  CPU: structure of
the host builder.  GPU: images and bounce counts bit-identical to the oracle's exhaustive loop (smallpt.cpp:54-70)."""
import ctypes as C

import numpy as np
import pytest


def _selftest(pkg, spheres):
    lib = pkg.load_library()
    out, why = (C.c_uint32 * 4)(), C.create_string_buffer(256)
    rc = lib.spt_selftest_sphere_bvh(spheres.ctypes.data_as(C.c_void_p), len(spheres), C.byref(out), why, 256)
    return rc, list(out), why.value.decode()


def _cluster_scene(pkg, n, seed, huge=True):
    """n spheres: small ones of very different sizes, overlapping, some concentric / coincident, all three materials, a few
    emitters; optionally closed in by the Cornell walls and light (radii 1e5 / 600: the always-tested list)."""
    rs = np.random.RandomState(seed)
    rows = []
    if huge:
        base = pkg.cornell9()
        for i in (0, 1, 2, 3, 4, 5, 8):
            b = base[i]
            rows.append((float(b["radius"]), tuple(b["center"]), tuple(b["emission"]), tuple(b["color"]), int(b["refl"])))
    while len(rows) < n:
        r = float(10 ** rs.uniform(-1.5, 0.8))
        c = (float(rs.uniform(5, 95)), float(rs.uniform(3, 75)), float(rs.uniform(10, 150)))
        if rows and rs.rand() < 0.05:
            c = tuple(float(v) for v in rows[rs.randint(len(rows))][1])          # concentric with an earlier sphere
        e = (0, 0, 0) if rs.rand() < 0.9 else tuple(float(v) for v in rs.uniform(0, 4, 3))
        col = tuple(float(v) for v in rs.uniform(.2, .95, 3))
        rows.append((r, c, e, col, int(rs.choice([0, 0, 0, 1, 2]))))
    return pkg.make_spheres(rows)


def test_sphere_hierarchy_structure(pkg):
    cases = {"empty": pkg.make_spheres([]), "one": pkg.make_spheres([(1.0, (0, 0, 0), (0, 0, 0), (.5, .5, .5), 0)]),
             "cornell9": pkg.cornell9(), "config 5": pkg.random_spheres(1024, 1024), "cluster 300": _cluster_scene(pkg, 300, 1),
             "no huge": _cluster_scene(pkg, 100, 2, huge=False), "4096": _cluster_scene(pkg, 4096, 3),
             "identical": pkg.make_spheres([(1.0, (1, 2, 3), (0, 0, 0), (.5, .5, .5), 0)] * 50)}
    for name, sc in cases.items():
        rc, (nodes, leaves, depth, always), why = _selftest(pkg, sc)
        assert rc == 0, (name, rc, why)
        assert depth <= 32 and always <= 32, (name, depth, always)
        if name == "config 5":
            assert always == 7                       # the six walls and the light
        if name == "identical":
            assert always == 0                       # nothing is more than 16 x the median radius
    bad = pkg.make_spheres([(1.0, (np.nan, 0, 0), (0, 0, 0), (.5, .5, .5), 0)] * 30)
    rc, _, why = _selftest(pkg, bad)
    assert rc == 1 and "non-finite" in why


@pytest.mark.gpu
def test_sphere_hierarchy_images_equal_oracle(pkg, renderer, oracle):
    scenes = [("cluster 25", _cluster_scene(pkg, 25, 5)), ("cluster 100", _cluster_scene(pkg, 100, 6)), ("cluster 600", _cluster_scene(pkg, 600, 7)),
              ("open 257", _cluster_scene(pkg, 257, 8, huge=False)), ("config 5", pkg.random_spheres(1024, 1024))]
    try:
        renderer.set_sphere_accel(pkg.ACCEL_BVH)
        for name, sc in scenes:
            w, h, samps, seed = (40, 30, 2, 3) if len(sc) < 600 else (32, 20, 1, 4)
            renderer.set_scene(sc)
            img, st = renderer.render(w, h, samps, seed=seed)
            assert renderer.last_kernel() == "sbvh", name
            ref, rst = oracle.render(sc, w, h, samps, seed=seed)
            assert np.array_equal(img, ref), (name, int((img != ref).any(axis=-1).sum()))
            assert st["bounces"] == rst["bounces"] and st["max_depth_kills"] == rst["max_depth_kills"], name
            cam = pkg.pinhole_camera(org=(50, 45, 160), vz=(0, 0, -1))                       # the viewer's camera, inside the scene
            img, st = renderer.render(w, h, samps, seed=seed + 1, camera=cam)
            ref, rst = oracle.render(sc, w, h, samps, seed=seed + 1, camera=cam)
            assert np.array_equal(img, ref) and st["bounces"] == rst["bounces"], (name, "pinhole")
        # tables the pool kernel handles stay with it; switching the mode off returns to the megakernel
        renderer.set_scene(pkg.cornell9())
        renderer.render(8, 8, 1)
        assert renderer.last_kernel() == "pool"
        renderer.set_sphere_accel(pkg.ACCEL_EXHAUSTIVE)
        renderer.set_scene(scenes[1][1])
        renderer.render(8, 8, 1)
        assert renderer.last_kernel() == "mega"
    finally:
        renderer.set_sphere_accel(pkg.ACCEL_EXHAUSTIVE)
        renderer.set_scene(pkg.cornell9())
