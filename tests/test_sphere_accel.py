"""spt_set_sphere_accel: sphere tables above the pool kernel's 24 through a structure that is exhaustive-equivalent by construction --
the uniform grid of csrc/spt_grid.h (default) or the hierarchy whose boxes are inflated per ray by the rounding-error bound of
intersectAnalytic.  CPU: structure of the host builders; the grid's traversal arithmetic against the
exhaustive loop on 700 k rays under ASan/UBSan (tests/sanitize/grid_main.cpp).  GPU: images and bounce counts bit-identical to
the oracle's exhaustive loop (smallpt.cpp:54-70)."""
import os
import subprocess
import ctypes as C

import numpy as np
import pytest


def _selftest(pkg, spheres):
    lib = pkg.load_library()
    out, why = (C.c_uint32 * 4)(), C.create_string_buffer(256)
    rc = lib.spt_selftest_sphere_bvh(spheres.ctypes.data_as(C.c_void_p), len(spheres), C.byref(out), why, 256)
    return rc, list(out), why.value.decode()


def _selftest_grid(pkg, spheres, density=0):
    lib = pkg.load_library()
    out, why = (C.c_uint32 * 8)(), C.create_string_buffer(256)
    rc = lib.spt_selftest_sphere_grid(spheres.ctypes.data_as(C.c_void_p), len(spheres), density, C.byref(out), why, 256)
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


def test_sphere_grid_structure(pkg):
    cases = {"one": pkg.make_spheres([(1.0, (0, 0, 0), (0, 0, 0), (.5, .5, .5), 0)]), "config 5": pkg.random_spheres(1024, 1024),
             "cluster 300": _cluster_scene(pkg, 300, 1), "no huge": _cluster_scene(pkg, 100, 2, huge=False), "4096": _cluster_scene(pkg, 4096, 3),
             "identical": pkg.make_spheres([(1.0, (1, 2, 3), (0, 0, 0), (.5, .5, .5), 0)] * 50)}
    for name, sc in cases.items():
        for density in (0, 2, 40):
            rc, (dx, dy, dz, refs, always, nbytes, usable, _), why = _selftest_grid(pkg, sc, density)
            assert rc == 0 and usable == 1, (name, density, rc, why)
            assert 1 <= dx <= 128 and 1 <= dy <= 128 and 1 <= dz <= 128 and always <= 1024, (name, dx, dy, dz, always)
            assert nbytes + 16 * max(1, len(sc)) <= 150 * 1024, (name, nbytes)          # tables + sphere records fit one CU's LDS
            if name == "config 5":
                assert always == 7                   # the six walls and the light are tested for every ray
    bad = pkg.make_spheres([(1.0, (np.nan, 0, 0), (0, 0, 0), (.5, .5, .5), 0)] * 30)
    rc, _, why = _selftest_grid(pkg, bad)
    assert rc == 1 and "non-finite" in why


def test_sphere_grid_walk_equals_exhaustive_on_cpu(tmp_path):
    """The traversal arithmetic the kernel uses (csrc/spt_grid.h) + the builder against the exhaustive loop, on the CPU: random,
    grazing, axis-parallel, on-cell-face, almost-zero-component, drifted-length and far-origin rays over six kinds of tables;
    an under-registered grid must fail (negative control)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "grid_main"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(root, "tests", "sanitize", "grid_main.cpp"),
                           os.path.join(root, "optix-test-smallpt_amd", "csrc", "spt_grid.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "mismatches 0," in r.stdout and "grid harness ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("lane_owned", [False, True], ids=["path pools", "lane-owned"])
def test_sphere_grid_images_equal_oracle(pkg, renderer, oracle, lane_owned):
    """Default mode: tables above 24 spheres run a grid kernel -- wave-private path pools with walker lanes (spt_gpool.hip, the default)
    or lanes that own their path (spt_grid.hip) --; image and bounce count equal the oracle's exhaustive loop."""
    scenes = [("cluster 25", _cluster_scene(pkg, 25, 5)), ("cluster 100", _cluster_scene(pkg, 100, 6)), ("cluster 600", _cluster_scene(pkg, 600, 7)),
              ("open 257", _cluster_scene(pkg, 257, 8, huge=False)), ("config 5", pkg.random_spheres(1024, 1024)), ("random 4096", pkg.random_spheres(4096, 7)),
              ("identical 50", pkg.make_spheres([(1.0, (50, 40, 80), (1, 1, 1), (.5, .5, .5), 0)] * 50)),
              # some of these 4096 are concentric with the wall spheres (centres 1e5 away): the extent is 1e5 long, nearly everything
              # shares a cell, the grid declines (spt_api.cpp build_sphere_grid_tables) and from 1024 spheres on the hierarchy takes over
              ("cluster 4096", _cluster_scene(pkg, 4096, 9)),
              # sphere records beyond one CU's LDS: the grid with its tables in global memory (spt_grid.hip GLOBAL_TABLES) up to 24 576 spheres,
              # the hierarchy beyond
              ("random 12000", pkg.random_spheres(12000, 11)), ("random 30000", pkg.random_spheres(30000, 12))]
    name_of = "grid" if lane_owned else "gpool"
    # 4096 sphere records + their grid leave no room for the pools' begun walks: the lane-owned kernel keeps such tables
    expect = {"cluster 4096": "sbvh", "random 12000": "grid", "random 30000": "sbvh", "random 4096": "grid"}
    try:
        renderer.set_grid_pools(lane_owned=lane_owned)
        for name, sc in scenes:
            w, h, samps, seed = (40, 30, 2, 3) if len(sc) < 600 else (32, 20, 1, 4)
            renderer.set_scene(sc)
            img, st = renderer.render(w, h, samps, seed=seed)
            assert renderer.last_kernel() == expect.get(name, name_of), name
            ref, rst = oracle.render(sc, w, h, samps, seed=seed)
            assert np.array_equal(img, ref), (name, int((img != ref).any(axis=-1).sum()))
            assert st["bounces"] == rst["bounces"] and st["max_depth_kills"] == rst["max_depth_kills"], name
            for cam in (pkg.pinhole_camera(org=(50, 45, 160), vz=(0, 0, -1)),               # the viewer's camera, inside the scene
                        pkg.pinhole_camera(org=(50, 45, 2000), vz=(0, 0, -1))):             # far outside: rays beyond the grid's error bound take the exhaustive loop
                img, st = renderer.render(w, h, samps, seed=seed + 1, camera=cam)
                ref, rst = oracle.render(sc, w, h, samps, seed=seed + 1, camera=cam)
                assert np.array_equal(img, ref) and st["bounces"] == rst["bounces"], (name, "pinhole")
        # several D9 sample blocks per cell, a ragged size and the normalised output
        sc = pkg.random_spheres(1024, 1024)
        renderer.set_scene(sc)
        img, st = renderer.render(9, 7, 70, seed=11, normalise=True)
        ref, rst = oracle.render(sc, 9, 7, 70, seed=11, normalise=True)
        assert renderer.last_kernel() == name_of and np.array_equal(img, ref) and st["bounces"] == rst["bounces"]
        ref2, rst2 = oracle.render(sc, 32, 20, 1, seed=4)
        if lane_owned:
            # the walk-phase policy (when a wave leaves its walk phase) and the resolution never change the image
            for variant in (1 << 16, 5 << 16, 65 << 16, (1 << 16) | (3 << 24), 40 << 24):
                renderer.set_tuning(0, variant)
                renderer.set_scene(sc)
                img2, st2 = renderer.render(32, 20, 1, seed=4)
                assert np.array_equal(img2, ref2) and st2["bounces"] == rst2["bounces"], hex(variant)
        else:
            # the pool geometry (slots and begun walks per wave, when walkers are exchanged, the smallest batch), the workgroup size and
            # the resolution never change the image
            for geom, variant in (((64, 48, 1, 1, 1), 0), ((256, 96, 64, 64, 16), 0), ((192, 52, 8, 16, 2), 3 << 24), ((0, 0, 0, 0, 0), 3 << 13), ((128, 64, 40, 8, 0), 10 << 24),
                                  ((0, 0, 0, 0, 0), 40 << 24)):          # 40 cells per sphere: more references than the walkers' 16-bit addresses reach -> lane-owned kernel
                renderer.set_grid_pools(False, *geom)
                renderer.set_tuning(0, variant)
                renderer.set_scene(sc)
                img2, st2 = renderer.render(32, 20, 1, seed=4)
                assert renderer.last_kernel() == ("grid" if variant == 40 << 24 else "gpool"), (geom, hex(variant))
                assert np.array_equal(img2, ref2) and st2["bounces"] == rst2["bounces"], (geom, hex(variant))
    finally:
        renderer.set_tuning(0, 0)
        renderer.set_grid_pools()
    renderer.set_scene(pkg.cornell9())
    renderer.render(8, 8, 1)
    assert renderer.last_kernel() == "pool"


@pytest.mark.gpu
def test_deep_paths_in_a_large_table_finish_quickly_in_every_mode(pkg, renderer, oracle):
    """Case 1369 of the parity fuzzer's default stream at its original size (tests/fuzz_recipe.py: 1500 spheres, 69 of them colour (1,1,1),
    298 huge; 24 x 12 pixels, 70 samples per jitter cell): 288 paths run to the depth cap of 4096 because some pixels look into closed white
    mirror / glass balls, where the roulette never kills.  Round 3 needed 2.7 ... 16 s for it -- all sample blocks of such a pixel sat in one
    wave, which dragged them through its phases long after the others had finished -- and a heavier case of this kind tripped the
    60-second kernel watchdog.  Since round 4 a pixel's blocks are dealt to different waves (spt_device.h deal_task) and a wave with a few
    rays left answers them with all its lanes: 0.3 ... 0.6 s in every mode (profiles/r04_fuzz_deep_cases.txt).  Bit-exact image, equal
    bounce and depth-cap counts, and a time bound a regression of either mechanism would break."""
    from fuzz_recipe import draw_case
    rs = np.random.RandomState(1234)
    for _ in range(1370):
        case = draw_case(rs, pkg)
    assert (case["n"], case["w"], case["h"], case["samps"], case["white"]) == (1500, 24, 12, 70, 69)
    sc = pkg.make_spheres(case["rows"])
    ref, rst = oracle.render(sc, case["w"], case["h"], case["samps"], seed=case["seed"], normalise=case["norm"])
    assert rst["max_depth_kills"] == 288
    try:
        for mode, accel, lane_owned, kernel in (("path pools", pkg.ACCEL_GRID, False, "gpool"), ("lane-owned", pkg.ACCEL_GRID, True, "grid"),
                                                ("hierarchy", pkg.ACCEL_BVH, False, "sbvh"), ("exhaustive", pkg.ACCEL_EXHAUSTIVE, False, "mega")):
            renderer.set_grid_pools(lane_owned=lane_owned)
            renderer.set_sphere_accel(accel)
            renderer.set_scene(sc)
            img, st = renderer.render(case["w"], case["h"], case["samps"], seed=case["seed"], normalise=case["norm"])
            assert renderer.last_kernel() == kernel, mode
            assert np.array_equal(img, ref, equal_nan=True), mode
            assert st["bounces"] == rst["bounces"] and st["max_depth_kills"] == 288, mode
            assert st["kernel_ms"] < 2000.0, (mode, st["kernel_ms"])
    finally:
        renderer.set_grid_pools()
        renderer.set_sphere_accel(pkg.ACCEL_GRID)


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
        renderer.set_sphere_accel(pkg.ACCEL_GRID)            # the default
        renderer.set_scene(pkg.cornell9())
