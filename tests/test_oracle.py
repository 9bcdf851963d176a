"""CPU tests of the oracle: reference known-answer values, the spec'd RNG / sincos /
camera restated independently in numpy, analytic sanity (white furnace), invariances and the committed
golden fixtures.  No GPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
f32 = np.float32


def _normalize(v):
    v = np.asarray(v, dtype=f32)
    d = f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])
    return (v * (f32(1) / np.sqrt(f32(d)))).astype(f32)


def _sphere(oracle, radius, center):
    s = np.zeros(1, dtype=oracle.SPHERE_DTYPE)
    s[0]["radius"] = radius
    s[0]["center"] = center
    return s


# ------------------------------------------------------------------ reference KATs
def test_kat_mirror_sphere(oracle):
    k = KATS["mirror_sphere"]
    L = oracle.lib()
    s = _sphere(oracle, k["radius"], k["center"])
    o = np.array(k["origin"], dtype=f32)
    d = _normalize(np.array(k["center"], dtype=f32) - o)
    x = oracle.f3(0, 0, 0)
    n = oracle.f3(0, 0, 0)
    t = L.orc_intersect_analytic(s.ctypes.data_as(C.c_void_p), oracle.f3(*o), oracle.f3(*d), x)
    L.orc_make_hit_normal(s.ctypes.data_as(C.c_void_p), x, n)
    # SURVEY prints 9 significant digits = the exact binary32 values
    assert f32(t) == f32(k["t"])
    assert [f32(v) for v in x] == [f32(v) for v in k["x"]]
    assert [f32(v) for v in n] == [f32(v) for v in k["n"]]
    t = L.orc_intersect_analytic(s.ctypes.data_as(C.c_void_p), oracle.f3(*o), oracle.f3(*k["miss_dir"]), x)
    assert f32(t) == f32(k["miss_dist"])


def test_kat_left_wall_fp32_cancellation(oracle):
    k = KATS["left_wall"]
    L = oracle.lib()
    s = _sphere(oracle, k["radius"], k["center"])
    for case in k["cases"]:
        d = _normalize(case["dir_unnormalised"])
        x = oracle.f3(0, 0, 0)
        t = L.orc_intersect_analytic(s.ctypes.data_as(C.c_void_p), oracle.f3(*k["origin"]), oracle.f3(*d), x)
        assert f32(t) == f32(case["t"])
        assert float("%.6f" % x[0]) == case["x_x"]       # SURVEY prints x.x with 6 decimals


def test_kat_tri_intersect(oracle):
    L = oracle.lib()
    for k in KATS["tri_intersect"]:
        t, u, v = C.c_float(), C.c_float(), C.c_float()
        L.orc_tri_intersect(oracle.f3(*k["ro"]), oracle.f3(*k["rd"]), oracle.f3(*k["v0"]), oracle.f3(*k["v1"]),
                            oracle.f3(*k["v2"]), C.byref(t), C.byref(u), C.byref(v))
        assert f32(t.value) == f32(k["t"])
        if "u" in k:
            assert (u.value, v.value) == (k["u"], k["v"])


def test_struct_sizes_match_reference(oracle):
    # scene.h: float3 12 B, Ray 24 B; Material {float3,float3,Refl_t} 28 B -> carried inside the 48-B sphere POD
    assert oracle.SPHERE_DTYPE.itemsize == 48
    assert C.sizeof(oracle.OrcCamera) == 56
    assert KATS["sizeof"]["Material"] == 28 and KATS["sizeof"]["Ray"] == 24


def test_global_closest_hit_tiebreak_and_miss(oracle, pkg):
    """smallpt.cpp:59-65: ascending index, strict '<' => lowest index wins ties; no hit => -1."""
    L = oracle.lib()
    two = pkg.make_spheres([(1.0, (0, 0, -5), (0, 0, 0), (.5, .5, .5), 0)] * 2)
    dist = C.c_float()
    x, n = oracle.f3(0, 0, 0), oracle.f3(0, 0, 0)
    i = L.orc_intersect_global_spheres(two.ctypes.data_as(C.c_void_p), 2, oracle.f3(0, 0, 0), oracle.f3(0, 0, -1),
                                       C.byref(dist), x, n)
    assert i == 0 and dist.value == 4.0 and list(n) == [0.0, 0.0, 1.0]
    i = L.orc_intersect_global_spheres(two.ctypes.data_as(C.c_void_p), 2, oracle.f3(0, 0, 0), oracle.f3(0, 1, 0),
                                       C.byref(dist), x, n)
    assert i == -1
    # origin inside the sphere: the far root is taken (t = b + det), eps = 1e-4
    i = L.orc_intersect_global_spheres(two.ctypes.data_as(C.c_void_p), 2, oracle.f3(0, 0, -5), oracle.f3(0, 0, -1),
                                       C.byref(dist), x, n)
    assert i == 0 and dist.value == 1.0


# ------------------------------------------------------------------ D7 RNG, restated in numpy
def _mix32(x):
    with np.errstate(over="ignore"):
        x = np.asarray(x, dtype=np.uint32)
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x21f0aaad)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x735a2d97)
        x = x ^ (x >> np.uint32(15))
    return x


def _rng_bits(k0, k1, ctr):
    with np.errstate(over="ignore"):
        x = np.uint32(k0) + np.asarray(ctr, dtype=np.uint32) * np.uint32(0x9E3779B9)
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x21f0aaad)
        x = x + np.uint32(k1)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x735a2d97)
        x = x ^ (x >> np.uint32(15))
    return x


def test_rng_matches_numpy_restatement(oracle):
    L = oracle.lib()
    rs = np.random.RandomState(1)
    with np.errstate(over="ignore"):
        for _ in range(200):
            seed = int(rs.randint(0, 2**63 - 1, dtype=np.int64)) | (int(rs.randint(0, 2)) << 63)
            pix, smp, ctr = (int(v) for v in rs.randint(0, 2**32, size=3, dtype=np.uint64))
            k0, k1 = C.c_uint32(), C.c_uint32()
            L.orc_sample_keys(C.c_uint64(seed), pix, smp, C.byref(k0), C.byref(k1))
            s0 = _mix32(np.uint32(seed & 0xFFFFFFFF) + np.uint32(0x243F6A88))
            s1 = _mix32(np.uint32(seed >> 32) ^ s0 ^ np.uint32(0x85A308D3))
            p0 = _mix32(np.uint32(pix) + s0)
            p1 = _mix32(np.uint32(pix) ^ s1)
            e0 = _mix32(p0 ^ (np.uint32(smp) * np.uint32(0x9E3779B9)))
            e1 = _mix32(p1 + np.uint32(smp) * np.uint32(0x85EBCA6B))
            assert (k0.value, k1.value) == (int(e0), int(e1))
            bits = L.orc_rng_bits(k0.value, k1.value, ctr)
            assert bits == int(_rng_bits(e0, e1, ctr))
            u = L.orc_rng_uniform(k0.value, k1.value, ctr)
            assert u == float(f32(bits >> 8) * f32(2.0**-24)) and 0.0 <= u < 1.0


def test_rng_known_answers(oracle):
    """Frozen values: any change to the generator invalidates every golden image."""
    L = oracle.lib()
    assert L.orc_mix32(0) == 0 and L.orc_mix32(1) == int(_mix32(1))
    k0, k1 = C.c_uint32(), C.c_uint32()
    L.orc_sample_keys(C.c_uint64(0), 0, 0, C.byref(k0), C.byref(k1))
    got = [k0.value, k1.value, L.orc_rng_bits(k0.value, k1.value, 0), L.orc_rng_bits(k0.value, k1.value, (1 << 28) | 1)]
    assert got == [237207816, 2832002581, 1579120537, 3326102655], got


def test_rng_uniformity(oracle):
    k0, k1 = 0x12345678, 0x9ABCDEF0
    ctr = np.arange(1 << 20, dtype=np.uint32)
    u = (_rng_bits(k0, k1, ctr) >> np.uint32(8)).astype(np.float64) * 2.0**-24
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    hist = np.bincount((u * 64).astype(int), minlength=64)
    chi2 = ((hist - len(u) / 64) ** 2 / (len(u) / 64)).sum()
    assert chi2 < 130          # 63 dof: P(chi2 > 130) ~ 1e-6
    # consecutive dimensions of one path are uncorrelated
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5e-3
    # bijection in the counter for fixed keys: no repeated outputs
    assert len(np.unique(_rng_bits(k0, k1, ctr))) == len(ctr)


# ------------------------------------------------------------------ D17 sincos
def test_sincos2pi(oracle):
    L = oracle.lib()
    s, c = C.c_float(), C.c_float()
    exact = {0.0: (0.0, 1.0), 0.25: (1.0, 0.0), 0.5: (0.0, -1.0), 0.75: (-1.0, 0.0)}
    for u, (es, ec) in exact.items():
        L.orc_sincos2pi(u, C.byref(s), C.byref(c))
        assert abs(s.value - es) <= 1.2e-7 and abs(c.value - ec) <= 1.2e-7
    us = np.linspace(0, 1, 20001, endpoint=False).astype(f32)
    err = 0.0
    for u in us:
        L.orc_sincos2pi(float(u), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(2 * np.pi * float(u))), abs(c.value - np.cos(2 * np.pi * float(u))))
    assert err < 3e-7, err


def test_sincos_matches_numpy_restatement(oracle):
    L = oracle.lib()
    coef = [f32(float.fromhex(h)) for h in ("0x1.921fb4p+0", "-0x1.4abbb6p-1", "0x1.46676ep-4", "-0x1.3232fap-8", "0x1.3c4b2cp-13")]

    def sq(z):
        z2 = f32(z * z)
        p = coef[4]
        for c_ in (coef[3], coef[2], coef[1], coef[0]):
            p = f32(f32(p * z2) + c_)
        return f32(p * z)

    s, c = C.c_float(), C.c_float()
    for u in np.random.RandomState(3).randint(0, 1 << 24, 3000):
        uf = f32(u) * f32(2.0**-24)
        t = f32(4) * uf
        q = int(t)
        f = f32(t - f32(q))
        S, Cc = sq(f), sq(f32(f32(1) - f))
        es, ec = [(S, Cc), (Cc, -S), (-S, -Cc), (-Cc, S)][q]
        L.orc_sincos2pi(float(uf), C.byref(s), C.byref(c))
        assert (f32(s.value), f32(c.value)) == (f32(es), f32(ec))


# ------------------------------------------------------------------ camera (smallpt.cpp:277-279,327-333)
def test_camera_constants(oracle):
    cam = oracle.camera_smallpt(1024, 768)
    d = _normalize([0, f32(-0.042612), -1])
    assert list(cam.dir) == [float(v) for v in d]
    assert cam.cx[0] == float(f32(1024 * .5135 / 768)) and cam.cx[1] == 0 and cam.cx[2] == 0
    assert cam.push == 140.0 and list(cam.origin) == [50.0, 52.0, float(f32(295.6))]
    # cy = normalize(cross(cx, dir)) * .5135 points up (+y): row 0 is the bottom row (D14)
    assert cam.cy[1] > 0.5 and abs(np.linalg.norm(list(cam.cy)) - .5135) < 1e-6


def test_camera_ray_double_promotion(oracle):
    """smallpt.cpp:331-332 is evaluated in double (size_t + .5): restate in numpy float64/float32."""
    L = oracle.lib()
    w, h = 1024, 768
    cam = oracle.camera_smallpt(w, h)
    rs = np.random.RandomState(5)
    for _ in range(300):
        px, py = int(rs.randint(0, w)), int(rs.randint(0, h))
        sx, sy = int(rs.randint(0, 2)), int(rs.randint(0, 2))
        u1, u2 = (f32(v) * f32(2.0**-24) for v in rs.randint(0, 1 << 24, 2))

        def tent(u):
            r = f32(2) * u
            return f32(np.sqrt(r) - f32(1)) if r < 1 else f32(f32(1) - np.sqrt(f32(f32(2) - r)))

        dx, dy = tent(u1), tent(u2)
        ax = f32(((sx + .5 + float(dx)) / 2.0 + px) / w - .5)
        ay = f32(((sy + .5 + float(dy)) / 2.0 + py) / h - .5)
        cx, cy, cd, co = (np.array(list(v), dtype=f32) for v in (cam.cx, cam.cy, cam.dir, cam.origin))
        dd = ((cx * ax).astype(f32) + (cy * ay).astype(f32)).astype(f32) + cd
        eo = co + (dd * f32(140)).astype(f32)
        ed = _normalize(dd)
        o, d = oracle.f3(0, 0, 0), oracle.f3(0, 0, 0)
        L.orc_camera_ray(C.byref(cam), w, h, px, py, sx, sy, float(u1), float(u2), o, d)
        assert [f32(v) for v in o] == list(eo) and [f32(v) for v in d] == list(ed)


def test_camera_ray_pinhole_box_in_cell(oracle):
    """Renderer::render sampling (smallpt.cpp:745-760) + sampleRay (:626-641), all binary32, restated in numpy."""
    L = oracle.lib()
    w, h = 1280, 720                                   # main(), smallpt.cpp:844-845
    cam = oracle.OrcCamera()
    L.orc_camera_pinhole(oracle.f3(1, 0, 0), oracle.f3(0, 1, 0), oracle.f3(0, 0, -1), oracle.f3(0, -1, 0), C.c_float(1.0), C.byref(cam))
    rs = np.random.RandomState(7)
    for _ in range(300):
        px, py = int(rs.randint(0, w)), int(rs.randint(0, h))
        sx, sy = int(rs.randint(0, 2)), int(rs.randint(0, 2))
        u1, u2 = (f32(v) * f32(2.0**-24) for v in rs.randint(0, 1 << 24, 2))
        jx, jy = f32(f32(f32(sx) + u1) * f32(.5)), f32(f32(f32(sy) + u2) * f32(.5))
        fx, fy = f32(f32(.5) * f32(f32(f32(2) * jx) - f32(1))), f32(f32(.5) * f32(f32(f32(2) * jy) - f32(1)))
        nx = f32(f32(f32(f32(px) + f32(.5)) + fx) * f32(f32(1) / f32(w)))
        ny = f32(f32(f32(f32(py) + f32(.5)) + fy) * f32(f32(1) / f32(h)))
        cx_, cy_ = f32(f32(f32(2) * nx) - f32(1)), f32(f32(f32(2) * ny) - f32(1))
        dd = np.array([cx_, cy_, f32(-1.0)], dtype=f32)      # vx*cx + vy*cy + vz*near with the axis-aligned basis
        o, d = oracle.f3(0, 0, 0), oracle.f3(0, 0, 0)
        L.orc_camera_ray(C.byref(cam), w, h, px, py, sx, sy, float(u1), float(u2), o, d)
        assert [f32(v) for v in d] == list(_normalize(dd)) and list(o) == [0.0, -1.0, 0.0]


def test_to_int(oracle):
    L = oracle.lib()
    assert [L.orc_to_int(v) for v in (-1.0, 0.0, 0.5, 1.0, 7.0)] == [0, 0, 186, 255, 255]


# ------------------------------------------------------------------ image-level checks
def test_white_furnace(oracle, pkg):
    """Camera inside a closed diffuse emitter of albedo a: E[L] = e/(1-a) pins weights, the cosine
    sampling, and the Russian-roulette compensation (smallpt.cpp:188-198)."""
    a, e = 0.5, 1.0
    scene = pkg.make_spheres([(1000.0, (50, 52, 200), (e,) * 3, (a,) * 3, pkg.DIFF)])
    img, st = oracle.render(scene, 32, 32, 16, seed=3, normalise=True)
    mean = img.mean(axis=(0, 1))
    assert np.all(np.abs(mean - e / (1 - a)) < 0.02), mean
    assert st["max_depth_kills"] == 0


def test_probe_image_statistic(oracle, pkg):
    """Loose statistical cross-check against the patched-reference probe of SURVEY.md 8(c)."""
    k = KATS["probe_image_statistic"]
    img, st = oracle.render(pkg.cornell9(), k["w"], k["h"], k["spp"] // 4, seed=0, normalise=True)
    img8 = (np.clip(img, 0, 1) ** (1 / 2.2) * 255 + .5).astype(int)
    mean = img8.mean(axis=(0, 1))
    assert np.all(np.abs(mean / np.array(k["mean_rgb_8bit"]) - 1) < 0.10), mean
    left, right = img8[:, :20].mean(axis=(0, 1)), img8[:, -20:].mean(axis=(0, 1))
    assert left[0] > left[2] and right[2] > right[0]          # left column redder, right column bluer
    assert 7.0 < st["bounces"] / st["samples"] < 12.0


def test_zero_weight_cut_is_result_preserving(oracle, pkg):
    scene = pkg.cornell9()
    a, sa = oracle.render(scene, 24, 20, 2, seed=4)
    b, sb = oracle.render(scene, 24, 20, 2, seed=4, zero_cut=False)
    assert np.array_equal(a, b) and sb["bounces"] > sa["bounces"]


def test_partition_and_thread_invariance(oracle, pkg):
    scene = pkg.random_spheres(32, 7)
    full, sf = oracle.render(scene, 31, 23, 2, seed=11, normalise=True, threads=0)
    one, s1 = oracle.render(scene, 31, 23, 2, seed=11, normalise=True, threads=1)
    assert np.array_equal(full, one) and sf == s1
    parts = [oracle.render(scene, 31, 23, 2, seed=11, normalise=True, row_begin=b, row_count=c)[0]
             for b, c in ((0, 5), (5, 1), (6, 17))]
    assert np.array_equal(np.concatenate(parts), full)


def test_seed_changes_image_and_normalise_flag(oracle, pkg):
    scene = pkg.cornell9()
    a, _ = oracle.render(scene, 16, 12, 1, seed=0)
    b, _ = oracle.render(scene, 16, 12, 1, seed=1)
    c, _ = oracle.render(scene, 16, 12, 1, seed=1 << 40)
    assert not np.array_equal(a, b) and not np.array_equal(a, c)
    n, _ = oracle.render(scene, 16, 12, 1, seed=0, normalise=True)
    assert np.array_equal(n, (a * f32(1.0 / 4)).astype(f32))


def test_depth_cap(oracle, pkg):
    """D18: inside a perfect mirror Russian roulette never kills (p = 1); paths end only by the depth cap
    (or by a miss once the un-normalised mirror direction of smallpt.cpp:218 has drifted)."""
    scene = pkg.make_spheres([(1000.0, (50, 52, 200), (0, 0, 0), (1, 1, 1), pkg.SPEC)])
    img, st = oracle.render(scene, 4, 4, 1, seed=0)
    assert 0 < st["max_depth_kills"] <= st["samples"] == 64
    assert 4096 * st["max_depth_kills"] <= st["bounces"] <= 64 * 4096 and not img.any()


def test_empty_scene_and_bad_args(oracle, pkg):
    img, st = oracle.render(pkg.make_spheres([]), 8, 8, 1)
    assert not img.any() and st["bounces"] == st["samples"]
    with pytest.raises(RuntimeError):
        oracle.render(pkg.cornell9(), 8, 8, 1, row_begin=4, row_count=8)


@pytest.mark.parametrize("name", ["cornell9_32x24_s2_seed1", "cornell9_e12_40x30_s1_seed0_sum",
                                  "rand64_33x17_s3_seed9", "rand1024_24x18_s1_seed2", "cornell9_16x12_s128_seed5", "rand1024_12x8_s40_seed3"])
def test_golden_fixtures(oracle, pkg, name):
    import golden.make_golden as mg
    mk, w, h, samps, seed, norm = mg.CASES[name]
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    img, st = oracle.render(mk(), w, h, samps, seed=seed, normalise=norm)
    assert np.array_equal(img, g["image"]) and st["bounces"] == int(g["bounces"])


def _rel_l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / np.sqrt((b.astype(np.float64) ** 2).sum()))


def test_d9_block_sums_are_a_reordering_within_the_gate(oracle, pkg):
    """D9 splits a jitter cell's samples into NB = 1/2/4/8 blocks so that the kernels can schedule blocks.  The
    reference adds every emission event into the pixel (`outColor[pixelIdx] +=`, smallpt.cpp:179; `/= spp` :358-361) -- in a
    wavefront order nobody can reproduce, but its closest path-ordered forms are ONE accumulator per jitter cell (round 1's spec,
    = classic smallpt) and ONE accumulator per pixel.  Same samples, same events, different association of the float sums: the
    three images must agree far inside north_star's 1e-4 per-pixel relative L2 gate, at the headline sample count and at config 3's."""
    scene = pkg.cornell9()
    worst = 0.0
    for row, samps in ((100, 256), (384, 256), (700, 256), (300, 4096)):          # config 2 (1024 spp) x 3 rows, config 3 (16384 spp) x 1 row
        spec, st0 = oracle.render(scene, 1024, 768, samps, seed=0, normalise=True, row_begin=row, row_count=1)
        for alt in ("cells", "pixel"):
            img, st1 = oracle.render(scene, 1024, 768, samps, seed=0, normalise=True, row_begin=row, row_count=1, summation=alt)
            assert st1["bounces"] == st0["bounces"]                                  # the same paths
            rel = _rel_l2(img, spec)
            per_pixel = np.abs(img.astype(np.float64) - spec).max(axis=-1) / np.maximum(np.abs(spec).max(axis=-1), 1e-12)
            worst = max(worst, rel)
            assert rel <= 1e-4 and rel < 5e-6, (row, samps, alt, rel)               # measured ~1e-7
            assert float(per_pixel.max()) <= 1e-4, (row, samps, alt, float(per_pixel.max()))
        assert not np.array_equal(oracle.render(scene, 1024, 768, samps, seed=0, normalise=True, row_begin=row, row_count=1, summation="pixel")[0], spec) or samps < 32
    assert worst > 0.0                                                               # the alternatives do differ in the last bits: the test can fail
