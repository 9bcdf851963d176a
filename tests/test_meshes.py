"""f4: the reference's triangle-mesh primitives (TriMesh, makeSphereTriMesh scene.cpp:3-48, triIntersect :52-70,
intersect(mesh) :95-116, makeHit(mesh) :73-93) and the Intersector seam (addTriangleMesh/build/traceRays,
smallpt.cpp:427-473).  CPU part: the oracle restatement against the SURVEY.md 8(c) known answers and the host
tessellator against the oracle's.  GPU part: spt_trace_rays and the mesh path tracer bit-exact against the oracle."""
import numpy as np
import pytest


def test_make_sphere_trimesh_counts_and_host_equals_oracle(pkg, oracle):
    """KAT: makeSphereTriMesh(0, 1) => 2145 vertices, 2145 normals, 12288 indices, 4096 triangles."""
    m = pkg.make_sphere_trimesh((0, 0, 0), 1.0)
    assert m.positions.shape == (2145, 3) and m.normals.shape == (2145, 3) and m.indices.size == 12288 and m.triangle_count == 4096
    pos, nor, idx = oracle.make_sphere_trimesh((0, 0, 0), 1.0)
    assert np.array_equal(m.positions, pos) and np.array_equal(m.normals, nor) and np.array_equal(m.indices, idx)
    assert np.abs(np.linalg.norm(m.normals, axis=1) - 1).max() < 1e-6 and int(m.indices.max()) == 2144
    m2 = pkg.make_sphere_trimesh((27, 16.5, 47), 16.5, 8)
    pos, nor, idx = oracle.make_sphere_trimesh((27, 16.5, 47), 16.5, 8)
    assert np.array_equal(m2.positions, pos) and np.array_equal(m2.indices, idx) and m2.triangle_count == 4 * 8 * 8
    assert np.array_equal(m2.positions, (np.float32(16.5) * nor + np.array((27, 16.5, 47), dtype=np.float32)))     # origin + radius * coords, :25


def test_oracle_trace_rays_single_triangle_kats(pkg, oracle):
    """KAT: triangle of smallpt.cpp:826, ro=(0,0,0), rd=(0,0,-1) => t=2, u=.25, v=.5; ro=(2,0,0) => miss 1e20.
    makeHit: x = (1-u-v)A + uB + vC, n likewise from the vertex normals (1,0,0),(0,1,0),(0,0,1) => n = (w, u, v)."""
    meshes, _ = pkg.single_triangle_scene()
    hits = oracle.trace_rays(meshes, [[0, 0, 0, 0, 0, -1], [2, 0, 0, 0, 0, -1], [0, 0, -4, 0, 0, 1], [0, 0, 0, 0, 0, 1]])
    h = hits[0]
    assert h["dist"] == 2.0 and tuple(h["uv"]) == (0.25, 0.5) and h["instId"] == 0 and h["triId"] == 0
    assert tuple(h["x"]) == (0.0, 0.0, -2.0) and tuple(h["n"]) == (0.25, 0.25, 0.5)       # the barycentric convention of :544-546
    assert hits[1]["dist"] == np.float32(1e20) and hits[3]["dist"] == np.float32(1e20)    # beside / behind
    assert hits[2]["dist"] == 2.0                                                         # hit from the back side: no culling


def _mesh_scene(pkg):
    """Tessellated counterparts of a small Cornell-like set-up: floor + light as coarse sphere meshes, a diffuse, a mirror
    and a glass ball, and the single triangle; few enough triangles for the oracle's brute force."""
    S = pkg.make_sphere_trimesh
    meshes = [S((0, -1e3 - 2, -6), 1e3, 24), S((0, 40, -6), 30.0, 8), S((-2.2, -1, -6), 1.0, 8), S((0, -1, -7.5), 1.0, 8),
              S((2.2, -1, -5.5), 1.0, 8), pkg.single_triangle_scene()[0][0]]
    mats = [((0, 0, 0), (.7, .7, .7), pkg.DIFF), ((3, 3, 3), (0, 0, 0), pkg.DIFF), ((0, 0, 0), (.75, .25, .25), pkg.DIFF),
            ((0, 0, 0), (.999, .999, .999), pkg.SPEC), ((0, 0, 0), (.999, .999, .999), pkg.REFR), ((1, 0, 0), (0, 0, 0), pkg.DIFF)]
    return meshes, mats


@pytest.mark.gpu
def test_trace_rays_matches_oracle(pkg, renderer, oracle):
    """The Intersector seam on the GPU: Hit{dist, instId, triId, x, n, uv} of every ray equals the oracle's, bit for bit."""
    meshes, mats = _mesh_scene(pkg)
    renderer.set_meshes(meshes, mats)
    rs = np.random.RandomState(7)
    n = 4000
    o = np.tile(np.array([0, -1, 0], dtype=np.float32), (n, 1)) + rs.uniform(-.5, .5, (n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32) + np.array([0, 0, -2], dtype=np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([np.concatenate([o, d], axis=1),
                           np.array([[0, 0, 0, 0, 0, -1], [2, 0, 0, 0, 0, -1], [0, -1, 0, 0, 1, 0], [0, -1, 0, 0, 0, 0]], dtype=np.float32)])
    got = renderer.trace_rays(rays)
    ref = oracle.trace_rays(meshes, rays)
    assert got.tobytes() == ref.tobytes()
    assert (got["dist"] < 1e20).sum() > n // 2 and len(set(got["instId"][got["dist"] < 1e20])) >= 4
    k = len(rays) - 4
    assert got[k]["dist"] == 2.0 and tuple(got[k]["uv"]) == (0.25, 0.5) and got[k]["instId"] == 5      # the KAT through the GPU
    # the same query on device buffers (spt_trace_rays_device), on the caller's stream, in both closest-hit modes
    import torch
    rays_t = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    for accel in (pkg.ACCEL_EXHAUSTIVE, pkg.ACCEL_BVH):
        renderer.set_mesh_accel(accel)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            hits_t = renderer.trace_rays_device(rays_t, stream=side)
        side.synchronize()
        assert hits_t.cpu().numpy().tobytes() == renderer.trace_rays(rays).tobytes()
        if accel == pkg.ACCEL_EXHAUSTIVE:
            assert hits_t.cpu().numpy().tobytes() == ref.tobytes()
    renderer.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
    renderer.set_scene(pkg.cornell9())
    with pytest.raises(pkg.SptError, match="no mesh scene"):
        renderer.trace_rays(rays[:2])
    with pytest.raises(pkg.SptError, match="no mesh scene"):
        renderer.trace_rays_device(rays_t)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,samps,seed,pinhole", [(32, 18, 1, 0, True), (24, 20, 2, 3, False)])
def test_mesh_path_tracer_matches_oracle(pkg, renderer, oracle, w, h, samps, seed, pinhole):
    """The path tracer over triangle meshes (interpolated un-normalised normals, every material, glass split) against the
    oracle's render over the same meshes."""
    meshes, mats = _mesh_scene(pkg)
    renderer.set_meshes(meshes, mats)
    cam = pkg.pinhole_camera() if pinhole else None
    img, st = renderer.render(w, h, samps, seed=seed, normalise=not pinhole, camera=cam)
    ref, rst = oracle.render_meshes(meshes, mats, w, h, samps, seed=seed, normalise=not pinhole, camera=cam)
    assert np.array_equal(img, ref), f"{int((img != ref).any(axis=-1).sum())} pixels differ"
    assert st["bounces"] == rst["bounces"] and st["samples"] == rst["samples"]
    if pinhole:
        assert img.max() > 0
    renderer.set_scene(pkg.cornell9())            # back to spheres for the other tests of the session


@pytest.mark.gpu
def test_single_triangle_scene_of_main(pkg, renderer, oracle):
    """main()'s SingleTriangleScene (smallpt.cpp:818-838) under the viewer camera at 1280x720 / 8, one frame."""
    meshes, mats = pkg.single_triangle_scene()
    renderer.set_meshes(meshes, mats)
    cam = pkg.pinhole_camera()
    img, st = renderer.render(160, 90, 1, seed=0, normalise=False, camera=cam)
    ref, rst = oracle.render_meshes(meshes, mats, 160, 90, 1, seed=0, normalise=False, camera=cam)
    assert np.array_equal(img, ref) and st["bounces"] == rst["bounces"]
    assert img[:, :, 0].max() > 0 and not img[:, :, 1:].any()        # the triangle emits pure red
    renderer.set_scene(pkg.cornell9())


def _mixed_mesh_file(pkg, tmp_path):
    tri, trimat = pkg.single_triangle_scene()
    meshes = [pkg.make_sphere_trimesh((0, -1, -6), 1.0, 8), pkg.make_sphere_trimesh((0, 40, -6), 30.0, 6), tri[0]]
    mats = [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((3, 3, 3), (0, 0, 0), pkg.DIFF), trimat[0]]
    gens = [((0, -1, -6), 1.0, 8), ((0, 40, -6), 30.0, 6), None]
    p = tmp_path / "meshes.json"
    p.write_text(pkg.meshes_to_json(meshes, mats, gens))
    return meshes, mats, p


def test_mesh_scene_json_through_the_cpp_loader(pkg, tmp_path):
    """The "meshes" section of the scene file: written by Python, read by the C++ loader (generator entries are
    re-tessellated there), written back by C++, read by Python -- buffers and materials survive bit for bit."""
    import subprocess
    import os
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "optix-test-smallpt_amd", "host", "smallpt_mi355x")
    meshes, mats, p = _mixed_mesh_file(pkg, tmp_path)
    back = tmp_path / "back.json"
    r = subprocess.run([cli, "--scene", str(p), "--dump-scene", str(back), "--parse-only"], capture_output=True)
    assert r.returncode == 0, r.stderr
    m2, mats2 = pkg.meshes_from_json(back.read_text())
    assert len(m2) == 3 and len(mats2) == 3
    for a, b in zip(meshes, m2):
        assert np.array_equal(a.positions, b.positions) and np.array_equal(a.normals, b.normals) and np.array_equal(a.indices, b.indices)
    for (e, c, rf), (e2, c2, rf2) in zip(mats, mats2):
        assert tuple(np.float32(e)) == tuple(np.float32(e2)) and tuple(np.float32(c)) == tuple(np.float32(c2)) and rf == rf2
    bad = tmp_path / "bad.json"
    bad.write_text('{"meshes": [{"positions": [[0,0,0]], "normals": [[0,0,1]], "indices": [[0,1,2]], "emission": [0,0,0], "color": [1,1,1], "refl": "DIFF"}]}')
    assert subprocess.run([cli, "--scene", str(bad), "--parse-only"], capture_output=True).returncode == 1      # index out of range


@pytest.mark.gpu
def test_cpp_cli_renders_mesh_scenes(pkg, oracle, tmp_path):
    """The cpuRender-shaped CLI over mesh scenes: a JSON file with generated and explicit meshes, and the reference's own
    shipped global table (two tessellated spheres, smallpt.cpp:31-34) -- PPMs equal the oracle's render of the same meshes."""
    import os
    import subprocess
    from test_gpu_parity import expected_ppm
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "optix-test-smallpt_amd", "host", "smallpt_mi355x")
    out = tmp_path / "m.ppm"
    r = subprocess.run([cli, "4", "--scene", "shipped-meshes", "--size", "24x16", "--seed", "2", "--out", str(out)], capture_output=True)
    assert r.returncode == 0, r.stderr
    meshes = [pkg.make_sphere_trimesh((50, 40.8, 81.6), 10.0), pkg.make_sphere_trimesh((50, 681.6 - .27, 81.6), 600.0)]
    mats = [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((1, 1, 1), (0, 0, 0), pkg.DIFF)]
    ref, _ = oracle.render_meshes(meshes, mats, 24, 16, 1, seed=2, normalise=True)
    assert out.read_bytes() == expected_ppm(oracle, ref) and ref.max() > 0
    out2 = tmp_path / "m_bvh.ppm"                                   # the same through the hierarchy (OptixIntersector's role)
    r = subprocess.run([cli, "4", "--scene", "shipped-meshes", "--accel", "bvh", "--size", "24x16", "--seed", "2", "--out", str(out2)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert out2.read_bytes() == out.read_bytes()


# ---- SPT_ACCEL_BVH: the optional hierarchy (the OptiX Prime model's role, smallpt.cpp:475-603) ----
def _selftest_bvh(pkg, meshes):
    import ctypes as C
    lib = pkg.load_library()
    ms = (pkg.SptMesh * max(1, len(meshes)))()
    for i, m in enumerate(meshes):
        ms[i].positions, ms[i].normals, ms[i].indices = m.positions.ctypes.data, m.normals.ctypes.data, m.indices.ctypes.data
        ms[i].nverts, ms[i].ntris = len(m.positions), len(m.indices)
    out, why = (C.c_uint32 * 4)(), C.create_string_buffer(256)
    rc = lib.spt_selftest_bvh(ms, len(meshes), C.byref(out), why, 256)
    return rc, list(out), why.value.decode()


def _soup(pkg, n, seed, flat=False):
    """n random triangles: sizes over three decades, positions over two; flat=True puts them all into one plane (every
    centroid split degenerates on one axis, many coincide)."""
    rs = np.random.RandomState(seed)
    c = rs.uniform(-50, 50, (n, 1, 3)) * (10 ** rs.uniform(-1, 0, (n, 1, 1)))
    v = c + rs.normal(size=(n, 3, 3)) * (10 ** rs.uniform(-2, 1, (n, 1, 1)))
    if flat:
        v[:, :, 1] = 3.0
    pos = v.reshape(-1, 3).astype(np.float32)
    nor = np.tile(np.array([0, 1, 0], dtype=np.float32), (len(pos), 1))
    return pkg.TriMesh(pos, nor, np.arange(3 * n, dtype=np.uint32).reshape(n, 3))


def test_bvh_builder_structure(pkg):
    """Host-side invariants of the hierarchy (no GPU): every triangle in exactly one leaf of <= 12, every box contains the
    padded triangles below it, no reference deeper than the traversal's 32-entry stack; degenerate inputs included."""
    S = pkg.make_sphere_trimesh
    tri = pkg.single_triangle_scene()[0][0]
    same = pkg.TriMesh(np.tile(tri.positions, (300, 1)), np.tile(tri.normals, (300, 1)), np.arange(900, dtype=np.uint32).reshape(300, 3))
    cases = {"empty": [], "one": [tri], "three": [tri, tri, tri], "five": [tri] * 5, "300 identical": [same],
             "shipped": [S((-1, 0, -4), 1.0), S((1.5, 0, -5), 1.0)], "mixed sizes": [S((0, -1e3 - 2, -6), 1e3, 24), S((0, 0, 0), 0.01, 8), tri],
             "soup": [_soup(pkg, 20000, 1)], "flat soup": [_soup(pkg, 5000, 2, flat=True)]}
    for name, meshes in cases.items():
        rc, (nodes, leaves, depth, ntris), why = _selftest_bvh(pkg, meshes)       # ntris = REGULAR triangles: the spatial hierarchy and the plane tree (spt_tribvh.h)
        assert rc == 0, (name, rc, why)
        total = sum(m.triangle_count for m in meshes)
        assert ntris <= total and depth <= 32 and nodes >= 1, (name, nodes, leaves, depth, ntris)
        # not regular = thin (angle at v0 below 1/32: the line tree) or with an edge of length zero (never hit, in no structure): only
        # makeSphereTriMesh's pole needles (two pole rows x 2L needles per sphere, scene.cpp:13-27; L = 32 / 24 / 8) -- the 2L needles of the
        # TOP row (v0 on the last ring, both edges to the pole) always; a bottom-row needle (v0 and v1 both pole copies) whose first edge
        # is not zero but an ulp or two at a usable angle is a regular triangle (spt_tribvh.h: only the angle at v0 matters)
        if "soup" in name:
            assert total - ntris <= total // 20, (name, total - ntris)            # a random triangle rarely has an angle below 1/32 at v0
        else:
            needles = {"shipped": 2 * 2 * 64, "mixed sizes": 2 * 48 + 2 * 16}.get(name, 0)
            assert needles // 2 <= total - ntris <= needles, (name, total - ntris)
        if ntris > 12:                                       # leaves of <= 12 triangles (kBvhLeafTris)
            assert leaves >= (ntris + 11) // 12 and nodes == leaves - 1, (name, nodes, leaves)
    rc, _, why = _selftest_bvh(pkg, [pkg.TriMesh(np.array([[0, 0, 0], [1, 0, 0], [np.inf, 1, 0]], dtype=np.float32), tri.normals, tri.indices)])
    assert rc == 1 and "non-finite" in why


def _adversarial_rays(meshes, rs, n_random):
    """Random rays plus the ones a padded hierarchy could get wrong: aimed exactly at vertices, edge midpoints and
    centroids (from outside and from a vertex itself), axis-parallel, grazing along triangle planes, origins on surfaces."""
    pos = np.concatenate([m.positions for m in meshes]).astype(np.float64)
    tri = np.concatenate([m.positions[m.indices.reshape(-1, 3)] for m in meshes]).astype(np.float64)     # (T, 3, 3)
    lo, hi = pos.min(0), pos.max(0)
    ext = np.maximum(hi - lo, 1e-3)
    rays = []
    o = rs.uniform(lo - ext, hi + ext, (n_random, 3)); d = rs.normal(size=(n_random, 3))
    rays.append(np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1))
    k = min(len(tri), 4000)
    pick = tri[rs.choice(len(tri), k, replace=len(tri) < k)]
    eye = rs.uniform(lo - ext, hi + ext, (k, 3))
    targets = [pick[:, 0], 0.5 * (pick[:, 0] + pick[:, 1]), pick.mean(1), pick[:, 2]]
    for t in targets:
        d = t - eye
        rays.append(np.concatenate([eye, d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)], axis=1))
    d = pick[:, 1] - pick[:, 0]                                                       # along an edge, starting on / before the vertex
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    rays.append(np.concatenate([pick[:, 0], d], axis=1))
    rays.append(np.concatenate([pick[:, 0] - 3 * d, d], axis=1))
    n = np.cross(pick[:, 1] - pick[:, 0], pick[:, 2] - pick[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    for eps in (1e-2, 1e-4, 1e-6, 0.0):                                               # grazing: in the triangle's plane, tilted by eps
        g = d + eps * n
        rays.append(np.concatenate([pick.mean(1) - 2 * g, g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)], axis=1))
    for ax in range(3):                                                               # axis-parallel (zero direction components)
        for sgn in (1.0, -1.0):
            d = np.zeros((k, 3)); d[:, ax] = sgn
            o = pick.mean(1).copy(); o[:, ax] -= sgn * 2 * ext[ax]
            rays.append(np.concatenate([o, d], axis=1))
            rays.append(np.concatenate([pick[:, 1] - 2 * ext[ax] * d, d], axis=1))    # through a vertex
    rays.append(np.concatenate([pick.mean(1), n], axis=1))                            # origin on the surface
    return np.concatenate(rays).astype(np.float32)


def _degenerate_rays(meshes, rs, k):
    """The rays for which triIntersect's determinant is zero to rounding: origin and direction in a triangle's plane -- anywhere in
    it, also tens of extents away from the triangle --, the same lifted or tilted out of the plane by 2^-10 ... 2^-26, and lines
    that cross the supporting line of a triangle's longer edge somewhere along it (for a needle: far beyond its tip)."""
    tri = np.concatenate([m.positions[m.indices.reshape(-1, 3)] for m in meshes]).astype(np.float64)
    ext = max(float(np.ptp(tri.reshape(-1, 3), axis=0).max()), 1e-3)
    rays = []
    for tilt in (None, 10, 18, 26):
        t = tri[rs.choice(len(tri), k)]
        e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
        n = np.cross(e1, e2)
        ok = (np.linalg.norm(n, axis=1) > 0) & (np.linalg.norm(e1, axis=1) > 0)
        t, e1, n = t[ok], e1[ok], n[ok]
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        b1 = e1 / np.linalg.norm(e1, axis=1, keepdims=True)
        b2 = np.cross(n, b1)
        reach = ext * np.where(rs.rand(len(t), 1) < 0.6, 1.0, 30.0)
        o = t[:, 0] + reach * (rs.uniform(-1, 1, (len(t), 1)) * b1 + rs.uniform(-1, 1, (len(t), 1)) * b2)
        ang = rs.uniform(-np.pi, np.pi, (len(t), 1))
        d = np.cos(ang) * b1 + np.sin(ang) * b2
        if tilt is not None:
            eps = 2.0 ** -tilt * rs.choice([-1.0, 1.0], (len(t), 1))
            lift = rs.rand(len(t), 1) < 0.5
            o = np.where(lift, o + eps * reach * n, o)
            d = np.where(lift, d, d + eps * n)
        rays.append(np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1))
    t = tri[rs.choice(len(tri), k)]
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    eL = np.where((np.linalg.norm(e1, axis=1) >= np.linalg.norm(e2, axis=1))[:, None], e1, e2)
    ok = np.linalg.norm(eL, axis=1) > 0
    t, eL = t[ok], eL[ok]
    target = t[:, 0] + eL / np.linalg.norm(eL, axis=1, keepdims=True) * rs.uniform(-3 * ext, 3 * ext, (len(t), 1))
    eye = target + rs.normal(size=(len(t), 3)) * ext
    d = target - eye
    rays.append(np.concatenate([eye, d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1))
    return np.concatenate(rays).astype(np.float32)


@pytest.mark.gpu
def test_bvh_trace_rays_equals_exhaustive(pkg, renderer):
    """spt_trace_rays through the hierarchy returns, bit for bit, the Hit of the exhaustive loop (itself equal to the oracle's, test
    above) on EVERY ray of five kinds of scene, random and adversarial -- since round 4 also on the rays rounds 2 and 3 had to except:
    rays lying in a regular triangle's plane (anywhere in it) and lines crossing the supporting line of a needle's long edge far from
    the needle, where triIntersect (no determinant cut-off, scene.cpp:62) reports noise that no bounding volume contains; the plane
    tree and the line tree of csrc/spt_tribvh.h find those triangles (CPU counterpart: tests/sanitize/tribvh_main.cpp)."""
    S = pkg.make_sphere_trimesh
    rs = np.random.RandomState(11)
    scenes = {"shipped": [S((-1, 0, -4), 1.0), S((1.5, 0, -5), 1.0)], "cornell-like": _mesh_scene(pkg)[0],
              "soup": [_soup(pkg, 3000, 4)], "flat soup + ball": [_soup(pkg, 1500, 5, flat=True), S((0, 3, 0), 2.0, 8)],
              "one": [pkg.single_triangle_scene()[0][0]]}
    total = hits = 0
    try:
        for name, meshes in scenes.items():
            mats = [((0, 0, 0), (.5, .5, .5), pkg.DIFF)] * len(meshes)
            rays = np.concatenate([_adversarial_rays(meshes, rs, 150000 if name == "shipped" else 60000), _degenerate_rays(meshes, rs, 4000)])
            renderer.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
            renderer.set_meshes(meshes, mats)
            ref = renderer.trace_rays(rays)
            renderer.set_mesh_accel(pkg.ACCEL_BVH)
            got = renderer.trace_rays(rays)
            bad = np.unique(np.nonzero(got.view(np.uint8).reshape(len(rays), -1) != ref.view(np.uint8).reshape(len(rays), -1))[0])
            assert len(bad) == 0, (name, len(bad), rays[bad[:3]], got[bad[:3]], ref[bad[:3]])
            assert (ref["dist"] < 1e20).sum() > len(rays) // 50, name
            total += len(rays); hits += int((ref["dist"] < 1e20).sum())
    finally:
        renderer.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
    assert total > 400000
    print(f"hierarchy == exhaustive on all {total} rays ({hits} hits)")


@pytest.mark.gpu
def test_exact_hierarchy_renders_and_camera_rays_list_the_camera_points_planes(pkg, oracle):
    """A mesh scene through SPT_ACCEL_BVH -- same image, same bounce count as the exhaustive kernel and the oracle, with both cameras.  For a pinhole camera the rays of depth 0 skip the plane tree and test the
    triangles in whose plane the camera's origin lies (spt_bvh.h camera_planes) instead: checked with the origin IN the plane of the
    scene's single triangle, looking along that plane (a non-empty list) and at main()'s position (an empty one)."""
    meshes, mats = _mesh_scene(pkg)
    tri = meshes[-1].positions[meshes[-1].indices.reshape(-1)].astype(np.float64)            # the single triangle of main()
    e1, e2 = tri[1] - tri[0], tri[2] - tri[0]
    org = tri[0] + 2.5 * e1 - 1.5 * e2                                                       # in its plane, beside it
    vz = tri.mean(0) - org
    vz /= np.linalg.norm(vz)                                                                 # looking along the plane at the triangle
    vx = np.cross(vz, np.cross(e1, e2))
    vx /= np.linalg.norm(vx)
    cams = [None, pkg.pinhole_camera(), pkg.pinhole_camera(vx=tuple(vx), vz=tuple(vz), org=tuple(org)), pkg.pinhole_camera(org=tuple(org))]
    w, h, samps, seed = 40, 30, 2, 5
    with pkg.Renderer(0) as r:
        r.set_mesh_accel(pkg.ACCEL_BVH)
        r.set_meshes(meshes, mats)
        imgs = []
        for cam in cams:
            img, st = r.render(w, h, samps, seed=seed, normalise=cam is None, camera=cam)
            assert r.last_kernel() == "mesh_bvh"
            imgs.append((img, st["bounces"]))
        r.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
        for cam, (img, bounces) in zip(cams, imgs):
            ref, st = r.render(w, h, samps, seed=seed, normalise=cam is None, camera=cam)
            assert np.array_equal(img, ref) and bounces == st["bounces"]
        for cam, (img, bounces) in zip(cams[:2], imgs[:2]):
            ref, rst = oracle.render_meshes(meshes, mats, w, h, samps, seed=seed, normalise=cam is None, camera=cam)
            assert np.array_equal(img, ref) and bounces == rst["bounces"]


@pytest.mark.gpu
def test_default_mode_picks_between_the_two_exact_modes(pkg, oracle):
    """SPT_ACCEL_AUTO, the default: a fresh context renders a mesh scene through the exact hierarchy; once a launch of a small scene
    (< 8192 triangles) has shown that more than 15 % of its closest-hit queries are bounce rays (they walk the plane tree) the next launch
    takes the exhaustive loop; a scene that is mostly camera rays, or large, stays on the hierarchy; fewer than 256 triangles always take
    the loop.  The image never depends on the choice."""
    S = pkg.make_sphere_trimesh
    meshes, mats = _mesh_scene(pkg)                          # 3329 triangles, 1.3-1.4 closest-hit queries per sample
    ref, rst = oracle.render_meshes(meshes, mats, 40, 30, 2, seed=5)
    with pkg.Renderer(0) as r:
        r.set_meshes(meshes, mats)                           # no set_mesh_accel: the default
        kernels = []
        for _ in range(3):
            img, st = r.render(40, 30, 2, seed=5)
            kernels.append(r.last_kernel())
            assert np.array_equal(img, ref) and st["bounces"] == rst["bounces"]
        assert kernels == ["mesh_bvh", "mesh", "mesh"], kernels
        shipped = [S((50, 40.8, 81.6), 10.0), S((50, 681.6 - .27, 81.6), 600.0)]            # 8192 triangles: the hierarchy whatever the bounce share
        r.set_meshes(shipped, [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((1, 1, 1), (0, 0, 0), pkg.DIFF)])
        for _ in range(2):
            r.render(64, 48, 1, seed=1)
            assert r.last_kernel() == "mesh_bvh"
        tiny, tmats = pkg.single_triangle_scene()
        r.set_meshes(tiny, tmats)
        r.render(64, 48, 1, seed=1, camera=pkg.pinhole_camera())
        assert r.last_kernel() == "mesh"
        r.set_mesh_accel(pkg.ACCEL_BVH)                      # asked for explicitly: taken
        r.render(64, 48, 1, seed=1, camera=pkg.pinhole_camera())
        assert r.last_kernel() == "mesh_bvh"


@pytest.mark.gpu
def test_bvh_fast_mode_agrees_on_ordinary_rays(pkg, renderer):
    """SPT_ACCEL_BVH_FAST (the spatial hierarchy alone, rounds 2-3; opt-in): random rays and rendered images agree with the exhaustive loop
    -- only rays lying in a triangle's plane to rounding may differ, which is why it is not the default."""
    S = pkg.make_sphere_trimesh
    meshes = [S((-1, 0, -4), 1.0), S((1.5, 0, -5), 1.0)]
    mats = [((0, 0, 0), (.5, .5, .5), pkg.DIFF)] * 2
    rs = np.random.RandomState(3)
    o = rs.uniform(-3, 3, (200000, 3)) + np.array([0, 0, 2.0])
    d = rs.normal(size=(200000, 3))
    rays = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1).astype(np.float32)
    try:
        renderer.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
        renderer.set_meshes(meshes, mats)
        ref = renderer.trace_rays(rays)
        img_ref, st_ref = renderer.render(48, 32, 2, seed=1, camera=pkg.pinhole_camera())
        renderer.set_mesh_accel(pkg.ACCEL_BVH_FAST)
        got = renderer.trace_rays(rays)
        img, st = renderer.render(48, 32, 2, seed=1, camera=pkg.pinhole_camera())
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)) and (ref["dist"] < 1e20).sum() > 1000
        assert np.array_equal(img, img_ref) and st["bounces"] == st_ref["bounces"]
    finally:
        renderer.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
        renderer.set_scene(pkg.cornell9())


@pytest.mark.gpu
def test_mesh_scenes_of_degenerate_triangles_only(pkg):
    """Scenes in which one of the hierarchy's structures (or all of them) is empty: only triangles with an edge of length zero (in no
    structure), only needles (the line table alone), one of each, a single regular triangle -- every mode returns the exhaustive loop's
    hits and image, nothing is read out of bounds."""
    def mesh(tris):
        v = np.asarray(tris, dtype=np.float32).reshape(-1, 3)
        return pkg.TriMesh(v, np.tile(np.array([0, 1, 0], dtype=np.float32), (len(v), 1)), np.arange(len(v), dtype=np.uint32).reshape(-1, 3))
    cases = {"only zero-edge triangles": [[[0, 0, 0], [0, 0, 0], [1, 0, 0]], [[1, 1, 1], [1, 1, 1], [2, 2, 2]]],
             "only needles": [[[0, 0, 0], [5, 0, 1e-7], [5, 0, 0]], [[0, 1, 0], [5, 1, 0], [5, 1, 2e-7]], [[0, 0, -3], [4, 0, -3], [2, 0, -3]]],
             "a zero-edge triangle and a regular one": [[[0, 0, 0], [0, 0, 0], [1, 0, 0]], [[-1, -1, -3], [1, -1, -3], [0, 1, -3]]],
             "one regular triangle": [[[-1, -1, -3], [1, -1, -3], [0, 1, -3]]]}
    rs = np.random.RandomState(0)
    o = rs.uniform(-2, 2, (20000, 3)); d = rs.normal(size=(20000, 3))
    rays = [np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1)]
    p = np.stack([rs.uniform(-10, 10, 2000), rs.choice([0.0, 1.0], 2000), rs.choice([0.0, 1e-7, -3.0], 2000)], axis=1)     # along the needles' lines, in their planes
    dd = np.stack([rs.choice([-1.0, 1.0], 2000), np.zeros(2000), rs.uniform(-1e-7, 1e-7, 2000)], axis=1)
    rays.append(np.concatenate([p, dd / np.linalg.norm(dd, axis=1, keepdims=True)], axis=1))
    rays = np.concatenate(rays).astype(np.float32)
    with pkg.Renderer(0) as r:
        for name, tris in cases.items():
            meshes = [mesh(tris)]
            mats = [((0, 0, 0), (.5, .5, .5), pkg.DIFF)]
            out = {}
            for accel in (pkg.ACCEL_EXHAUSTIVE, pkg.ACCEL_BVH, pkg.ACCEL_BVH_FAST):
                r.set_mesh_accel(accel)
                r.set_meshes(meshes, mats)
                out[accel] = (r.trace_rays(rays), r.render(32, 24, 1, seed=1, camera=pkg.pinhole_camera())[0])
            ref = out[pkg.ACCEL_EXHAUSTIVE]
            assert np.array_equal(out[pkg.ACCEL_BVH][0].view(np.uint8), ref[0].view(np.uint8)) and np.array_equal(out[pkg.ACCEL_BVH][1], ref[1]), name
            assert np.array_equal(out[pkg.ACCEL_BVH_FAST][1], ref[1]), name
            if "regular" in name:
                assert (ref[0]["dist"] < 1e20).sum() > 100, name


@pytest.mark.gpu
def test_bvh_render_equals_exhaustive_and_oracle(pkg, renderer, oracle):
    """The mesh path tracer through the hierarchy: same image, same bounce count as the exhaustive kernel and the oracle."""
    meshes, mats = _mesh_scene(pkg)
    w, h, samps, seed = 40, 30, 2, 5
    ref, rst = oracle.render_meshes(meshes, mats, w, h, samps, seed=seed)
    try:
        for accel in (pkg.ACCEL_EXHAUSTIVE, pkg.ACCEL_BVH):
            renderer.set_mesh_accel(accel)
            renderer.set_meshes(meshes, mats)                # the hierarchy is (re)built with the meshes
            img, st = renderer.render(w, h, samps, seed=seed)
            assert np.array_equal(img, ref) and st["bounces"] == rst["bounces"], accel
    finally:
        renderer.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
        renderer.set_scene(pkg.cornell9())
