"""Scene model kept from the reference: Sphere{radius, center, Material{emission, color, refl}}
(scene.h:58-92), the Cornell-9 table (smallpt.cpp:36-48), the 1024-sphere stress scene of
SURVEY.md 8(d) config 5, and the JSON scene file of SURVEY.md 8(f)."""
import json

import numpy as np

DIFF, SPEC, REFR = 0, 1, 2  # Refl_t, scene.h:64
REFL_NAMES = {DIFF: "DIFF", SPEC: "SPEC", REFR: "REFR"}
REFL_IDS = {v: k for k, v in REFL_NAMES.items()}

# binary layout of spt_sphere (include/smallpt_mi355x.h), 48 bytes
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("emission", "<f4", 3),
                         ("color", "<f4", 3), ("refl", "<i4"), ("pad", "<u4")])
assert SPHERE_DTYPE.itemsize == 48


def make_spheres(rows):
    """rows: iterable of (radius, center, emission, color, refl) -- the Sphere ctor order, scene.h:91."""
    rows = list(rows)
    a = np.zeros(len(rows), dtype=SPHERE_DTYPE)
    for i, (r, c, e, col, t) in enumerate(rows):
        a[i]["radius"] = r
        a[i]["center"] = c
        a[i]["emission"] = e
        a[i]["color"] = col
        a[i]["refl"] = t
    return a


def cornell9(light_emission=1.0):
    """The 9-sphere Cornell box commented out at smallpt.cpp:38-46 (D11: emission (1,1,1),
    mirror/glass colour .999).  light_emission=12 gives the classic smallpt variant."""
    z = (0, 0, 0)
    e = (light_emission,) * 3
    return make_spheres([
        (1e5, (1e5 + 1, 40.8, 81.6), z, (.75, .25, .25), DIFF),    # Left   :38
        (1e5, (-1e5 + 99, 40.8, 81.6), z, (.25, .25, .75), DIFF),  # Rght   :39
        (1e5, (50, 40.8, 1e5), z, (.75, .75, .75), DIFF),          # Back   :40
        (1e5, (50, 40.8, -1e5 + 170), z, z, DIFF),                 # Frnt   :41
        (1e5, (50, 1e5, 81.6), z, (.75, .75, .75), DIFF),          # Botm   :42
        (1e5, (50, -1e5 + 81.6, 81.6), z, (.75, .75, .75), DIFF),  # Top    :43
        (16.5, (27, 16.5, 47), z, (.999, .999, .999), SPEC),       # Mirr   :44
        (16.5, (73, 16.5, 78), z, (.999, .999, .999), REFR),       # Glas   :45
        (600, (50, 681.6 - .27, 81.6), e, z, DIFF),                # Lite   :46
    ])


class _SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self):
        return (self.next() >> 11) * (1.0 / (1 << 53))


def random_spheres(total=1024, seed=1024):
    """Config 5: the 6 walls + light of Cornell-9 plus (total-7) random spheres from
    SplitMix64(seed); per sphere in order: radius=0.5+2u; center=(5+90u, 3+70u, 10+140u);
    colour=.25+.7u x3; refl u<.70 DIFF, <.85 SPEC, else REFR; emission 0."""
    base = cornell9()
    keep = [0, 1, 2, 3, 4, 5, 8]
    rng = _SplitMix64(seed)
    rows = []
    for _ in range(total - len(keep)):
        r = 0.5 + 2 * rng.uniform()
        c = (5 + 90 * rng.uniform(), 3 + 70 * rng.uniform(), 10 + 140 * rng.uniform())
        col = (.25 + .7 * rng.uniform(), .25 + .7 * rng.uniform(), .25 + .7 * rng.uniform())
        u = rng.uniform()
        t = DIFF if u < .70 else (SPEC if u < .85 else REFR)
        rows.append((r, c, (0, 0, 0), col, t))
    out = np.concatenate([base[keep], make_spheres(rows)])
    assert len(out) == total
    return out


# ---- JSON scene file: vectors are 3-number arrays like the reference's
# request messages (smallpt.cpp:913,983); field order = Sphere ctor (scene.h:91) ----
def spheres_to_json(spheres, camera=None):
    doc = {"spheres": [
        {"radius": float(s["radius"]), "center": [float(x) for x in s["center"]],
         "emission": [float(x) for x in s["emission"]], "color": [float(x) for x in s["color"]],
         "refl": REFL_NAMES[int(s["refl"])]} for s in spheres]}
    if camera is not None:
        doc["camera"] = camera
    return json.dumps(doc)


def spheres_from_json(text):
    doc = json.loads(text)
    rows = []
    for s in doc["spheres"]:
        refl = s["refl"]
        refl = REFL_IDS[refl] if isinstance(refl, str) else int(refl)
        rows.append((s["radius"], s["center"], s["emission"], s["color"], refl))
    return make_spheres(rows), doc.get("camera")


# ---- triangle meshes (scene.h:6-15 TriMesh, scene.cpp:3-48 makeSphereTriMesh) ----
HIT_DTYPE = np.dtype([("dist", "<f4"), ("instId", "<u4"), ("triId", "<u4"), ("x", "<f4", 3), ("n", "<f4", 3), ("uv", "<f4", 2)])   # Hit, scene.h:31-43
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3)])                                                                           # Ray, scene.h:58-62
assert HIT_DTYPE.itemsize == 44 and RAY_DTYPE.itemsize == 24


class TriMesh:
    """positionBuffer / normalBuffer / indexBuffer of the reference's TriMesh as contiguous numpy arrays."""

    def __init__(self, positions, normals, indices):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        assert len(self.positions) == len(self.normals)

    @property
    def triangle_count(self):          # TriMesh::triangleCount, scene.h:11-14
        return len(self.indices)


def make_sphere_trimesh(origin, radius, subdiv_longitude=32):
    """makeSphereTriMesh(origin, radius, subdivLongitude = 32) (scene.cpp:3-48, scene.h:17) through the library's host
    helper (the reference generates the table on the host with the C library's float cos/sin)."""
    import ctypes as C
    from ._lib import load_library
    n = int(subdiv_longitude)
    pos = np.zeros(((n + 1) * (2 * n + 1), 3), dtype=np.float32)
    nor = np.zeros_like(pos)
    idx = np.zeros((4 * n * n, 3), dtype=np.uint32)
    nt = load_library().spt_make_sphere_trimesh((C.c_float * 3)(*[float(v) for v in origin]), float(radius), n,
                                                pos.ctypes.data_as(C.c_void_p), nor.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p))
    assert nt == len(idx)
    return TriMesh(pos, nor, idx)


def single_triangle_scene():
    """SingleTriangleScene of the reference's main() (smallpt.cpp:818-832): one triangle, material emission (1,0,0), DIFF."""
    mesh = TriMesh([(-0.5, -0.5, -2), (0.5, -0.5, -2), (0, 0.5, -2)], [(1, 0, 0), (0, 1, 0), (0, 0, 1)], [(0, 1, 2)])
    return [mesh], [((1, 0, 0), (0, 0, 0), DIFF)]


def meshes_to_json(meshes, materials, generators=None, camera=None, spheres=None):
    """Scene file with a "meshes" array (host/scene.hpp): entry i is {"sphere": {"center", "radius", "subdiv"}} when
    generators[i] = (center, radius, subdiv) (re-tessellated by the loader with makeSphereTriMesh) or explicit
    "positions"/"normals"/"indices" buffers, plus the instance's material."""
    out = []
    for i, (m, (e, col, refl)) in enumerate(zip(meshes, materials)):
        ent = {}
        gen = generators[i] if generators is not None else None
        if gen is not None:
            c, r, sd = gen
            ent["sphere"] = {"center": [float(np.float32(v)) for v in c], "radius": float(np.float32(r)), "subdiv": int(sd)}
        else:
            ent["positions"] = [[float(v) for v in p] for p in m.positions]
            ent["normals"] = [[float(v) for v in p] for p in m.normals]
            ent["indices"] = [[int(v) for v in t] for t in m.indices]
        ent["emission"] = [float(np.float32(v)) for v in e]
        ent["color"] = [float(np.float32(v)) for v in col]
        ent["refl"] = REFL_NAMES[int(refl)]
        out.append(ent)
    doc = {"meshes": out}
    if spheres is not None:
        doc["spheres"] = json.loads(spheres_to_json(spheres))["spheres"]
    if camera is not None:
        doc["camera"] = camera
    return json.dumps(doc)


def meshes_from_json(text):
    """Inverse of meshes_to_json: returns (meshes, materials); "sphere" entries are tessellated with make_sphere_trimesh."""
    doc = json.loads(text)
    meshes, mats = [], []
    for ent in doc.get("meshes", []):
        if "sphere" in ent:
            g = ent["sphere"]
            meshes.append(make_sphere_trimesh(g["center"], g["radius"], g.get("subdiv", 32)))
        else:
            meshes.append(TriMesh(ent["positions"], ent["normals"], ent["indices"]))
        refl = ent["refl"]
        mats.append((tuple(ent["emission"]), tuple(ent["color"]), REFL_IDS[refl] if isinstance(refl, str) else int(refl)))
    return meshes, mats
