"""The random-scene recipe of the parity fuzzer (tools/fuzz_parity.py), shared with tools/replay_deep_fuzz.py and the test that pins one of
its heavy cases: one call draws one case from a numpy RandomState -- sphere table (all materials, emitters, overlapping spheres, tiny and
huge radii, 5 % colour (1,1,1)), image size, samples per jitter cell, seed, camera, normalisation, closest-hit mode.  The order of the
draws is part of the recipe: case k of a seed is the same scene in every tool."""
import numpy as np


def draw_case(rs, pkg):
    n = int(rs.choice([1, 2, 3, 5, 9, 17, 24, 25, 40, 100, 257, 600, 1500]))
    rows = []
    for _ in range(n):
        kind = rs.rand()
        rad = float(10 ** rs.uniform(-1, 1.3)) if kind < 0.8 else float(10 ** rs.uniform(2, 5))
        c = (rs.uniform(-20, 120), rs.uniform(-20, 100), rs.uniform(-50, 250))
        if kind >= 0.8:   # huge "wall" sphere placed so that the camera is inside or just outside
            c = tuple(float(v) for v in (np.array([50, 40, 80]) + (rs.randn(3) / np.linalg.norm(rs.randn(3) + 1e-9)) * rad * rs.uniform(0.9, 1.1)))
        e = (0, 0, 0) if rs.rand() < 0.8 else tuple(rs.uniform(0, 5, 3))
        col = tuple(rs.uniform(0, 1, 3)) if rs.rand() < 0.9 else (0, 0, 0)
        if rs.rand() < 0.05:
            col = (1.0, 1.0, 1.0)
        rows.append((rad, c, e, col, int(rs.choice([0, 0, 0, 1, 2]))))
    w, h = int(rs.randint(1, 70)), int(rs.randint(1, 50))
    samps = int(rs.choice([1, 1, 2, 3, 7, 33, 70, 130]))       # >= 32: several D9 sample blocks per jitter cell
    if samps > 7:
        w, h = min(w, 24), min(h, 16)
    seed = int(rs.randint(0, 2**31)) * int(rs.choice([1, 2**20]))
    pinhole = not (rs.rand() < 0.6)
    norm = bool(rs.rand() < 0.5)
    accel = [pkg.ACCEL_GRID, pkg.ACCEL_GRID, pkg.ACCEL_BVH, pkg.ACCEL_EXHAUSTIVE][rs.randint(4)]     # tables above 24 spheres: grid (default), hierarchy or megakernel
    return dict(n=n, rows=rows, w=w, h=h, samps=samps, seed=seed, pinhole=pinhole, norm=norm, accel=accel,
                white=sum(1 for q in rows if q[3] == (1.0, 1.0, 1.0)), huge=sum(1 for q in rows if q[0] >= 100))


def camera_of(case, pkg):
    return pkg.pinhole_camera(org=(50, 45, 250), vz=(0, 0, -1)) if case["pinhole"] else None
