"""Randomised GPU-vs-oracle parity: random sphere tables (all materials, emitters, overlapping spheres, tiny and huge
radii), random image sizes / spp / seeds / cameras.  Every case must be bit-identical, including bounce counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
skip = int(os.environ.get("FUZZ_SKIP", "0"))          # replay: draw the first N cases without rendering them (same random sequence)
wd = float(os.environ.get("FUZZ_WATCHDOG", "60"))
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
r = pkg.Renderer(0)
r.set_watchdog(wd)
kernels = {}
t0 = time.time(); cases = 0; bad = 0; last_note = t0; reordered = 0
while time.time() - t0 < budget:
    n = int(rs.choice([1, 2, 3, 5, 9, 17, 24, 25, 40, 100, 257, 600, 1500]))
    rows = []
    for i in range(n):
        kind = rs.rand()
        rad = float(10 ** rs.uniform(-1, 1.3)) if kind < 0.8 else float(10 ** rs.uniform(2, 5))
        c = (rs.uniform(-20, 120), rs.uniform(-20, 100), rs.uniform(-50, 250))
        if kind >= 0.8:   # huge "wall" sphere placed so that the camera is inside or just outside
            c = tuple(float(v) for v in (np.array([50, 40, 80]) + (rs.randn(3) / np.linalg.norm(rs.randn(3)+1e-9)) * rad * rs.uniform(0.9, 1.1)))
        e = (0, 0, 0) if rs.rand() < 0.8 else tuple(rs.uniform(0, 5, 3))
        col = tuple(rs.uniform(0, 1, 3)) if rs.rand() < 0.9 else (0, 0, 0)
        if rs.rand() < 0.05: col = (1.0, 1.0, 1.0)
        rows.append((rad, c, e, col, int(rs.choice([0, 0, 0, 1, 2]))))
    sc = pkg.make_spheres(rows)
    w, h = int(rs.randint(1, 70)), int(rs.randint(1, 50))
    samps = int(rs.choice([1, 1, 2, 3, 7, 33, 70, 130]))       # >= 32: several D9 sample blocks per jitter cell
    if samps > 7:
        w, h = min(w, 24), min(h, 16)
    if n > 600:          # a 4096-bounce path (white spheres never die in the roulette) over 1500 spheres costs 0.1-0.3 s on a lone lane, and the oracle as much
        samps = min(samps, 3)
        w, h = min(w, 24), min(h, 16)
    seed = int(rs.randint(0, 2**31)) * int(rs.choice([1, 2**20]))
    cam = None if rs.rand() < 0.6 else pkg.pinhole_camera(org=(50, 45, 250), vz=(0, 0, -1))
    norm = bool(rs.rand() < 0.5)
    accel = [pkg.ACCEL_GRID, pkg.ACCEL_GRID, pkg.ACCEL_BVH, pkg.ACCEL_EXHAUSTIVE][rs.randint(4)]     # tables above 24 spheres: grid (default), hierarchy or megakernel
    if cases < skip:
        cases += 1
        continue
    r.set_sphere_accel(accel)
    r.set_scene(sc)
    try:
        img, st = r.render(w, h, samps, seed=seed, normalise=norm, camera=cam)
    except Exception as e:
        print("FAILED case", cases, dict(n=n, w=w, h=h, samps=samps, seed=seed, pinhole=cam is not None, norm=norm, accel=accel), e, flush=True)
        np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_failed_scene_{cases}.npy"), sc)
        raise
    ref, rst = orc.render(sc, w, h, samps, seed=seed, normalise=norm, camera=cam)
    ok = np.array_equal(img, ref, equal_nan=True) and st["bounces"] == rst["bounces"] and st["max_depth_kills"] == rst["max_depth_kills"]
    if samps >= 16 and r.last_kernel() == "pool":
        # the same view again: the pool kernel now hands its task chunks out in the cost order this launch left (spt_api.cpp)
        img2, st2 = r.render(w, h, samps, seed=seed, normalise=norm, camera=cam)
        ok = ok and np.array_equal(img2, ref, equal_nan=True) and st2["bounces"] == rst["bounces"] and len(r.chunk_order()) == (w * h * 4 * (8 if samps >= 128 else 4 if samps >= 64 else 2 if samps >= 32 else 1) + 63) // 64
        reordered += 1
    cases += 1
    kernels[r.last_kernel()] = kernels.get(r.last_kernel(), 0) + 1
    if time.time() - last_note > 30:             # a silent GPU command is taken for hung after a few minutes
        last_note = time.time()
        print(f"... {cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    if not ok:
        bad += 1
        print("MISMATCH case", cases, dict(n=n, w=w, h=h, samps=samps, seed=seed, pinhole=cam is not None, norm=norm),
              "pixels differ", int((img != ref).any(axis=-1).sum()), "bounces", st["bounces"], rst["bounces"], flush=True)
print(f"fuzz: {cases} cases, {bad} mismatches, {time.time()-t0:.0f} s, kernels {kernels}, {reordered} of them rendered a second time in cost order")
sys.exit(1 if bad else 0)
