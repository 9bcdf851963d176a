"""First-light / A-B script of the grid-pool kernel (spt_gpool.hip): parity with the oracle on small cases (several tables, sizes,
seeds, both cameras), then config 5 timing against the lane-owned grid kernel (tuning bit 24)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

r = pkg.Renderer(0)
r.set_watchdog(30.0)
bad = 0
cases = [(pkg.random_spheres(200, 5), 48, 36, 1, 3), (pkg.random_spheres(1024, 1024), 64, 48, 2, 0), (pkg.random_spheres(1024, 1024), 40, 30, 40, 7),
         (pkg.random_spheres(600, 11), 33, 17, 3, 9), (pkg.random_spheres(30, 2), 64, 64, 4, 1), (pkg.random_spheres(1500, 4), 32, 24, 8, 2)]
for i, (sc, w, h, samps, seed) in enumerate(cases):
    r.set_scene(sc)
    img, st = r.render(w, h, samps, seed=seed, normalise=True)
    ref, rst = orc.render(sc, w, h, samps, seed=seed, normalise=True)
    ok = bool(np.array_equal(img, ref)) and st["bounces"] == rst["bounces"]
    bad += 0 if ok else 1
    print(f"case {i}: n={len(sc)} {w}x{h} samps={samps} kernel={r.last_kernel()} bounces {st['bounces']} / {rst['bounces']} bit_exact={bool(np.array_equal(img, ref))} "
          f"maxdiff={float(np.abs(img - ref).max()):.3e} kernel_ms={st['kernel_ms']:.2f}", flush=True)
if bad:
    print("PARITY FAILED", bad)
    sys.exit(1)
sc = pkg.random_spheres(1024, 1024)
refs = {row: orc.render(sc, 1024, 768, 64, seed=0, normalise=True, row_begin=row, row_count=1)[0] for row in (100, 500)}
for variant in (0, 1):
    r.set_grid_pools(lane_owned=bool(variant))
    r.set_scene(sc)
    best = None
    for _ in range(3):
        img, st = r.render(1024, 768, 64, seed=0, normalise=True)
        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
    exact = all(bool(np.array_equal(img[row:row + 1], ref)) for row, ref in refs.items())
    print(f"config 5 at 256 spp, lane_owned={variant}: kernel={r.last_kernel()} kernel_ms={best['kernel_ms']:.2f} Msamples/s={best['samples'] / best['kernel_ms'] / 1e3:.1f} bit_exact={exact}", flush=True)
