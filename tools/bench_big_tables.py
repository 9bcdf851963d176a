"""Sphere tables beyond one CU's LDS (16 384 ... 65 535 random spheres; pkg.random_spheres = config 5's recipe): the grid with its tables in
global memory (spt_grid.hip GLOBAL_TABLES, default since round 4) against the hierarchy (SPT_ACCEL_BVH, round 3's default for these), an
oracle row for parity.  usage: bench_big_tables.py [samps per cell] [n,n,...] [blocks per CU:threads/128-1, ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

samps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ns = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16384]
geoms = [tuple(int(x) for x in g.split(":")) for g in sys.argv[3].split(",")] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else [(0, 0)]
w, h = 1024, 768
r = pkg.Renderer(0)
r.set_watchdog(120.0)
for n in ns:
    sc = pkg.random_spheres(n, 7)
    ref = orc.render(sc, w, h, samps, seed=0, normalise=True, row_begin=300, row_count=1)[0]
    for accel, name in ((pkg.ACCEL_GRID, "grid"), (pkg.ACCEL_GRID, "grid, tables forced into global memory"), (pkg.ACCEL_GRID, "grid, records forced into global memory (grid in LDS)"),
                        (pkg.ACCEL_BVH, "hierarchy")):
        if "tables forced" in name and "--force-global" not in sys.argv:
            continue
        if "records forced" in name and "--force-hybrid" not in sys.argv:
            continue
        for per_cu, tsel in (geoms if accel == pkg.ACCEL_GRID else [(0, 0)]):
            r.set_tuning(per_cu, tsel << 13)
            r.set_grid_pools(lane_owned=2 if "tables forced" in name else (3 if "records forced" in name else 0))
            r.set_sphere_accel(accel)
            r.set_scene(sc)
            best = None
            for _ in range(3):
                img, st = r.render(w, h, samps, seed=0, normalise=True)
                best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
            print(json.dumps({"spheres": n, "mode": name, "kernel": r.last_kernel(), "blocks": best["grid_blocks"], "threads": best["block_threads"], "spp": 4 * samps,
                              "kernel_ms": round(best["kernel_ms"], 2), "msamples_s": round(best["samples"] / best["kernel_ms"] / 1e3, 1),
                              "bounces_per_sample": round(best["bounces"] / best["samples"], 3), "bit_exact_row_300": bool(np.array_equal(img[300:301], ref))}), flush=True)
r.set_tuning(0, 0)
r.set_grid_pools()
r.set_sphere_accel(pkg.ACCEL_GRID)
