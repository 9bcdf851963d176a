"""Config 5 (1024 spheres, 1024x768) and larger tables on the grid kernel (spt_grid.hip): kernel time, Msamples/s, walk statistics
(cell steps / sphere tests per ray, lane utilisation of both loop bodies), oracle rows for parity, and A/B over the tuning knobs.
usage: bench_grid.py [samps] [variant,variant,...] [--lane-owned] [--stats] [--nocheck] [--big]
(variants: hex words for spt_set_tuning, e.g. 0x0,0x50000,0x10000000; --lane-owned: spt_grid.hip instead of the path pools of spt_gpool.hip)"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

samps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = [(int(v.split(":")[0], 16), int(v.split(":")[1]) if ":" in v else 0) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else [(0, 0)]   # variant[:blocks per CU]
scenes = [("config 5: 1024 spheres", pkg.random_spheres(1024, 1024), [100, 500])]
if "--big" in sys.argv:
    scenes.append(("4096 spheres", pkg.random_spheres(4096, 7), [300]))
r = pkg.Renderer(0)
r.set_watchdog(120.0)
if "--lane-owned" in sys.argv:
    r.set_grid_pools(lane_owned=True)
rows = []
w, h = 1024, 768
for name, sc, check_rows in scenes:
    refs = {row: orc.render(sc, w, h, samps, seed=0, normalise=True, row_begin=row, row_count=1)[0] for row in check_rows} if "--nocheck" not in sys.argv else {}
    for variant, per_cu in variants:
        r.set_tuning(per_cu, variant | (0x100 if "--stats" in sys.argv else 0))
        r.set_scene(sc)
        img, st = r.render(w, h, samps, seed=0, normalise=True)
        best = st
        for _ in range(2):
            img, st = r.render(w, h, samps, seed=0, normalise=True)
            if st["kernel_ms"] < best["kernel_ms"]:
                best = st
        st = best
        exact = all(bool(np.array_equal(img[row:row + 1], ref)) for row, ref in refs.items())
        d = r.diag()
        rays = st["bounces"]
        out = {"scene": name, "variant": hex(variant), "blocks": st["grid_blocks"], "threads": st["block_threads"], "kernel": r.last_kernel(), "image": f"{w}x{h}", "spp": 4 * samps,
               "kernel_ms": round(st["kernel_ms"], 2), "msamples_s": round(st["samples"] / st["kernel_ms"] / 1e3, 1),
               "bounces_per_sample": round(st["bounces"] / st["samples"], 4), "oracle_rows": len(refs), "bit_exact": exact}
        if r.last_kernel() == "grid" and rays and "--stats" in sys.argv:
            out.update({"steps_per_ray": round(d[0] / rays, 2), "tests_per_ray": round(d[1] / rays, 2),
                        "step_lanes": round(d[0] / max(1, d[2]) / 64, 3), "test_lanes": round(d[1] / max(1, d[3]) / 64, 3),
                        "exhaustive_rays": d[4], "rounds": d[5], "shade_lanes": round(d[7] / max(1, d[5]) / 64, 3)})
            tot = max(1, d[13])
            out["phase_share"] = {k: round(d[8 + i] / tot, 3) for i, k in enumerate(("regen", "begin", "test", "step", "shade"))}
        if r.last_kernel() == "gpool" and rays and "--stats" in sys.argv:
            it = max(1, d[2])
            out.update({"steps_per_ray": round(d[0] / rays, 2), "tests_per_ray": round(d[1] / rays, 2), "walk_iterations_per_64_rays": round(64 * d[2] / rays, 2),
                        "walking_lanes": round(d[3] / it / 64, 3), "step_lanes": round(d[0] / it / 64, 3), "test_lanes": round(d[1] / it / 64, 3),
                        "exhaustive_rays": d[4], "exchanges_per_64_rays": round(64 * d[5] / rays, 2),
                        "batch_lanes": {k: round(d[11 + i] / max(1, d[8 + i]), 1) for i, k in enumerate(("gen", "hit", "hitr"))},
                        "batches_per_64_rays": {k: round(64 * d[8 + i] / rays, 3) for i, k in enumerate(("gen", "hit", "hitr"))}})
            tot = max(1, d[18])
            out["phase_share"] = {k: round(d[14 + i] / tot, 3) for i, k in enumerate(("walk", "exchange", "gen+begin", "shade+begin"))}
            nw = st["grid_blocks"] * st["block_threads"] // 64
            out["wave_ms_at_2.4GHz"] = {"mean": round(d[18] / nw / 2.4e6, 2), "longest": round(d[19] / 2.4e6, 2)}
        print(json.dumps(out), flush=True)
        rows.append(out)
r.set_tuning(0, 0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "grid_bench.json"), "w"), indent=1)
