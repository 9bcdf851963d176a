"""The fuzz generator's 1500-sphere cases at their ORIGINAL size (tools/fuzz_parity.py before its `n > 600` clamp: up to 24 x 16 pixels,
33 / 70 / 130 samples per jitter cell), once per closest-hit mode, with the oracle timed beside them.  Round 3's watchdog trip (case 4723
of a run whose seed and saved scene were in the scratch directory and are gone) was such a case; this replays the first ones of the
default stream (seed 1234) instead, or the cases named on the command line (found with the oracle: the ones in which paths reach the
depth cap of 4096 -- colour (1,1,1) mirrors and glass never die in the roulette).
usage: replay_deep_fuzz.py [number of cases | case,case,...] [watchdog seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc
from fuzz_recipe import draw_case, camera_of

arg = sys.argv[1] if len(sys.argv) > 1 else "3"
named = sorted(int(v) for v in arg.split(",")) if "," in arg or int(arg) > 100 else None
want = len(named) if named else int(arg)
wd = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rs = np.random.RandomState(1234)
r = pkg.Renderer(0)
r.set_watchdog(wd)
found = 0
case = -1
print("| case | spheres (white, huge) | image, samples per cell | mode | kernel | kernel ms | bounces | deepest cut (depth 4096) | oracle s (all host cores) | bit-exact |")
print("|---|---|---|---|---|---|---|---|---|---|")
while found < want:
    case += 1
    cs = draw_case(rs, pkg)
    n, w, h, samps, seed, norm, rows = cs["n"], cs["w"], cs["h"], cs["samps"], cs["seed"], cs["norm"], cs["rows"]
    cam = camera_of(cs, pkg)
    if (named is not None and case not in named) or (named is None and (n != 1500 or samps < 33)):
        continue
    found += 1
    sc = pkg.make_spheres(rows)
    white = sum(1 for q in rows if q[3] == (1.0, 1.0, 1.0)); huge = sum(1 for q in rows if q[0] >= 100)
    t0 = time.time()
    ref, rst = orc.render(sc, w, h, samps, seed=seed, normalise=norm, camera=cam)
    t_or = time.time() - t0
    for mode, accel, lane_owned in (("grid, path pools (default)", pkg.ACCEL_GRID, False), ("grid, lane-owned", pkg.ACCEL_GRID, True),
                                    ("hierarchy", pkg.ACCEL_BVH, False), ("exhaustive", pkg.ACCEL_EXHAUSTIVE, False)):
        r.set_grid_pools(lane_owned=lane_owned)
        r.set_sphere_accel(accel)
        r.set_scene(sc)
        try:
            img, st = r.render(w, h, samps, seed=seed, normalise=norm, camera=cam)
            ok = bool(np.array_equal(img, ref, equal_nan=True)) and st["bounces"] == rst["bounces"] and st["max_depth_kills"] == rst["max_depth_kills"]
            print(f"| {case} | {n} ({white}, {huge}) | {w}x{h}, {samps}{' pinhole' if cam is not None else ''} | {mode} | {r.last_kernel()} | {st['kernel_ms']:.1f} | {st['bounces']} | "
                  f"{st['max_depth_kills']} | {t_or:.2f} | {ok} |", flush=True)
        except Exception as e:
            print(f"| {case} | {n} ({white}, {huge}) | {w}x{h}, {samps} | {mode} | {r.last_kernel()} | FAILED: {e} | | | {t_or:.2f} | |", flush=True)
r.set_grid_pools()
