"""Where one bounce of a lone path spends its 1.5 us: the pool kernel built with -DSPT_POOL_PHASES (s_memtime stamps between
the phases of an iteration; run with SPT_LIB pointing at that build) on the closed-mirror scene of probe_chain_latency.py."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg

r = pkg.Renderer(0)
r.set_watchdog(20.0)
cam = pkg.pinhole_camera(vx=(1, 0, 0), vz=(0, 0, -1), org=(90, 0, 0), near=1.0)
names = ["select + pop", "class code (normal, reflection)", "closest hit", "post (material, roulette, store)", "push"]
for label, refl, w in (("mirror, 4 live lanes (narrow closest hit)", pkg.SPEC, 1), ("mirror, 64 live lanes", pkg.SPEC, 4), ("diffuse, 4 live lanes", pkg.DIFF, 1)):
    sc = pkg.make_spheres([(100.0, (0, 0, 0), (0, 0, 0), (1, 1, 1), refl)] + [(1.0, (1e4 + 10 * i, 0, 0), (0, 0, 0), (.5, .5, .5), pkg.DIFF) for i in range(8)])
    r.set_scene(sc)
    out, st = r.render(w, w, 1, seed=1, normalise=True, camera=cam)
    out, st = r.render(w, w, 1, seed=1, normalise=True, camera=cam)
    d = r.diag()
    its = d[20]
    print(f"{label}: kernel {st['kernel_ms']:.3f} ms, {its} iterations over all waves")
    tot = sum(d[15:20])
    for n, v in zip(names, d[15:20]):
        print(f"    {n:34s} {v / max(its, 1):8.0f} ticks per iteration  {100 * v / max(tot, 1):5.1f} %")
    print(f"    {'total':34s} {tot / max(its, 1):8.0f} ticks = {tot / max(its, 1) / 2.4e3:.3f} us at 2.4 GHz (s_memtime ticks at the shader clock)")
