"""Config 5's table plus the Cornell box's mirror and glass balls (colour .999: chains of thousands of bounces inside the mirror
ball): how many rays leave the grid because their direction length has drifted, and what that costs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

base = pkg.random_spheres(1024, 1024)
c9 = pkg.cornell9()
sc = np.concatenate([base, c9[6:8]])            # + mirror ball, glass ball
r = pkg.Renderer(0)
r.set_watchdog(120.0)
samps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, scene in (("config 5", base), ("config 5 + mirror and glass balls", sc)):
    for variant in (0x100, 0):
        r.set_tuning(0, variant)
        r.set_scene(scene)
        r.render(1024, 768, samps, seed=0, normalise=True)
        img, st = r.render(1024, 768, samps, seed=0, normalise=True)
        d = r.diag()
        extra = f" exhaustive-loop rays {d[4]} of {st['bounces']} ({d[4] / st['bounces']:.2e})" if variant else ""
        print(f"{name}: {'stats build' if variant else 'product'} kernel {st['kernel_ms']:.1f} ms, {st['samples'] / st['kernel_ms'] / 1e3:.0f} Msamples/s, "
              f"bounces/sample {st['bounces'] / st['samples']:.3f}, depth kills {st['max_depth_kills']}{extra}", flush=True)
    ref, rst = orc.render(scene, 1024, 768, samps, seed=0, normalise=True, row_begin=200, row_count=1)
    print("   row 200 bit-exact:", bool(np.array_equal(img[200:201], ref)))
r.set_tuning(0, 0)
