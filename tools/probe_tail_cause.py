"""Which sphere makes the launch tail?  Config-2 shape with the glass ball turned into a mirror, the mirror ball into a diffuse
one, or both; prints kernel time, the longest drain after the queue ran dry and the depth-cap kills."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg

r = pkg.Renderer(0)
r.set_watchdog(30.0)
for label, edits in (("cornell9", {}), ("glass -> mirror", {7: pkg.SPEC}), ("mirror -> diffuse .75", {6: pkg.DIFF}), ("both", {6: pkg.DIFF, 7: pkg.SPEC})):
    sc = pkg.cornell9()
    for i, refl in edits.items():
        sc[i]["refl"] = refl
        if refl == pkg.DIFF:
            sc[i]["color"] = (.75, .75, .75)
    r.set_scene(sc)
    _, st = r.render(1024, 768, 256, seed=0, normalise=True)
    _, st = r.render(1024, 768, 256, seed=0, normalise=True)
    d = r.diag()
    print(f"{label:22s}: {st['kernel_ms']:.2f} ms, bounces/sample {st['bounces'] / st['samples']:.3f}, mean wave {d[13] / 4096 / 2.4e6:.2f} ms, "
          f"drain max {d[11] / 2.4e6:.2f} mean {d[12] / 4096 / 2.4e6:.2f} ms, depth-cap kills {st['max_depth_kills']}", flush=True)
