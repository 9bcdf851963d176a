"""Latency of one bounce of a lone path (what sets the end of a launch): camera inside a closed sphere of colour 1, so
every path runs to the depth cap (4096 bounces); a 1x1 image at 1 sample per cell is one wave with 4 live lanes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg

r = pkg.Renderer(0)
r.set_watchdog(20.0)
if len(sys.argv) > 1:
    r.set_tuning(0, int(sys.argv[1], 0))          # e.g. 0x400: the megakernel (one lane owns a path) instead of the pool kernel
cam = pkg.pinhole_camera(vx=(1, 0, 0), vz=(0, 0, -1), org=(90, 0, 0), near=1.0)
for label, refl in (("mirror", pkg.SPEC), ("glass", pkg.REFR), ("diffuse", pkg.DIFF)):
    sc = pkg.make_spheres([(100.0, (0, 0, 0), (0, 0, 0), (1, 1, 1), refl)] + [(1.0, (1e4 + 10 * i, 0, 0), (0, 0, 0), (.5, .5, .5), pkg.DIFF) for i in range(8)])
    r.set_scene(sc)
    for w, h in ((1, 1), (4, 4), (16, 16)):
        out, st = r.render(w, h, 1, seed=1, normalise=True, camera=cam)
        out, st = r.render(w, h, 1, seed=1, normalise=True, camera=cam)
        b = st["bounces"] / st["samples"]
        print(f"{label} {w}x{h}: kernel {st['kernel_ms']:.3f} ms, {b:.0f} bounces/sample, {st['kernel_ms'] * 1e3 / max(b, 1):.3f} us per bounce of the chain, kernel {r.last_kernel()}", flush=True)
