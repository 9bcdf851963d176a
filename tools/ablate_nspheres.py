"""Marginal cost per sphere test: Cornell-9 plus k never-hit dummy spheres (far outside the box) renders
the same paths; kernel time vs N gives time/bounce = a + b*N."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import optix_test_smallpt_amd as pkg
r = pkg.Renderer(0)
samps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
base = pkg.cornell9()
res = []
for extra in (0, 3, 9, 18, 36):
    dummy = pkg.make_spheres([(0.5, (5000.0 + 3 * i, 5000.0, 5000.0), (0, 0, 0), (.5, .5, .5), 0) for i in range(extra)])
    sc = np.concatenate([base, dummy]) if extra else base
    r.set_scene(sc)
    best = 1e9
    for _ in range(3):
        _, st = r.render(1024, 768, samps)
        best = min(best, st["kernel_ms"])
    nb = st["bounces"]
    res.append((len(sc), best, nb))
    print("N=%d kernel_ms=%.2f bounces=%d  ns/bounce(chip)=%.4f  SIMD-cycles/wave-bounce=%.0f" % (len(sc), best, nb, best * 1e6 / nb, best * 1e-3 * 2.4e9 * 1024 / (nb / 64)), flush=True)
(n0, t0, b0), (n1, t1, b1) = res[0], res[-1]
slope = (t1 - t0) / (n1 - n0)
print("per-sphere: %.3f ms per sphere => %.1f SIMD-cycles per wave sphere-test; intercept %.2f ms (%.0f cycles/wave-bounce)" % (
    slope, slope * 1e-3 * 2.4e9 * 1024 / (b0 / 64), t0 - slope * n0, (t0 - slope * n0) * 1e-3 * 2.4e9 * 1024 / (b0 / 64)))
