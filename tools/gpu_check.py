"""Dev script (GPU box): parity vs oracle on small cases + timing of the headline config."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

def rel_l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / max(1e-30, np.sqrt((b.astype(np.float64) ** 2).sum())))

r = pkg.Renderer(0)
cases = [("cornell", pkg.cornell9(), 64, 48, 2, 0), ("cornell", pkg.cornell9(), 100, 37, 16, 5),
         ("rand64", pkg.random_spheres(64, 3), 80, 60, 4, 1), ("rand1024", pkg.random_spheres(1024, 1024), 48, 36, 2, 0)]
for name, sc, w, h, samps, seed in cases:
    r.set_scene(sc)
    img, st = r.render(w, h, samps, seed=seed, normalise=True)
    ref, rst = orc.render(sc, w, h, samps, seed=seed, normalise=True)
    print(name, w, h, samps, "rel_l2=%.3e" % rel_l2(img, ref), "exact=", bool((img == ref).all()),
          "mismatch_px=", int((img != ref).any(axis=2).sum()), "bounces", st["bounces"], rst["bounces"], "ms=%.2f" % st["kernel_ms"], flush=True)

if len(sys.argv) > 1:
    samps = int(sys.argv[1])
    r.set_scene(pkg.cornell9())
    for it in range(3):
        img, st = r.render(1024, 768, samps, seed=0, normalise=True)
        print("cornell 1024x768 spp=%d kernel_ms=%.2f Msamples/s=%.1f Bbar=%.3f blocks=%d" % (4 * samps, st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3, st["bounces"] / st["samples"], st["grid_blocks"]), flush=True)
    pkg.write_ppm(os.path.join(ROOT, "gpurun_out", "cornell.ppm"), img)
