"""Is the duration of a 4-spp interactive frame set by its longest path?  Same frame with the mirror/glass spheres
(colour .999: roulette survival .999 per bounce) and with both replaced by colour-.75 diffuse spheres."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import optix_test_smallpt_amd as pkg
import torch

r = pkg.Renderer(0)
w, h = 1280, 720
cam = pkg.pinhole_camera(vx=(1, 0, 0), vz=(0, 0, -1), org=(50, 45, 168), near=1.0)
frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
for label in ("cornell9", "all-diffuse"):
    sc = pkg.cornell9()
    if label == "all-diffuse":
        for i in (6, 7):
            sc[i]["color"] = (.75, .75, .75)
            sc[i]["refl"] = pkg.DIFF
    r.set_scene(sc)
    ks = []
    for seed in range(12):
        r.render_rows_device(frame, w, h, 0, h, 1, seed=seed, normalise=False, camera=cam)
        st = r.sync()
        ks.append(st["kernel_ms"])
    print(label, "kernel ms per frame:", " ".join(f"{k:.2f}" for k in ks), f"| bounces/sample {st['bounces'] / st['samples']:.2f} depth-cap kills {st['max_depth_kills']}", flush=True)
