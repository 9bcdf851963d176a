"""Wave-local timeline of the pool kernel (its s_memtime counters, tools/pool_check.py pool_report): longest and mean wave, time a
wave keeps running after it found the task queue empty -- config 2 (1024x768, 1024 spp) and one interactive frame (1280x720, 4 spp).
SPT_LIB selects the library build (kernel A/B)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import optix_test_smallpt_amd as pkg
from pool_check import pool_report

r = pkg.Renderer(0)
r.set_watchdog(20.0)
r.set_scene(pkg.cornell9())
for label, w, h, samps, cam in (("config 2", 1024, 768, 256, None),
                                ("interactive frame", 1280, 720, 1, pkg.pinhole_camera(vx=(1, 0, 0), vz=(0, 0, -1), org=(50, 45, 168), near=1.0))):
    for rep in range(3):
        img, st = r.render(w, h, samps, seed=rep, normalise=True, camera=cam)
    print(f"{label}: kernel_ms {st['kernel_ms']:.3f}{pool_report(r, st)}", flush=True)
