"""Cost-ordered dispatch of the pool kernel (spt_api.cpp): config 2 in the static order (tuning bit 13), with the HOT chunks first (default
since round 4) and with ALL chunks sorted by cost (round 3's order, tuning bit 14) -- re-rendering ONE seed (the previous launch is an exact
prediction) and stepping the seed every launch (a progressive loop; what bench.py times).  One process, alternating.
usage: ab_order.py [samps per cell]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import optix_test_smallpt_amd as pkg

r = pkg.Renderer(0)
r.set_scene(pkg.cornell9())
t = torch.empty((768, 1024, 3), dtype=torch.float32, device="cuda")
samps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r.render_rows_device(t, 1024, 768, 0, 768, samps, seed=99, normalise=True); r.sync()      # warm the device
for rnd in range(2):
    for label, variant in (("static order", 0x2000), ("hot chunks first", 0), ("all chunks by cost", 0x4000)):
        for seeds, stepping in (("one seed", False), ("new seed every launch", True)):
            r.set_tuning(0, variant)
            ks = []
            for i in range(7):
                r.render_rows_device(t, 1024, 768, 0, 768, samps, seed=(100 * rnd + i) if stepping else 0, normalise=True); st = r.sync(); ks.append(st["kernel_ms"])
            later = ks[2:]
            print(f"{label:20s} {seeds:22s}: kernel_ms first {ks[0]:.2f} second {ks[1]:.2f} then {[round(k, 2) for k in later]} mean {sum(later) / len(later):.2f} "
                  f"-> {st['samples'] / (sum(later) / len(later)) / 1e3:.0f} Msamples/s", flush=True)
r.set_tuning(0, 0)
