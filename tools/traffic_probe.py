"""One render of config 2's shape for the PMC traffic passes: which part of the pool kernel's L2<->fabric bytes is the glass-split
stack?  usage (under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): python tools/traffic_probe.py cornell9|noglass [samps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg

which = sys.argv[1] if len(sys.argv) > 1 else "cornell9"
samps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
r = pkg.Renderer(0)
r.set_watchdog(30.0)
sc = pkg.cornell9()
if which == "noglass":
    sc[7]["refl"] = pkg.SPEC                     # the glass sphere becomes a second mirror: no splits, no stack records
r.set_scene(sc)
_, st = r.render(1024, 768, samps, seed=0, normalise=True)
print(which, "kernel_ms", round(st["kernel_ms"], 2), "bounces/sample", round(st["bounces"] / st["samples"], 3), r.last_kernel())
