"""Megakernel with its glass-split stack in global memory: workgroups per CU vs kernel time (config 5 and Cornell-9)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import optix_test_smallpt_amd as pkg
import oracle_binding as orc

r = pkg.Renderer(0)
for name, sc, samps in (("rand1024", pkg.random_spheres(1024, 1024), 64), ("cornell9 (forced megakernel)", pkg.cornell9(), 256)):
    r.set_scene(sc)
    for per_cu in (3, 4, 5, 6, 8):
        r.set_tuning(per_cu, 0x400)
        best = 1e9
        for _ in range(2):
            img, st = r.render(1024, 768, samps, seed=0, normalise=True)
            best = min(best, st["kernel_ms"])
        print(f"{name}: blocks/CU {per_cu}: {best:.2f} ms  {st['samples'] / best / 1e3:.1f} Msamples/s grid {st['grid_blocks']} kernel {r.last_kernel()}", flush=True)
    ref, rst = orc.render(sc, 1024, 768, samps, seed=0, normalise=True, row_begin=300, row_count=1)
    print("  row 300 bit-exact:", bool(np.array_equal(img[300:301], ref)), flush=True)
