"""A/B of pool sizes and workgroups per CU on the headline workload (dev tool)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optix_test_smallpt_amd as pkg

r = pkg.Renderer(0)
r.set_watchdog(30.0)
r.set_scene(pkg.cornell9())


def run(label, per_cu, variant, samps=256):
    r.set_tuning(per_cu, variant)
    best = 1e9
    for _ in range(2):
        _, st = r.render(1024, 768, samps, seed=0, normalise=True)
        best = min(best, st["kernel_ms"])
    d = r.diag()
    it, ln = sum(d[0:3]), sum(d[3:6])
    print(f"{label}: {best:.2f} ms fill {ln / max(it, 1) / 64:.3f} grid {st['grid_blocks']} {st['samples'] / best / 1e3:.0f} Msamples/s", flush=True)


P128 = 3 << 11
for args in sys.argv[1:] or ["128:5", "128:4", "160:4", "160:3", "mega"]:
    if args == "mega":
        run("mega", 0, 0x400)
    else:
        p, x = args.split(":")
        run(f"P{p} x{x}", int(x), P128 if p == "128" else 0)
