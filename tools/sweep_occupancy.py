import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optix_test_smallpt_amd as pkg
r = pkg.Renderer(0); r.set_scene(pkg.cornell9())
samps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
per_cus = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 5, 6, 8]
variants = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for v in variants:
    for per_cu in per_cus:
        r.set_tuning(blocks_per_cu=per_cu, variant=v)
        best = 1e9
        for _ in range(3):
            _, st = r.render(1024, 768, samps)
            best = min(best, st["kernel_ms"])
        print("variant", v, "blocks_per_cu", per_cu, "grid", st["grid_blocks"], "kernel_ms %.2f" % best, "Msamples/s %.0f" % (st["samples"] / best / 1e3), flush=True)
