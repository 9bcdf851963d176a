"""Runs the instrumented (DIAG) kernel build and prints the per-phase share of wave time + lane statistics."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optix_test_smallpt_amd as pkg
r = pkg.Renderer(0); r.set_scene(pkg.cornell9())
samps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
park = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r.set_tuning(0, 0x100 | park)
r.render(1024, 768, samps)
_, st = r.render(1024, 768, samps)
d = r.diag()
names = ["A stack pop", "B task fetch", "C1 gen", "C2 pop ring", "D1 intersect", "D2 shade diff/spec", "D3 glass", "loop top"]
tot = sum(d[:8])
print("kernel_ms %.2f bounces %d" % (st["kernel_ms"], st["bounces"]))
for n, v in zip(names, d[:8]):
    print("  %-20s %6.2f %%   %8.0f clocks/iter" % (n, 100.0 * v / tot, v / max(d[8], 1)))
iters, l1, l2, l3, r3, rc1, lc1 = d[8:15]
print("iterations (wave) %d; lanes in intersect/iter %.1f; lanes shaded (D2 entry)/iter %.1f" % (iters, l1 / iters, l2 / iters))
print("D3 runs/iter %.3f lanes/run %.1f ; C1 runs/iter %.3f lanes/run %.1f" % (r3 / iters, l3 / max(r3, 1), rc1 / iters, lc1 / max(rc1, 1)))
print("total wave-clocks/iter %.0f ; bounces per iteration %.1f" % (tot / iters, st["bounces"] / iters))
