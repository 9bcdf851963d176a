"""Interleaved A/B timing of tuning variants in one process (guide rule 24): min and median kernel ms.
usage: python tools/ab.py <scene: cornell|rand1024|rand64> <samps> <variant,variant,...> [rounds] [blocks_per_cu]"""
import sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optix_test_smallpt_amd as pkg
scene = {"cornell": pkg.cornell9, "rand1024": lambda: pkg.random_spheres(1024, 1024), "rand64": lambda: pkg.random_spheres(64, 3)}[sys.argv[1]]()
samps = int(sys.argv[2])
variants = [int(x) for x in sys.argv[3].split(",")]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
per_cu = int(sys.argv[5]) if len(sys.argv) > 5 else 0
r = pkg.Renderer(0); r.set_scene(scene)
times = {v: [] for v in variants}
r.render(1024, 768, samps)
for _ in range(rounds):
    for v in variants:
        r.set_tuning(per_cu, v)
        _, st = r.render(1024, 768, samps)
        times[v].append(st["kernel_ms"])
for v in variants:
    t = times[v]
    print("variant %3d: min %.2f ms  median %.2f ms  -> %.0f Msamples/s (min)  Bbar %.3f" % (v, min(t), statistics.median(t), st["samples"] / min(t) / 1e3, st["bounces"] / st["samples"]), flush=True)
