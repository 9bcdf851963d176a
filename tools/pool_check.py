"""First-light / A-B script for the material-sorted pool kernel (csrc/spt_pool.hip): parity against the oracle on small
cases, then timing against the megakernel (variant bit 10 forces it) on the headline workload.  Every launch runs under
the kernel watchdog so that a scheduling bug reports an error instead of hanging the box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import optix_test_smallpt_amd as pkg  # noqa: E402
import oracle_binding as orc  # noqa: E402

FORCE_MEGA = 0x400


def pool_report(r, st):
    d = r.diag()
    it, ln = d[0:3], d[3:6]
    tot_it, tot_ln = sum(it), sum(ln)
    tl = ""
    if d[10]:
        nw = st["grid_blocks"] * 4
        tl = (f" | timeline (wave-local): longest wave {d[10] / 2.4e6:.2f} ms (at 2.4 GHz ticks), mean wave {d[13] / nw / 2.4e6:.2f} ms, "
              f"after its queue-dry point a wave runs {d[12] / nw / 2.4e6:.2f} ms on average, {d[11] / 2.4e6:.2f} ms at most")
    return tl + " " + (f"batches GEN/DIFF/REFR {it} lanes/batch "
            f"{[round(l / max(i, 1), 1) for l, i in zip(ln, it)]} overall fill {tot_ln / max(tot_it, 1) / 64:.3f} "
            f"full batches {d[9] / max(tot_it, 1):.3f} tail batches {d[7] / max(tot_it, 1):.3f} tail fill {d[8] / max(d[7], 1) / 64:.3f} "
            f"pre-tail fill {(tot_ln - d[8]) / max(tot_it - d[7], 1) / 64:.3f} "
            f"| pending-child records {d[14]} ({d[14] / max(st['samples'], 1):.4f} per sample) | bounces shaded in solo mode {d[21]}"
            + (f" | wave with the longest drain: {(d[22] >> 40) * 256 / 2.4e6:.2f} ms, {(d[22] >> 20) & 0xFFFFF} batches after dry, {d[22] & 0xFFFFF} lanes, last task {d[23] & 0xFFFFFFFF}" if d[22] else ""))


def main():
    quick = "--quick" in sys.argv
    r = pkg.Renderer(0)
    r.set_watchdog(20.0)
    ok = True
    cases = [("cornell9", pkg.cornell9(), 64, 48, 2, 0), ("cornell9", pkg.cornell9(), 37, 53, 5, 7),
             ("cornell9_e12", pkg.cornell9(12.0), 50, 20, 3, 2), ("rand16", pkg.random_spheres(16, 1), 80, 60, 4, 1),
             ("rand24", pkg.random_spheres(24, 5), 40, 30, 2, 3), ("rand10", pkg.random_spheres(10, 5), 40, 30, 2, 3),
             ("single", pkg.make_spheres([(10, (50, 40.8, 81.6), (0, 0, 0), (.75, .25, .25), pkg.DIFF)]), 32, 32, 4, 5),
             ("glass_only", pkg.make_spheres([(1e5, (50, 1e5, 81.6), (.2, .2, .2), (.75, .75, .75), pkg.DIFF),
                                              (16.5, (50, 30, 90), (0, 0, 0), (.999, .999, .999), pkg.REFR),
                                              (600, (50, 681.6 - .27, 81.6), (1, 1, 1), (0, 0, 0), pkg.DIFF)]), 48, 40, 8, 6),
             ("empty", pkg.make_spheres([]), 16, 8, 2, 0), ("cornell9", pkg.cornell9(), 1, 1, 1, 0),
             ("cornell9", pkg.cornell9(), 3, 2, 300, 4), ("cornell9", pkg.cornell9(), 256, 256, 1, 0)]
    for name, sc, w, h, samps, seed in cases:
        r.set_scene(sc)
        r.set_tuning(0, 0)
        try:
            img, st = r.render(w, h, samps, seed=seed, normalise=True)
        except pkg.SptError as e:
            print(f"{name} {w}x{h}x{samps}: ERROR {e}", flush=True)
            ok = False
            break
        kern = r.last_kernel()
        ref, rst = orc.render(sc, w, h, samps, seed=seed, normalise=True)
        exact = bool(np.array_equal(img, ref))
        bo = st["bounces"] == rst["bounces"] and st["max_depth_kills"] == rst["max_depth_kills"]
        print(f"{name} {w}x{h}x{samps} seed {seed}: kernel={kern} bit_exact={exact} bounces_equal={bo} "
              f"({st['bounces']} vs {rst['bounces']}) ndiff={int((img != ref).any(axis=-1).sum())} kernel_ms={st['kernel_ms']:.3f}", flush=True)
        ok &= exact and bo and kern == "pool"
    if not ok:
        print("PARITY FAILED", flush=True)
        return 1
    print("parity ok", flush=True)
    # ---- timing on the headline workload ----
    sc = pkg.cornell9()
    r.set_scene(sc)
    samps = 16 if quick else 256
    res = {}
    for label, variant in (("pool128", 0), ("mega", FORCE_MEGA), ("pool128", 0), ("mega", FORCE_MEGA)):
        r.set_tuning(0, variant)
        img, st = r.render(1024, 768, samps, seed=0, normalise=True)
        res.setdefault(label, []).append(st["kernel_ms"])
        extra = pool_report(r, st) if r.last_kernel() == "pool" else ""
        print(f"{label}: kernel_ms {st['kernel_ms']:.2f}  {st['samples'] / st['kernel_ms'] / 1e3:.0f} Msamples/s  "
              f"bounces/sample {st['bounces'] / st['samples']:.4f} grid {st['grid_blocks']}x{st['block_threads']} {extra}", flush=True)
        res[label + "_img"] = img
    print("images identical:", bool(np.array_equal(res["pool128_img"], res["mega_img"])), flush=True)
    print(f"speedup pool/mega = {min(res['mega']) / min(res['pool128']):.3f}", flush=True)
    r.set_tuning(0, 0)
    _, st = r.render(2048, 1536, samps // 4, seed=0, normalise=True)      # same samples, 4x the tasks: shorter tail
    print(f"pool128 2048x1536 samps/4: kernel_ms {st['kernel_ms']:.2f} {pool_report(r, st)}", flush=True)
    r.set_tuning(0, FORCE_MEGA)
    _, st = r.render(2048, 1536, samps // 4, seed=0, normalise=True)
    print(f"mega 2048x1536 samps/4: kernel_ms {st['kernel_ms']:.2f}", flush=True)
    for per_cu in ():
        r.set_tuning(per_cu, 0)
        _, st = r.render(1024, 768, samps, seed=0, normalise=True)
        print(f"pool blocks/CU {per_cu}: kernel_ms {st['kernel_ms']:.2f}", flush=True)
    r.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
