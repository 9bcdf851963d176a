#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mega path-samples/s at 1024x768 (BASELINE.json metric).

A "step" is one render of the Cornell-9 scene, 1024x768, 1024 spp (BASELINE.json configs[1]) by the
gfx950 megakernel through the C-ABI, with the scene resident in HBM and the framebuffer left in HBM.
N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling -- the image becomes
1024 x (768*N), row-tiled across the ranks (each renders its own 768 rows with global pixel indices),
and every step ends with the RCCL gather of the bands to rank 0.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     -- FP32-VALU roofline of the megakernel (flops/sample = 45 + B*(17N+100),
                  B = bounces executed per sample, counted by the kernel), duration from HIP events
                  recorded on the launch stream inside the library.
  cpu_baseline -- the oracle (CPU port of the same arithmetic) timed on this host's cores on a bounded
                  sample of the same workload.  A reported baseline, not a target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H_PER_GPU, SAMPS = 1024, 768, 256          # 1024 spp = 4 cells x 256 (smallpt.cpp:276,286)
N_SPHERES = 9
PEAK_FP32_TFLOPS = 157.3                      # MI355X_MICROARCH.md: peak FP32 vector, 256 CU x 4 SIMD x 32 lanes x 2 (FMA) x 2.4 GHz
HBM_PEAK_GBPS = 8000.0


def flops_per_sample(bbar, n):
    return 45.0 + bbar * (17.0 * n + 100.0)


def cpu_baseline(pkg, budget_s=15.0):
    """Times the oracle on all host cores: same scene, same 1024x768 image, reduced spp (rate is
    spp-independent), sized from a calibration run to ~budget_s of wall time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as orc
    scene = pkg.cornell9()
    cores = orc.lib().orc_num_threads()
    orc.render(scene, 64, 48, 1, seed=0)                       # spin up the OpenMP team
    t0 = time.perf_counter()
    _, st = orc.render(scene, W, H_PER_GPU, 1, seed=0, normalise=True)       # calibration: 4 spp of the same image
    cal = st["samples"] / (time.perf_counter() - t0)
    samps = int(max(1, min(64, budget_s * cal / (W * H_PER_GPU * 4))))
    t0 = time.perf_counter()
    _, st = orc.render(scene, W, H_PER_GPU, samps, seed=0, normalise=True)
    dt = time.perf_counter() - t0
    return {"value": round(st["samples"] / dt / 1e6, 3), "unit": "Msamples/s", "cores": int(cores), "kind": "port",
            "sample": f"Cornell-9 {W}x{H_PER_GPU} at {4 * samps} spp ({st['samples']} samples, {dt:.1f} s wall, "
                      f"OpenMP dynamic 16-pixel chunks, oracle/smallpt_oracle.c)",
            "bounces_per_sample": round(st["bounces"] / st["samples"], 4)}


def d2h_inclusive(r, samps, steps):
    """spt_render: kernel + finalize + copy of the 9.4 MB float3 image to (pageable) host memory, host wall clock."""
    r.render(W, H_PER_GPU, samps, seed=0, normalise=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        _, st = r.render(W, H_PER_GPU, samps, seed=0, normalise=True)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(st["samples"] / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3),
            "note": "same step via spt_render: includes the D2H copy of the w*h*3 float image to pageable host memory"}


def interactive(pkg, r, dev, frames=600):
    """Render-thread loop of the viewer (smallpt.cpp:895-942) at the reference's window size: Cornell-9 seen by the
    pinhole Camera{vx,vy,vz,org,near=1} placed just inside the box's open front (the smallpt eye point lies outside the
    front-wall sphere; cpuRender pushes its rays 140 units forward, sampleRay does not), 1 sample per jitter cell per frame."""
    w, h, samps = 1280, 720, 1
    cam = pkg.pinhole_camera(vx=(1, 0, 0), vz=(0, 0, -1), org=(50, 45, 168), near=1.0)
    import torch
    prog = pkg.ProgressiveRenderer(r, w, h, samps, camera=cam)
    for _ in range(10):
        prog.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        prog.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    kms, fms = [], []                                  # per-kernel device times of 32 further frames, one at a time
    for _ in range(32):
        prog.step()
        st = r.sync()
        kms.append(st["kernel_ms"]); fms.append(st["finalize_ms"])
    # the same loop with two frames in flight (two contexts/streams, accumulation in frame order: identical accumBuffer)
    prog2 = pkg.ProgressiveRenderer(r, w, h, samps, camera=cam, pipeline=2)
    for _ in range(10):
        prog2.step()
    prog2.flush()
    t0 = time.perf_counter()
    for _ in range(frames):
        prog2.step()
    prog2.flush()
    dt2 = (time.perf_counter() - t0) / frames
    prog2.close()
    # the product's own host: the C++ render loop (host/viewer.cpp, spt_progressive_frame / _frame_async behind the C-ABI) through the CLI
    cpp = {}
    cli = os.path.join(ROOT, "optix-test-smallpt_amd", "host", "smallpt_mi355x")
    if os.path.exists(cli) and not os.environ.get("SPT_BENCH_NO_CPP"):    # (kernel A/B runs select the library with SPT_LIB, which the CLI does not read)
        import subprocess
        for lanes in (1, 2, 4, 8):
            try:
                o = subprocess.run([cli, "4", "--viewer", "--size", f"{w}x{h}", "--org", "50,45,168", "--pipeline", str(lanes), "--bench-frames", str(frames)],
                                   capture_output=True, text=True, timeout=300)
                cpp[f"frames_per_s_cpp_pipeline{lanes}"] = json.loads(o.stdout.strip().splitlines()[-1])["frames_per_s"] if o.returncode == 0 else None
            except Exception:
                cpp[f"frames_per_s_cpp_pipeline{lanes}"] = None
    return {**cpp, "workload": f"Cornell-9, {w}x{h}, 4 spp per frame (1 per jitter cell), pinhole camera + box-in-cell sampling, "
                        f"frame accumulated in HBM, {frames} frames", "frames_per_s": round(1.0 / dt, 1),
            "frames_per_s_two_in_flight": round(1.0 / dt2, 1),
            "kernel_ms": round(sum(kms) / len(kms), 4), "finalize_ms": round(sum(fms) / len(fms), 4),
            "ms_per_frame": round(dt * 1e3, 4), "value": round(w * h * 4 * samps / dt / 1e6, 1), "unit": "Msamples/s"}


KERNEL_NAMES = {"pool": "spt::poolkernel<144,3>", "mega": "spt::megakernel", "grid": "spt::gridkernel<false>", "gpool": "spt::gpoolkernel<false>", "mesh": "spt::meshkernel<0>",
                "sbvh": "spt::meshkernel<2>"}


def _timed_launches(r, render, reps=3):
    """One warm launch + `reps` timed ones; kernel time from the HIP events the library records on the launch stream."""
    render()
    kms, st = [], None
    for _ in range(reps):
        st = render()
        kms.append(st["kernel_ms"])
    return sum(kms) / len(kms), st


def extra_sphere_config(pkg, label, scene, samps, reps=3):
    """A further BASELINE.json configuration beside the headline: same image, same camera, own roofline block.
    Tables that run a grid kernel (spt_gpool.hip, spt_grid.hip) do not test every sphere, so their algorithmic flops use the sphere tests the
    kernel actually executes per closest-hit query (counted by its instrumented build at 16 spp; the count does not depend on the
    spp); the formula's rate -- what the exhaustive loop of smallpt.cpp:54-70 would have to sustain -- is given beside it."""
    import torch
    r = pkg.Renderer(torch.cuda.current_device())
    r.set_watchdog(120.0)
    r.set_scene(scene)
    out_t = torch.empty((H_PER_GPU, W, 3), dtype=torch.float32, device="cuda")

    def render(s=samps):
        r.render_rows_device(out_t, W, H_PER_GPU, 0, H_PER_GPU, s, seed=0, normalise=True)
        return r.sync()

    k_ms, st = _timed_launches(r, render, reps)
    kern = r.last_kernel()
    n = len(scene)
    bbar = st["bounces"] / st["samples"]
    fl_formula = flops_per_sample(bbar, n)
    res = {"workload": f"{label}, {W}x{H_PER_GPU}, {4 * samps} spp, seed 0, smallpt camera + 2x2 tent filter", "spheres": n,
           "kernel": KERNEL_NAMES.get(kern, kern), "launches": reps, "kernel_ms": round(k_ms, 3),
           "value": round(st["samples"] / k_ms / 1e3, 1), "unit": "Msamples/s", "bounces_per_sample": round(bbar, 4)}
    if kern in ("grid", "gpool"):
        r.set_tuning(0, 0x100)                                   # instrumented build: walk statistics (both kernels: [0] cell steps, [1] sphere tests, [4] exhaustive-loop rays)
        st4 = render(4)
        d = r.diag()
        r.set_tuning(0, 0)
        rays = max(1, st4["bounces"])
        radii = sorted(abs(float(x["radius"])) for x in scene)
        always = sum(1 for v in radii if v > 16.0 * radii[n // 2])      # spheres more than 16 x the median radius are tested for every ray
        tests = d[1] / rays + min(always, 1024)
        fl = 45.0 + bbar * (17.0 * tests + 100.0)
        ach = st["samples"] * fl / (k_ms * 1e-3) / 1e12
        # HBM-side bytes per launch of this kernel on this configuration, from the committed PMC passes (constants, not measured in this run)
        traffic, prov = None, None
        pj_path = os.path.join(ROOT, "profiles", "pmc_config5.json")
        if n == 1024 and samps == 256 and os.path.exists(pj_path):
            try:
                pj = json.load(open(pj_path))
                if pj.get("kernel", "").endswith(KERNEL_NAMES.get(kern, kern).split("::")[-1]):
                    traffic = pj.get("hbm_bytes_per_launch")
                    prov = {"file": "profiles/pmc_config5.json", "kernel": pj.get("kernel"), "measured_at_kernel_ms": pj.get("kernel_ms"), "commit": pj.get("commit"),
                            "note": "FETCH_SIZE x 2 + WRITE_SIZE of the committed rocprofv3 --pmc passes: the slots' path state (96 B per slot in global memory, "
                                    "75 MB per launch, beyond the L2) goes through the fabric once per bounce; Infinity-Cache hits are counted"}
            except Exception:
                traffic, prov = None, None
        res["roofline"] = {"bound": "valu", "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
                           "traffic": traffic, "from_committed_profile": prov, "flops_per_sample": round(fl, 1), "sphere_tests_per_query": round(tests, 2),
                           "cell_steps_per_query": round(d[0] / rays, 2), "exhaustive_loop_rays": d[4],
                           "exhaustive_equivalent_tflops": round(st["samples"] * fl_formula / (k_ms * 1e-3) / 1e12, 2),
                           "note": "algorithmic flops = 45 + B (17 T + 100) with T = sphere tests executed per closest-hit query (uniform grid, "
                                   "exhaustive-equivalent by construction); exhaustive_equivalent_tflops prices the same image by SURVEY.md 8(d)'s "
                                   "formula with T = N -- the work the reference's loop would do -- and is not a fraction of any peak"}
    else:
        ach = st["samples"] * fl_formula / (k_ms * 1e-3) / 1e12
        res["roofline"] = {"bound": "valu", "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
                           "traffic": None, "flops_per_sample": round(fl_formula, 1)}
    r.close()
    return res


def extra_mesh_config(pkg, samps, reps=3, bvh=False):
    """The scene the reference ships (smallpt.cpp:31-34: r = 10 diffuse sphere + r = 600 light, 4096 triangles each, scene.h:91), 256x256
    (cpuRender's size, :274-275): through the exhaustive triangle loop (scene.cpp:95-116, what CPUIntersector does) or, bvh=True, through
    the library's default since round 4, the exhaustive-equivalent hierarchy (the OptixIntersector's role, smallpt.cpp:475-603)."""
    import torch
    r = pkg.Renderer(torch.cuda.current_device())
    r.set_mesh_accel(pkg.ACCEL_BVH if bvh else pkg.ACCEL_EXHAUSTIVE)
    meshes = [pkg.make_sphere_trimesh((50, 40.8, 81.6), 10.0), pkg.make_sphere_trimesh((50, 681.6 - .27, 81.6), 600.0)]
    mats = [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((1, 1, 1), (0, 0, 0), pkg.DIFF)]
    r.set_meshes(meshes, mats)
    w = h = 256
    out_t = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")

    def render():
        r.render_rows_device(out_t, w, h, 0, h, samps, seed=0, normalise=True)
        return r.sync()

    k_ms, st = _timed_launches(r, render, reps)
    ntri = sum(m.triangle_count for m in meshes)
    if bvh:
        img = out_t.clone()
        r.set_mesh_accel(pkg.ACCEL_EXHAUSTIVE)
        st_x = render()
        res = {"workload": f"the reference's shipped scene (2 spheres x 4096 triangles), {w}x{h}, {4 * samps} spp, seed 0, hierarchy (SPT_ACCEL_BVH, the default)",
               "triangles": ntri, "kernel": "spt::meshkernel<1>", "launches": reps, "kernel_ms": round(k_ms, 3),
               "value": round(st["samples"] / k_ms / 1e3, 2), "unit": "Msamples/s", "bounces_per_sample": round(st["bounces"] / st["samples"], 4),
               "rays_per_s": round(st["bounces"] / (k_ms * 1e-3) / 1e9, 3), "rays_unit": "Grays/s (closest-hit queries)",
               "bit_identical_to_exhaustive": bool(torch.equal(img, out_t)) and st["bounces"] == st_x["bounces"],
               "roofline": None, "note": "latency-bound pointer chasing (one lane = one ray, dependent 64-byte node loads): no flop or byte roofline is claimed"}
        r.close()
        return res
    tests = st["bounces"] * ntri
    ach = tests * 52.0 / (k_ms * 1e-3) / 1e12                   # 52 flop per triIntersect as the reference writes it (scene.cpp:52-70)
    res = {"workload": f"the reference's shipped scene (2 spheres x 4096 triangles), {w}x{h}, {4 * samps} spp, seed 0, exhaustive triangle loop",
           "triangles": ntri, "kernel": KERNEL_NAMES[r.last_kernel()], "launches": reps, "kernel_ms": round(k_ms, 3),
           "value": round(st["samples"] / k_ms / 1e3, 2), "unit": "Msamples/s", "bounces_per_sample": round(st["bounces"] / st["samples"], 4),
           "roofline": {"bound": "valu", "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
                        "traffic": None, "ray_triangle_tests_per_s": round(tests / (k_ms * 1e-3) / 1e9, 1), "flops_per_test": 52}}
    r.close()
    return res


def extra_mesh_viewer(pkg, frames=100):
    """The viewer's render loop (ProgressiveRenderer, smallpt.cpp:895-942: one sample per jitter cell per frame, pinhole Camera :607-641) on the
    reference's shipped mesh scene at 1280x720, through the library's default (SPT_ACCEL_BVH: the exhaustive loop's Hit for every ray), the
    plain hierarchy (SPT_ACCEL_BVH_FAST) and the exhaustive loop; the accumulators of the first two after 8 frames against the third's."""
    import time
    import torch
    meshes = [pkg.make_sphere_trimesh((50, 40.8, 81.6), 10.0), pkg.make_sphere_trimesh((50, 681.6 - .27, 81.6), 600.0)]
    mats = [((0, 0, 0), (.75, .25, .25), pkg.DIFF), ((1, 1, 1), (0, 0, 0), pkg.DIFF)]
    cam = pkg.pinhole_camera(vz=(0, -0.042573, -0.999093), org=(50, 52, 295.6))
    res = {"workload": "the reference's shipped scene (2 spheres x 4096 triangles), 1280x720, 1 sample per jitter cell per frame, pinhole camera at the "
                       "smallpt camera's position, serial frames", "unit": "frames/s"}
    accs = {}
    for name, accel, n in (("bvh", pkg.ACCEL_BVH, frames), ("bvh_fast", pkg.ACCEL_BVH_FAST, frames), ("exhaustive", pkg.ACCEL_EXHAUSTIVE, max(8, frames // 10))):
        r = pkg.Renderer(torch.cuda.current_device())
        r.set_mesh_accel(accel)
        r.set_meshes(meshes, mats)
        prog = pkg.ProgressiveRenderer(r, 1280, 720, 1, camera=cam)
        for _ in range(8):
            prog.step()
        prog.flush()
        accs[name] = prog.accum.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            prog.step()
        prog.flush()
        torch.cuda.synchronize()
        res[name] = round(n / (time.perf_counter() - t0), 1)
        prog.close(); r.close()
    res["accum_identical_to_exhaustive"] = {k: bool(torch.equal(accs[k], accs["exhaustive"])) for k in ("bvh", "bvh_fast")}
    res["note"] = ("bvh = the default: exhaustive-equivalent for every ray (csrc/spt_tribvh.h); a pinhole frame tests the planes through the camera's origin "
                   "instead of walking the plane tree.  bvh_fast = the plain hierarchy (rays lying in a triangle's plane to rounding may differ)")
    return res


EXTRAS = ("config3", "config5", "mesh_4spp", "mesh_256spp", "mesh_4spp_bvh", "mesh_256spp_bvh", "mesh_viewer")


def run_extra(pkg, name):
    if name == "config3":
        return extra_sphere_config(pkg, "config 3: Cornell-9 (9 spheres)", pkg.cornell9(), 4096)
    if name == "config5":
        return extra_sphere_config(pkg, "config 5: 1024 spheres (6 walls + light + 1017 random, SplitMix64 seed 1024)", pkg.random_spheres(1024, 1024), 256)
    if name == "mesh_4spp":
        return extra_mesh_config(pkg, 1)
    if name == "mesh_256spp":
        return extra_mesh_config(pkg, 64)
    if name == "mesh_4spp_bvh":
        return extra_mesh_config(pkg, 1, bvh=True)
    if name == "mesh_256spp_bvh":
        return extra_mesh_config(pkg, 64, bvh=True)
    if name == "mesh_viewer":
        return extra_mesh_viewer(pkg)
    raise SystemExit(f"unknown extra {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the d2h_inclusive, interactive and extra-configuration measurements (profiling runs)")
    ap.add_argument("--only-extra", choices=EXTRAS, help="run one of the extra configurations alone and print its JSON (profiling runs)")
    ap.add_argument("--samps", type=int, default=SAMPS, help=argparse.SUPPRESS)   # dev only; default = config
    ap.add_argument("--variant", type=lambda v: int(v, 0), default=0, help=argparse.SUPPRESS)   # dev only: kernel A/B (csrc/spt_internal.h)
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # the host driver of this pool only supports dmabuf IPC (RCCL needs it)
    import torch
    import torch.distributed as dist
    import optix_test_smallpt_amd as pkg
    from optix_test_smallpt_amd.distributed import FrameAssembler, row_band

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    # SPT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices and the
    # gather goes through host memory); the driver's runs use the default, nccl = RCCL over xGMI.
    backend = os.environ.get("SPT_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if args.only_extra:
        print(json.dumps({args.only_extra: run_extra(pkg, args.only_extra)}), flush=True)
        return
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    h = H_PER_GPU * world
    # N > 1: the rows are dealt out round-robin in blocks of 16 rows (every rank renders 768 of the 768*N rows): contiguous
    # bands of this image differ by up to 1.24x in cost (tools/probe_band_balance.py), interleaved blocks do not
    interleave = 16 if world > 1 else 0
    begin, count = row_band(h, world, rank)
    samps = args.samps
    r = pkg.Renderer(dev_index)
    r.set_scene(pkg.cornell9())
    if args.variant:
        r.set_tuning(0, args.variant)
    # destination memory of the exchange: rank 0 owns the whole framebuffer and renders its band in place, the other
    # ranks render into their band tensor; the bands are received straight into the framebuffer's row slices
    fa = FrameAssembler(W, h, device=dev if backend == "nccl" else "cpu", interleave=interleave)
    count = fa.count
    band = fa.band if backend == "nccl" else torch.empty((count, W, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    timing = {"render_s": [], "gather_s": []}                 # per step on this rank (N > 1: reported per rank beside the line)
    seed_of = {"next": 0}

    def step(fixed_seed=None):
        # Every step renders the same view with the NEXT seed (seed = step index, warm-up steps included): the render loop of
        # smallpt.cpp:895-942 passes its frame counter as the seed.  (Round 3 repeated seed 0: the library then dispatches the task
        # chunks in the order of their cost in the previous launch -- an exact prediction for the same seed, 76.3 instead of 79.4 ms,
        # and no use to anyone who wants a new image.  With a new seed the library runs its static order, as for a view's first launch.)
        if fixed_seed is None:
            seed = seed_of["next"]; seed_of["next"] += 1
        else:
            seed = fixed_seed
        t_a = time.perf_counter()
        # N > 1: every rank completes (or fails) its rows, the ranks agree on that (FrameAssembler.all_ok: one 1-element
        # all_reduce), and only then enter the point-to-point exchange -- a rank whose render failed would otherwise leave the
        # others waiting for rows that never come.  The agreement costs ~0.1 ms of an 80 ms step and is inside the timed region.
        st, err = None, None
        try:
            if interleave:
                r.render_interleaved_device(band, W, h, interleave, world, rank, samps, seed=seed, normalise=True, stream=stream)
            else:
                r.render_rows_device(band, W, h, begin, count, samps, seed=seed, normalise=True, stream=stream)
            st = r.sync()
        except pkg.SptError as e:
            if world == 1:
                raise
            err = e
        t_b = time.perf_counter()
        timing["render_s"].append(t_b - t_a)
        if world == 1:
            return band, st
        if backend != "nccl" and err is None:
            fa.band.copy_(band)          # rehearsal transport: through host memory
        try:
            img = fa.gather(ok=err is None)
            torch.cuda.synchronize()
            timing["gather_s"].append(time.perf_counter() - t_b)
            return img, st
        except RuntimeError:
            raise SystemExit(f"rank {rank}: {err if err is not None else 'the render failed on another rank'}")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # communicator set-up (RCCL channels over xGMI are created lazily on the first collective) stays outside
        # the timed region even with --warmup 0
        fa.gather()
        fa.all_ok(True)              # ... and so does the all-reduce of the status agreement
    # (the first warm-up step is the process's first launch: it includes the device's warm-up and is reported beside the timed steps)
    first_ms = None
    for i in range(args.warmup):
        _, st = step()
        if i == 0 and st is not None:
            first_ms = st["kernel_ms"]
    fence()
    t0 = time.perf_counter()
    kms, fms, bounces, samples = [], [], 0, 0
    for _ in range(args.steps):
        _, st = step()
        kms.append(st["kernel_ms"]); fms.append(st["finalize_ms"])
        bounces += st["bounces"]; samples += st["samples"]
    fence()
    elapsed = time.perf_counter() - t0
    # N > 1: what every rank spent where, for rank 0's line (the first hardware run of the exchange should be readable from one line)
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "device": dev_index, "rows": int(count), "samples_per_step": int(count * W * 4 * samps),
                "render_ms": round(1e3 * sum(timing["render_s"][-args.steps:]) / max(1, args.steps), 3),
                "gather_ms": round(1e3 * sum(timing["gather_s"][-args.steps:]) / max(1, len(timing["gather_s"][-args.steps:])), 3) if timing["gather_s"] else None,
                "kernel_ms": round(sum(kms) / len(kms), 3), "bounces_per_sample": round(bounces / max(1, samples), 4)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    tot = torch.tensor([elapsed, float(samples), float(bounces), sum(kms) / len(kms)], dtype=torch.float64,
                       device=dev if backend == "nccl" else "cpu")
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, samples, bounces = float(mx[0]), float(sm[1]), float(sm[2])
    # Beside the timed steps, outside the timed region: the SAME seed rendered again and again, which the library's cost-ordered dispatch
    # turns into its best case (each launch starts the chunks that were expensive in the previous one first; a launch records only when it repeats its predecessor, so the order starts with the third launch and converges over 2-3 more).
    rerender = None
    if world == 1:
        ks = []
        for _ in range(5):                       # 1: static, 2: static + records (it repeats its predecessor), 3-5: ordered
            _, st1 = step(fixed_seed=0xC0FFEE)
            ks.append(st1["kernel_ms"])
        rerender = {"kernel_ms_per_launch": [round(k, 3) for k in ks], "kernel_ms": round(ks[-1], 3),
                    "value": round(st1["samples"] / (ks[-1] + st1["finalize_ms"]) / 1e3, 2), "unit": "Msamples/s (kernel + finalize)",
                    "note": "the same view AND seed re-rendered: launches 2+ dispatch their task chunks in the cost order of the previous one "
                            "(profiles/r04_cost_order_seeds.txt); not `value`, which steps the seed"}
    if rank == 0:
        value = samples / elapsed / 1e6
        # roofline of the dominant kernel (megakernel) on rank 0, per launch
        k_s = (sum(kms) / len(kms)) * 1e-3
        my_samples = count * W * 4 * samps
        bbar = (st["bounces"] / st["samples"])
        fl = flops_per_sample(bbar, N_SPHERES)
        achieved = my_samples * fl / k_s / 1e12
        f_s = (sum(fms) / len(fms)) * 1e-3
        nb = 8 if samps >= 128 else (4 if samps >= 64 else (2 if samps >= 32 else 1))   # D9 sample blocks per jitter cell
        # PMC-derived constants of the same command (tools/prof_round.sh -> profiles/pmc_latest.json): HBM-side bytes per launch
        # and the VALU instruction count / lane utilisation used for the four-factor decomposition of `frac`
        traffic, decomposition, provenance = None, None, None
        prof = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(prof) and world == 1 and samps == SAMPS:
            try:
                pj = json.load(open(prof))
                traffic = pj.get("hbm_bytes_per_launch")
                insts, util = float(pj["valu_wave_insts_per_launch"]), float(pj["valu_lane_utilisation"])
                simd_cycles = k_s * 2.4e9 * 1024                       # 256 CUs x 4 SIMDs at the 2.4 GHz the peak is quoted for
                decomposition = {
                    "no_fma": 0.5,                                      # 1 flop per lane-op (non-contracted mul/add) vs 2 for FMA
                    "issue": round(2.0 * insts / simd_cycles, 4),       # 2 clk per wave-instruction at full rate / measured clk per instruction
                    "lane_utilisation": round(util, 4),                 # PMC VALUUtilization (active lanes / 64)
                    "algorithmic_share": round(my_samples * fl / (insts * 64.0 * util), 4),   # algorithmic flops per executed lane-op
                    "valu_wave_insts_per_lane_bounce": round(insts * 64.0 / (st["bounces"] * 64.0), 3),
                    "source": pj.get("_source", "profiles/pmc_latest.json")}
                provenance = {"file": "profiles/pmc_latest.json", "kernel": pj.get("kernel"), "measured_at_kernel_ms": pj.get("kernel_ms"),
                              "commit": pj.get("commit"), "note": "traffic and the instruction count / lane utilisation behind `decomposition` are constants of "
                              "the committed rocprofv3 --pmc passes of this command, not measured in this run; only kernel_ms is live"}
            except Exception:
                traffic, decomposition, provenance = None, None, None
        out = {
            "metric": "Mega path-samples/sec at 1024x768", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "rerender_same_seed": rerender,
            "config": {"workload": f"Cornell-9 (9 spheres), {W}x{H_PER_GPU} per GPU, {4 * samps} spp, seed = step index (a new seed every step), "
                                   f"smallpt camera + 2x2 tent filter; image {W}x{h} row-tiled over {world} GPU(s)"
                                   + (f" (rows dealt out round-robin in blocks of {interleave})" if interleave else "")
                                   + (f", {'RCCL' if backend == 'nccl' else backend} gather to rank 0 each step" if world > 1 else "")
                                   + "; every step is a view's launch with a new seed = the library's static dispatch order (what a one-shot "
                                     "render and a progressive loop get); rerender_same_seed = the cost-ordered best case beside it",
                       "spheres": N_SPHERES, "width": W, "height": h, "spp": 4 * samps, "rows_per_gpu": count},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                         "kernel": KERNEL_NAMES.get(r.last_kernel(), r.last_kernel()), "kernel_ms": round(k_s * 1e3, 3),
                         "kernel_ms_first_launch_of_the_process": None if first_ms is None else round(first_ms, 3),
                         "kernel_ms_per_step": [round(k, 3) for k in kms],
                         "flops_per_sample": round(fl, 1), "bounces_per_sample": round(bbar, 4),
                         "decomposition": decomposition, "from_committed_profile": provenance,
                         "note": "FP32 VALU-bound (no MFMA-shaped work, HBM traffic ~12 B/pixel/launch); algorithmic "
                                 "flops per SURVEY.md 8(d); arithmetic is non-contracted IEEE mul/add (1 flop/instr) "
                                 "for bit-parity with the reference's host arithmetic, so frac <= 0.5 by construction",
                         "hbm_store_kernel": {"kernel": "spt::finalize", "ms": round(f_s * 1e3, 4),
                                              "achieved_GBps": round(count * W * (64 * nb + 12) / f_s / 1e9, 1),
                                              "bytes_per_pixel": 64 * nb + 12,
                                              "peak_GBps": HBM_PEAK_GBPS,
                                              "note": "reads the block sums the path kernel has just written (0.4 GB: Infinity-Cache "
                                                      "assisted, not a pure HBM rate) and stores the 12 B/pixel image"}},
        }
        if world == 1 and not args.no_extras:
            # Outside the timed region: (1) the same step through spt_render, i.e. INCLUDING the framebuffer D2H copy into
            # host memory ("kernel + framebuffer D2H/gather"); never `value`, reported beside it.
            # (2) the reference's live use (smallpt.cpp:844-846,922): 1280x720, 1 sample per jitter cell per frame,
            # pinhole Camera + box-in-cell sampling, device-resident accumulation -- frames/s of the render-thread loop.
            out["d2h_inclusive"] = d2h_inclusive(r, samps, max(1, min(10, args.steps)))
            out["interactive"] = interactive(pkg, r, dev)
            # (3) the other single-GPU configurations of BASELINE.json and the reference's shipped triangle scene: one warm + three
            # timed launches each, kernel time from the library's HIP events, each with its own roofline block
            out["extras"] = {name: run_extra(pkg, name) for name in EXTRAS}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
