"""Host-side mirror of the reference's render interface on top of the C-ABI.

``Renderer.render`` plays the role of ``Renderer::render`` (smallpt.cpp:692-814, un-normalised sum,
row 0 = bottom) and ``Renderer.cpu_render_equivalent`` that of ``cpuRender`` (smallpt.cpp:269-379,
normalised).  PyTorch is only used for device memory / streams when the caller wants the image to
stay resident in HBM.
"""
import ctypes as C

import numpy as np

from ._lib import SptCamera, SptMaterial, SptMesh, SptMultiStats, SptStats, load_library, load_multi_library
from .scene import HIT_DTYPE, RAY_DTYPE, SPHERE_DTYPE

FLAG_NORMALISE = 1
FLAG_ONE_SHOT = 2          # scheduling only: no dispatch order used or recorded for this launch (include/smallpt_mi355x.h)
ACCEL_EXHAUSTIVE, ACCEL_BVH, ACCEL_GRID, ACCEL_BVH_FAST, ACCEL_AUTO = 0, 1, 2, 3, 4


class SptError(RuntimeError):
    pass


def smallpt_camera(w, h):
    """Camera constants of cpuRender (smallpt.cpp:277-279) for a w x h image."""
    lib = load_library()
    cam = SptCamera()
    if lib.spt_camera_smallpt(w, h, C.byref(cam)):
        raise SptError("spt_camera_smallpt failed")
    return cam


def pinhole_camera(vx=(1, 0, 0), vy=None, vz=(0, 0, -1), org=(0, -1, 0), near=1.0):
    """Camera{vx, vy, vz, org, nearPlaneDistance} of the interactive driver (smallpt.cpp:607-624); the defaults
    are main()'s values (:885-888,899), vy = normalize(cross(vx, vz))."""
    lib = load_library()
    f = np.float32
    if vy is None:
        a, b = np.asarray(vx, dtype=f), np.asarray(vz, dtype=f)
        c = np.array([f(a[1] * b[2]) - f(a[2] * b[1]), f(a[2] * b[0]) - f(a[0] * b[2]), f(a[0] * b[1]) - f(a[1] * b[0])], dtype=f)
        q = f(f(c[0] * c[0]) + f(c[1] * c[1])) + f(c[2] * c[2])
        vy = c * (f(1) / np.sqrt(f(q)))
    cam = SptCamera()
    v3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    if lib.spt_camera_pinhole(v3(vx), v3(vy), v3(vz), v3(org), float(near), C.byref(cam)):
        raise SptError("spt_camera_pinhole failed")
    return cam


def _stats_dict(st):
    return {"samples": int(st.samples), "bounces": int(st.bounces), "max_depth_kills": int(st.max_depth_kills),
            "kernel_ms": float(st.kernel_ms), "finalize_ms": float(st.finalize_ms), "total_ms": float(st.total_ms),
            "grid_blocks": int(st.grid_blocks), "block_threads": int(st.block_threads)}


class Renderer:
    """One context = one HIP device (spt_create)."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        h = C.c_void_p()
        if self._lib.spt_create(int(device_id), C.byref(h)):
            raise SptError(self._lib.spt_last_error(None).decode())
        self._h = h
        self.device_id = int(device_id)
        self._scene = None
        # everything a second context needs to render the same frames (ProgressiveRenderer(pipeline=2) replays it on its extra
        # lane): the current scene (sphere table or meshes + materials), the closest-hit modes, tuning and watchdog
        self._state = {"scene": None, "sphere_accel": None, "mesh_accel": None, "tuning": None, "watchdog": None}
        self._state_version = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.spt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc:
            raise SptError(self._lib.spt_last_error(self._h).decode())

    def set_scene(self, spheres):
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        self._scene = spheres
        self._check(self._lib.spt_set_scene(self._h, spheres.ctypes.data_as(C.c_void_p), len(spheres)))
        self._state["scene"] = ("spheres", spheres)
        self._state_version += 1

    def replay_state_on(self, other):
        """Brings another context (same device) to this one's scene, closest-hit modes, tuning and watchdog."""
        st = self._state
        if st["tuning"] is not None:
            other.set_tuning(*st["tuning"])
        if st.get("grid_pools") is not None:
            other.set_grid_pools(*st["grid_pools"])
        if st["watchdog"] is not None:
            other.set_watchdog(st["watchdog"])
        if st["sphere_accel"] is not None:
            other.set_sphere_accel(st["sphere_accel"])
        if st["mesh_accel"] is not None:
            other.set_mesh_accel(st["mesh_accel"])
        if st["scene"] is not None:
            if st["scene"][0] == "spheres":
                other.set_scene(st["scene"][1])
            else:
                other.set_meshes(st["scene"][1], st["scene"][2])

    def set_meshes(self, meshes, materials):
        """Intersector::addTriangleMesh for every TriMesh + build() (smallpt.cpp:437-447); materials[i] = (emission, color,
        refl) of instance i (smallpt.cpp:170).  Makes the mesh scene current for render()."""
        ms = (SptMesh * max(1, len(meshes)))()
        mats = (SptMaterial * max(1, len(meshes)))()
        self._mesh_keepalive = list(meshes)
        for i, (m, (e, col, refl)) in enumerate(zip(meshes, materials)):
            ms[i].positions = m.positions.ctypes.data
            ms[i].normals = m.normals.ctypes.data
            ms[i].indices = m.indices.ctypes.data
            ms[i].nverts, ms[i].ntris = len(m.positions), len(m.indices)
            mats[i].emission = (C.c_float * 3)(*[float(v) for v in e])
            mats[i].color = (C.c_float * 3)(*[float(v) for v in col])
            mats[i].refl = int(refl)
        self._check(self._lib.spt_set_meshes(self._h, ms, len(meshes), mats))
        self._scene = None                                   # the sphere table is no longer the current scene
        self._state["scene"] = ("meshes", list(meshes), list(materials))
        self._state_version += 1

    def set_sphere_accel(self, accel):
        """How sphere tables above 24 spheres find their closest hit: ACCEL_GRID (default: uniform grid in LDS), ACCEL_BVH (hierarchy) --
        both exhaustive-equivalent by construction (include/smallpt_mi355x.h) -- or ACCEL_EXHAUSTIVE."""
        self._check(self._lib.spt_set_sphere_accel(self._h, int(accel)))
        self._state["sphere_accel"] = int(accel)
        self._state_version += 1

    def set_mesh_accel(self, accel):
        """ACCEL_AUTO (default: the faster of the two exact modes per launch), ACCEL_BVH (the role of the reference's OptiX Prime model,
        smallpt.cpp:475-603; the exhaustive loop's Hit for every ray, include/smallpt_mi355x.h), ACCEL_EXHAUSTIVE (every triangle, the reference's CPU loops: the parity anchor) or
        ACCEL_BVH_FAST (the plain hierarchy of rounds 2-3: several times faster, but rays lying in a triangle's plane to rounding may differ)."""
        self._check(self._lib.spt_set_mesh_accel(self._h, int(accel)))
        self._state["mesh_accel"] = int(accel)
        self._state_version += 1

    def trace_rays(self, rays):
        """Intersector::traceRays (smallpt.cpp:460-470): rays = array of RAY_DTYPE (or (n, 6) floats); returns HIT_DTYPE[n]."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6) if not (hasattr(rays, "dtype") and rays.dtype == RAY_DTYPE) else rays
        rays = np.ascontiguousarray(rays)
        n = len(rays)
        hits = np.zeros(n, dtype=HIT_DTYPE)
        self._check(self._lib.spt_trace_rays(self._h, rays.ctypes.data_as(C.c_void_p), n, hits.ctypes.data_as(C.c_void_p)))
        return hits

    def trace_rays_device(self, rays_t, hits_t=None, stream=None):
        """spt_trace_rays_device: rays_t = contiguous float32 CUDA tensor (n, 6) on this renderer's device; returns the float32 tensor
        (n, 11) of Hit records (dist, instId and triId as raw bits, x, n, uv), enqueued on `stream` (a torch stream; default: current)."""
        import torch
        assert rays_t.is_cuda and rays_t.dtype == torch.float32 and rays_t.is_contiguous() and rays_t.shape[-1] == 6
        n = rays_t.numel() // 6
        if hits_t is None:
            hits_t = torch.empty((n, 11), dtype=torch.float32, device=rays_t.device)
        st = (stream if stream is not None else torch.cuda.current_stream(rays_t.device)).cuda_stream
        self._check(self._lib.spt_trace_rays_device(self._h, C.c_void_p(rays_t.data_ptr()), n, C.c_void_p(hits_t.data_ptr()), C.c_void_p(st)))
        return hits_t

    def set_tuning(self, blocks_per_cu=0, variant=0):
        self._check(self._lib.spt_set_tuning(self._h, blocks_per_cu, variant))
        self._state["tuning"] = (int(blocks_per_cu), int(variant))
        self._state_version += 1

    def set_grid_pools(self, lane_owned=False, slots=0, ready=0, drain=0, min_batch=0, walk_iters=0):
        """Large sphere tables (csrc/spt_internal.h spt_set_grid_pools): keep the lane-owned grid kernel, or set the pool geometry of the
        default one (0 = default).  Results never depend on it."""
        self._check(self._lib.spt_set_grid_pools(self._h, int(lane_owned), slots, ready, drain, min_batch, walk_iters))   # (2: lane-owned with the tables in global memory, A/B)
        self._state["grid_pools"] = (bool(lane_owned), slots, ready, drain, min_batch, walk_iters)
        self._state_version += 1

    def render(self, w, h, samps_per_cell, seed=0, normalise=False, camera=None):
        """Full image to host memory: (h, w, 3) float32, row 0 = bottom.  Returns (image, stats)."""
        cam = camera if camera is not None else smallpt_camera(w, h)
        out = np.empty((h, w, 3), dtype=np.float32)
        st = SptStats()
        self._check(self._lib.spt_render(self._h, C.byref(cam), w, h, samps_per_cell, seed,
                                         FLAG_NORMALISE if normalise else 0,
                                         out.ctypes.data_as(C.c_void_p), C.byref(st)))
        return out, _stats_dict(st)

    def cpu_render_equivalent(self, w, h, spp, seed=0):
        """cpuRender(argv[1]=spp) semantics: samps = spp/4 (smallpt.cpp:276), normalised image."""
        return self.render(w, h, max(1, int(spp) // 4), seed=seed, normalise=True)

    def render_rows_device(self, out_tensor, w, h, row_begin, row_count, samps_per_cell, seed=0,
                           normalise=False, camera=None, stream=None):
        """Enqueues the render of rows [row_begin, row_begin+row_count) into ``out_tensor`` (a CUDA/HIP
        float32 torch tensor with row_count*w*3 elements on this context's device).  Asynchronous:
        call ``sync()`` for completion + statistics.  ``stream``: a raw hipStream_t handle (int), e.g.
        ``torch.cuda.current_stream().cuda_stream``; None = the context's own stream."""
        if out_tensor.numel() != row_count * w * 3 or not out_tensor.is_contiguous():
            raise ValueError("out_tensor must be contiguous with row_count*w*3 float32 elements")
        if str(out_tensor.dtype) != "torch.float32" or out_tensor.device.type != "cuda":
            raise ValueError("out_tensor must be a float32 tensor on the GPU")
        cam = camera if camera is not None else smallpt_camera(w, h)
        self._check(self._lib.spt_render_rows_device(
            self._h, C.byref(cam), w, h, row_begin, row_count, samps_per_cell, seed,
            FLAG_NORMALISE if normalise else 0, C.c_void_p(out_tensor.data_ptr()),
            C.c_void_p(stream) if stream else None))

    def set_watchdog(self, seconds):
        """Pool kernel: a launch whose waves run longer than this fails in sync() instead of hanging (0 = off)."""
        self._check(self._lib.spt_set_watchdog(self._h, float(seconds)))
        self._state["watchdog"] = float(seconds)
        self._state_version += 1

    def last_kernel(self):
        """'pool' (spt_pool.hip, material-sorted), 'mega' (spt_kernel.hip), 'mesh' (spt_mesh.hip, triangles), 'sbvh' (spt_mesh.hip over a
        sphere hierarchy), 'gpool' (spt_gpool.hip, uniform grid over a large sphere table driven by wave-private path pools: the default above 24
        spheres) or 'grid' (spt_grid.hip, the same grid with lanes that own their path: tables that leave no LDS for the pools) for the last launch."""
        return {0: "mega", 1: "pool", 2: "mesh", 3: "sbvh", 4: "grid", 5: "gpool", 6: "mesh_bvh", 7: "mesh_bvh_fast"}[self._lib.spt_last_kernel(self._h)]

    def render_interleaved_device(self, out_tensor, w, h, block_rows, world, rank, samps_per_cell, seed=0,
                                  normalise=False, camera=None, stream=None):
        """Like render_rows_device for the rows of rank `rank` when the image is dealt out to `world` ranks round-robin in
        blocks of `block_rows` rows (spt_render_interleaved_device); out_tensor holds those rows packed in ascending order."""
        rows = int(self._lib.spt_interleaved_row_count(h, block_rows, world, rank))
        if out_tensor.numel() != rows * w * 3 or not out_tensor.is_contiguous():
            raise ValueError(f"out_tensor must be contiguous with {rows}*w*3 float32 elements")
        if str(out_tensor.dtype) != "torch.float32" or out_tensor.device.type != "cuda":
            raise ValueError("out_tensor must be a float32 tensor on the GPU")
        cam = camera if camera is not None else smallpt_camera(w, h)
        self._check(self._lib.spt_render_interleaved_device(
            self._h, C.byref(cam), w, h, block_rows, world, rank, samps_per_cell, seed,
            FLAG_NORMALISE if normalise else 0, C.c_void_p(out_tensor.data_ptr()), C.c_void_p(stream) if stream else None))

    def diag(self):
        """Phase timings / lane counters of the last launch of the instrumented build (variant bit 8)."""
        arr = (C.c_uint64 * 24)()
        self._check(self._lib.spt_diag(self._h, C.byref(arr)))
        return [int(v) for v in arr]

    def chunk_order(self):
        """The pool kernel's chunk order for the next launch of the same view (spt_chunk_order_snapshot): a permutation of the
        64-task chunks, most expensive first; empty when the last launch recorded none."""
        cap = 1 << 22
        buf = np.empty(cap, dtype=np.uint32)
        n = C.c_uint32(0)
        self._check(self._lib.spt_chunk_order_snapshot(self._h, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return buf[:n.value].copy()

    def selftest_math(self, op, x, w=1024):
        """Runs device helper `op` over the float32 array x (see spt_selftest_math)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self._check(self._lib.spt_selftest_math(self._h, int(op), x.ctypes.data_as(C.c_void_p),
                                                out.ctypes.data_as(C.c_void_p), x.size, int(w)))
        return out

    def sync(self):
        st = SptStats()
        self._check(self._lib.spt_sync(self._h, C.byref(st)))
        return _stats_dict(st)


class MultiRenderer:
    """spt_multi_* (include/smallpt_mi355x_multi.h): ONE process, one host thread + context per device, row bands,
    RCCL exchange into the root device's framebuffer.  `self_exchange` routes a single device's band through RCCL too
    (rehearsal of the exchange step on a one-GPU box)."""

    SELF_EXCHANGE, CONTIGUOUS, COPY_EXCHANGE = 1, 2, 4

    def __init__(self, device_ids=(0,), self_exchange=False, contiguous=False, copy_exchange=False):
        self._lib = load_multi_library()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        flags = (self.SELF_EXCHANGE if self_exchange else 0) | (self.CONTIGUOUS if contiguous else 0) | (self.COPY_EXCHANGE if copy_exchange else 0)
        if self._lib.spt_multi_create(ids, len(device_ids), flags, C.byref(h)):
            raise SptError(self._lib.spt_multi_last_error(None).decode())
        self._h = h
        self.device_ids = tuple(int(d) for d in device_ids)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.spt_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise SptError(self._lib.spt_multi_last_error(self._h).decode())

    def set_scene(self, spheres):
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        self._check(self._lib.spt_multi_set_scene(self._h, spheres.ctypes.data_as(C.c_void_p), len(spheres)))

    def set_meshes(self, meshes, materials):
        """spt_multi_set_meshes: the triangle scene on every device (see Renderer.set_meshes)."""
        ms = (SptMesh * max(1, len(meshes)))()
        mats = (SptMaterial * max(1, len(meshes)))()
        self._mesh_keepalive = list(meshes)
        for i, (m, (e, col, refl)) in enumerate(zip(meshes, materials)):
            ms[i].positions, ms[i].normals, ms[i].indices = m.positions.ctypes.data, m.normals.ctypes.data, m.indices.ctypes.data
            ms[i].nverts, ms[i].ntris = len(m.positions), len(m.indices)
            mats[i].emission = (C.c_float * 3)(*[float(v) for v in e])
            mats[i].color = (C.c_float * 3)(*[float(v) for v in col])
            mats[i].refl = int(refl)
        self._check(self._lib.spt_multi_set_meshes(self._h, ms, len(meshes), mats))

    def set_mesh_accel(self, accel):
        self._check(self._lib.spt_multi_set_mesh_accel(self._h, int(accel)))

    def set_sphere_accel(self, accel):
        self._check(self._lib.spt_multi_set_sphere_accel(self._h, int(accel)))

    def set_rank_watchdog(self, rank, seconds):
        """Test hook (csrc/spt_internal.h): kernel watchdog of one rank's context."""
        self._check(self._lib.spt_multi_set_rank_watchdog(self._h, int(rank), float(seconds)))

    def inject_exchange_failure(self, rank):
        """Test hook (csrc/spt_internal.h): `rank` fails inside its part of the next RCCL exchange."""
        self._check(self._lib.spt_multi_inject_exchange_failure(self._h, int(rank)))

    def render(self, w, h, samps_per_cell, seed=0, normalise=False, camera=None, to_host=True):
        """Returns ((h, w, 3) float32 image or None, stats dict); with to_host=False the framebuffer stays on the root
        device (``framebuffer_ptr()``)."""
        cam = camera if camera is not None else smallpt_camera(w, h)
        out = np.empty((h, w, 3), dtype=np.float32) if to_host else None
        st = SptMultiStats()
        self._check(self._lib.spt_multi_render(self._h, C.byref(cam), w, h, samps_per_cell, seed,
                                               FLAG_NORMALISE if normalise else 0,
                                               out.ctypes.data_as(C.c_void_p) if to_host else None, C.byref(st)))
        return out, {"samples": int(st.samples), "bounces": int(st.bounces), "max_depth_kills": int(st.max_depth_kills),
                     "render_ms": float(st.render_ms), "gather_ms": float(st.gather_ms), "total_ms": float(st.total_ms),
                     "ndev": int(st.ndev)}

    def framebuffer_ptr(self):
        return self._lib.spt_multi_framebuffer(self._h)

    # the viewer's render loop over all devices (spt_multi_progressive_*): accumBuffer on the root device
    def progressive_begin(self, w, h):
        self._check(self._lib.spt_multi_progressive_begin(self._h, w, h))
        self._prog = (w, h)

    def progressive_frame(self, samps_per_cell, seed, clear=False, camera=None):
        """outImage = render(camera, ..., seed) on all devices; accumBuffer = outImage (clear) or += outImage on the root."""
        w, h = self._prog
        cam = camera if camera is not None else pinhole_camera()
        st = SptMultiStats()
        self._check(self._lib.spt_multi_progressive_frame(self._h, C.byref(cam), samps_per_cell, seed, 1 if clear else 0, C.byref(st)))
        return {"samples": int(st.samples), "bounces": int(st.bounces), "render_ms": float(st.render_ms), "gather_ms": float(st.gather_ms), "ndev": int(st.ndev)}

    def progressive_snapshot(self):
        w, h = self._prog
        out = np.empty((h, w, 3), dtype=np.float32)
        self._check(self._lib.spt_multi_progressive_snapshot(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def progressive_end(self):
        self._check(self._lib.spt_multi_progressive_end(self._h))


class ProgressiveRenderer:
    """The viewer's render-thread loop (smallpt.cpp:895-942) with the accumulation buffer resident in HBM:
    every ``step()`` renders one frame with seed = frame counter (:893,922,926) as an un-normalised sum
    (Renderer::render convention), adds it to the accumulation tensor (:935) and returns the display weight
    1/(frames*spp) of :957.  ``update_camera`` mirrors the "update_camera" request (:911-916): new camera,
    the next frame is still rendered with the RUNNING frame counter as its seed (:922), replaces the buffer
    (:931-935), and only then is the counter reset to 1 (:938-939).

    ``pipeline=2`` keeps two frames in flight on two contexts/streams of the same device: frame k+1 starts while the
    last, longest paths of frame k are still finishing (the end of a 4-spp frame is a handful of mirror<->glass chains,
    DESIGN.md section 5), the accumulation kernels stay in frame order (stream events), so ``accum`` is bit-identical
    to the serial loop; ``step()`` then returns without waiting and ``flush()`` waits for everything in flight."""

    def __init__(self, renderer, w, h, samps_per_cell, camera=None, pipeline=1):
        import torch
        self.r, self.w, self.h, self.samps = renderer, w, h, samps_per_cell
        self.camera = camera if camera is not None else pinhole_camera()
        dev = torch.device("cuda", renderer.device_id)
        self.accum = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        self.frames = 0          # sampleCount, smallpt.cpp:893
        self._clear = True       # the zero-initialised accumBuffer (:882): replacing it == adding to zeros
        self.pipeline = max(1, int(pipeline))
        self._lanes = []
        for i in range(self.pipeline):
            rr = renderer if i == 0 else Renderer(renderer.device_id)
            if i:
                renderer.replay_state_on(rr)              # spheres OR meshes + materials, closest-hit modes, tuning, watchdog
            self._lanes.append({"r": rr, "version": renderer._state_version, "frame": torch.empty((h, w, 3), dtype=torch.float32, device=dev),
                                # alternate stream priorities: HIP maps equal-priority streams of a process onto a shared hardware queue,
                                # which would serialise the two frames
                                "stream": torch.cuda.current_stream(dev) if self.pipeline == 1 else torch.cuda.Stream(dev, priority=-(i % 2)),
                                "done": None})
        self.frame = self._lanes[0]["frame"]
        self._issued = 0
        self._last_acc = None    # event after the most recent accumulation kernel

    def update_camera(self, camera):
        self.camera = camera
        self._clear = True

    def step(self):
        import torch
        lane = self._lanes[self._issued % self.pipeline]
        self._issued += 1
        r, stream = lane["r"], lane["stream"]
        if r is not self.r and lane["version"] != self.r._state_version:      # the primary's scene / modes changed since this lane was set up
            lane["stream"].synchronize()
            r.sync()
            self.r.replay_state_on(r)
            lane["version"] = self.r._state_version
        seed = self.frames       # :922 renders with the running sampleCount, also on the clearing frame
        if self.pipeline > 1 and lane["done"] is not None:
            stream.wait_event(lane["done"])               # the lane's frame buffer was read by its last accumulation
        r.render_rows_device(lane["frame"], self.w, self.h, 0, self.h, self.samps, seed=seed, normalise=False,
                             camera=self.camera, stream=stream.cuda_stream)
        if self.pipeline > 1 and self._last_acc is not None:
            stream.wait_event(self._last_acc)             # accumulations stay in frame order
        r._check(r._lib.spt_accumulate_device(r._h, C.c_void_p(self.accum.data_ptr()), C.c_void_p(lane["frame"].data_ptr()),
                                               self.accum.numel(), 1 if self._clear else 0, C.c_void_p(stream.cuda_stream)))
        if self.pipeline > 1:
            ev = torch.cuda.Event()
            ev.record(stream)
            lane["done"] = self._last_acc = ev
        self.frames = 1 if self._clear else self.frames + 1
        self._clear = False
        if self.pipeline == 1:
            r.sync()
        return 1.0 / (self.frames * 4 * self.samps)

    def flush(self):
        """Waits for every frame in flight (needed before reading ``accum`` when pipeline > 1)."""
        for lane in self._lanes:
            lane["stream"].synchronize()
            lane["r"].sync()

    def close(self):
        self.flush()
        for lane in self._lanes[1:]:
            lane["r"].close()


def to_int(x):
    """toInt, smallpt.cpp:52."""
    return load_library().spt_to_int(float(x))


def write_ppm(path, rgb):
    """flipY + writeImage (smallpt.cpp:125-142) for an (h, w, 3) float32 image, row 0 = bottom."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    if load_library().spt_write_ppm(str(path).encode(), rgb.ctypes.data_as(C.c_void_p), w, h):
        raise SptError(f"cannot write {path}")
