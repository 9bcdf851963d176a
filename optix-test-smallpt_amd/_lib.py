"""ctypes binding of the C-ABI in include/smallpt_mi355x.h (libsmallpt_mi355x.so).

There is deliberately no fallback: if the HIP library is missing, or no gfx950 device is present
when a context is created, this raises.  Nothing here imports or calls the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPT_LIB overrides the library path (A/B of compiler-flag variants of the same source; tools/build_variants.sh)
LIB_PATH = os.environ.get("SPT_LIB") or os.path.join(_HERE, "csrc", "libsmallpt_mi355x.so")


class SptSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("emission", C.c_float * 3),
                ("color", C.c_float * 3), ("refl", C.c_int32), ("pad", C.c_uint32)]


class SptCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("cx", C.c_float * 3),
                ("cy", C.c_float * 3), ("push", C.c_float), ("sampler", C.c_uint32)]


class SptStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("bounces", C.c_uint64), ("max_depth_kills", C.c_uint64),
                ("kernel_ms", C.c_float), ("finalize_ms", C.c_float), ("total_ms", C.c_float),
                ("grid_blocks", C.c_uint32), ("block_threads", C.c_uint32), ("pad", C.c_uint32)]


class SptMesh(C.Structure):          # TriMesh, scene.h:6-15
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("indices", C.c_void_p),
                ("nverts", C.c_uint32), ("ntris", C.c_uint32)]


class SptMaterial(C.Structure):      # Material, scene.h:66-73
    _fields_ = [("emission", C.c_float * 3), ("color", C.c_float * 3), ("refl", C.c_int32), ("pad", C.c_uint32)]


# every symbol include/smallpt_mi355x.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "spt_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "spt_destroy": (None, [_P]),
    "spt_last_error": (C.c_char_p, [_P]),
    "spt_api_version": (C.c_int, []),
    "spt_device_count": (C.c_int, []),
    "spt_set_scene": (C.c_int, [_P, _P, C.c_uint32]),
    "spt_set_meshes": (C.c_int, [_P, C.POINTER(SptMesh), C.c_uint32, C.POINTER(SptMaterial)]),
    "spt_set_mesh_accel": (C.c_int, [_P, C.c_int]),
    "spt_set_sphere_accel": (C.c_int, [_P, C.c_int]),
    "spt_trace_rays": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "spt_trace_rays_device": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "spt_make_sphere_trimesh": (C.c_uint32, [C.c_float * 3, C.c_float, C.c_uint32, _P, _P, _P]),
    "spt_camera_smallpt": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(SptCamera)]),
    "spt_camera_pinhole": (C.c_int, [C.c_float * 3, C.c_float * 3, C.c_float * 3, C.c_float * 3, C.c_float, C.POINTER(SptCamera)]),
    "spt_render": (C.c_int, [_P, C.POINTER(SptCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                             C.c_uint32, _P, C.POINTER(SptStats)]),
    "spt_render_rows_device": (C.c_int, [_P, C.POINTER(SptCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _P, _P]),
    "spt_interleaved_row_count": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "spt_render_interleaved_device": (C.c_int, [_P, C.POINTER(SptCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint64, C.c_uint32, _P, _P]),
    "spt_accumulate_device": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_int, _P]),
    "spt_progressive_begin": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "spt_progressive_frame": (C.c_int, [_P, C.POINTER(SptCamera), C.c_uint32, C.c_uint64, C.c_int, C.POINTER(SptStats)]),
    "spt_progressive_snapshot": (C.c_int, [_P, _P]),
    "spt_progressive_attach": (C.c_int, [_P, _P]),
    "spt_progressive_frame_async": (C.c_int, [_P, _P, C.POINTER(SptCamera), C.c_uint32, C.c_uint64, C.c_int]),
    "spt_progressive_wait": (C.c_int, [_P, C.POINTER(SptStats)]),
    "spt_progressive_end": (C.c_int, [_P]),
    "spt_sync": (C.c_int, [_P, C.POINTER(SptStats)]),
    "spt_to_int": (C.c_int, [C.c_float]),
    "spt_write_ppm": (C.c_int, [C.c_char_p, _P, C.c_uint32, C.c_uint32]),
}

# test / tuning hooks declared in csrc/spt_internal.h (not part of the drop-in boundary)
INTERNAL_SYMBOLS = {
    "spt_selftest_sphere_bvh": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint32 * 4), C.c_char_p, C.c_uint32]),
    "spt_selftest_sphere_grid": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32 * 8), C.c_char_p, C.c_uint32]),
    "spt_selftest_bvh": (C.c_int, [C.POINTER(SptMesh), C.c_uint32, C.POINTER(C.c_uint32 * 4), C.c_char_p, C.c_uint32]),
    "spt_set_tuning": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "spt_set_grid_pools": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "spt_diag": (C.c_int, [_P, C.POINTER(C.c_uint64 * 24)]),
    "spt_selftest_math": (C.c_int, [_P, C.c_int, _P, _P, C.c_uint32, C.c_uint32]),
    "spt_selftest_range": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "spt_set_watchdog": (C.c_int, [_P, C.c_double]),
    "spt_last_kernel": (C.c_int, [_P]),
    "spt_chunk_order_snapshot": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
}

class SptMultiStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("bounces", C.c_uint64), ("max_depth_kills", C.c_uint64),
                ("render_ms", C.c_float), ("gather_ms", C.c_float), ("total_ms", C.c_float),
                ("ndev", C.c_uint32), ("pad", C.c_uint32)]


# every symbol include/smallpt_mi355x_multi.h declares (libsmallpt_mi355x_multi.so, links RCCL)
MULTI_LIB_PATH = os.path.join(_HERE, "csrc", "libsmallpt_mi355x_multi.so")
MULTI_SYMBOLS = {
    "spt_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_uint32, C.POINTER(_P)]),
    "spt_multi_destroy": (None, [_P]),
    "spt_multi_last_error": (C.c_char_p, [_P]),
    "spt_multi_device_count": (C.c_int, [_P]),
    "spt_multi_set_scene": (C.c_int, [_P, _P, C.c_uint32]),
    "spt_multi_set_meshes": (C.c_int, [_P, C.POINTER(SptMesh), C.c_uint32, C.POINTER(SptMaterial)]),
    "spt_multi_set_mesh_accel": (C.c_int, [_P, C.c_int]),
    "spt_multi_set_sphere_accel": (C.c_int, [_P, C.c_int]),
    "spt_multi_row_band": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "spt_multi_render": (C.c_int, [_P, C.POINTER(SptCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.c_uint32, _P, C.POINTER(SptMultiStats)]),
    "spt_multi_framebuffer": (_P, [_P]),
    "spt_multi_progressive_begin": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "spt_multi_progressive_frame": (C.c_int, [_P, C.POINTER(SptCamera), C.c_uint32, C.c_uint64, C.c_int, C.POINTER(SptMultiStats)]),
    "spt_multi_progressive_snapshot": (C.c_int, [_P, _P]),
    "spt_multi_progressive_end": (C.c_int, [_P]),
}

MULTI_INTERNAL_SYMBOLS = {"spt_multi_set_rank_watchdog": (C.c_int, [_P, C.c_uint32, C.c_double]),   # csrc/spt_internal.h
                          "spt_multi_inject_exchange_failure": (C.c_int, [_P, C.c_uint32])}

_lib = None
_multi_lib = None


def load_multi_library():
    """Loads libsmallpt_mi355x_multi.so (multi-GPU front, RCCL) and binds every symbol; raises if it is missing."""
    global _multi_lib
    if _multi_lib is not None:
        return _multi_lib
    load_library()
    if not os.path.exists(MULTI_LIB_PATH):
        raise RuntimeError(f"{MULTI_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(MULTI_LIB_PATH)
    for name, (res, args) in {**MULTI_SYMBOLS, **MULTI_INTERNAL_SYMBOLS}.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _multi_lib = lib
    return lib


def load_library():
    """Loads libsmallpt_mi355x.so and binds every symbol; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch wheels bundle their own ROCm runtime (libamdhip64 / libhsa-runtime64 under torch/lib).  If this library
    # pulled in the system runtime first, a later `import torch` in the same process would initialise a second HSA
    # runtime and report "No HIP GPUs are available"; loading torch's runtime first makes both share one.  Plumbing only
    # (device tensors / streams / torch.distributed); nothing of the render path runs through torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in {**SYMBOLS, **INTERNAL_SYMBOLS}.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
