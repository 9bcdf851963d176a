"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("emission", "<f4", 3),
                         ("color", "<f4", 3), ("refl", "<i4"), ("pad", "<u4")])


class OrcCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("cx", C.c_float * 3),
                ("cy", C.c_float * 3), ("push", C.c_float), ("sampler", C.c_uint32)]


class OrcStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("bounces", C.c_uint64), ("max_depth_kills", C.c_uint64)]


class OrcMesh(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("indices", C.c_void_p), ("nverts", C.c_uint32), ("ntris", C.c_uint32)]


class OrcMaterial(C.Structure):
    _fields_ = [("emission", C.c_float * 3), ("color", C.c_float * 3), ("refl", C.c_int32), ("pad", C.c_uint32)]


HIT_DTYPE = np.dtype([("dist", "<f4"), ("instId", "<u4"), ("triId", "<u4"), ("x", "<f4", 3), ("n", "<f4", 3), ("uv", "<f4", 2)])

FLAG_NORMALISE = 1
FLAG_NO_ZERO_WEIGHT_CUT = 2
FLAG_SEQUENTIAL_CELLS = 4
FLAG_SEQUENTIAL_PIXEL = 8
_lib = None


def build_oracle():
    src = os.path.join(ORACLE_DIR, "smallpt_oracle.c")
    if (not os.path.exists(ORACLE_LIB)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(ORACLE_LIB)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_LIB)
        L.orc_intersect_analytic.restype = C.c_float
        L.orc_rng_uniform.restype = C.c_float
        L.orc_rng_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_bits.restype = C.c_uint32
        L.orc_rng_bits.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_mix32.restype = C.c_uint32
        L.orc_mix32.argtypes = [C.c_uint32]
        L.orc_sample_keys.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_sincos2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_to_int.argtypes = [C.c_float]
        L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p,
                                 C.POINTER(OrcStats)]
        L.orc_camera_ray.argtypes = [C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_make_sphere_trimesh.restype = C.c_uint32
        L.orc_make_sphere_trimesh.argtypes = [C.c_float * 3, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_rays.restype = None
        L.orc_trace_rays.argtypes = [C.POINTER(OrcMesh), C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_render_meshes.argtypes = [C.POINTER(OrcMesh), C.c_uint32, C.POINTER(OrcMaterial), C.POINTER(OrcCamera), C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(OrcStats)]
        _lib = L
    return _lib


def f3(*v):
    return (C.c_float * 3)(*v)


def camera_smallpt(w, h):
    cam = OrcCamera()
    lib().orc_camera_smallpt(C.c_uint32(w), C.c_uint32(h), C.byref(cam))
    return cam


def camera_from(other):
    """Copies any ctypes camera struct with the same 13-float layout (e.g. the product's SptCamera)."""
    cam = OrcCamera()
    C.memmove(C.byref(cam), C.byref(other), C.sizeof(OrcCamera))
    return cam


def render(spheres, w, h, samps, seed=0, normalise=False, row_begin=0, row_count=None, threads=0,
           camera=None, zero_cut=True, summation=None):
    """Oracle render of rows [row_begin, row_begin+row_count); returns ((rows, w, 3) float32, stats dict)."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    if row_count is None:
        row_count = h - row_begin
    cam = camera if camera is not None else camera_smallpt(w, h)
    if not isinstance(cam, OrcCamera):
        cam = camera_from(cam)
    out = np.zeros((row_count, w, 3), dtype=np.float32)
    st = OrcStats()
    flags = (FLAG_NORMALISE if normalise else 0) | (0 if zero_cut else FLAG_NO_ZERO_WEIGHT_CUT)
    flags |= {None: 0, "cells": FLAG_SEQUENTIAL_CELLS, "pixel": FLAG_SEQUENTIAL_PIXEL}[summation]   # D9 alternatives (bounding tests only)
    rc = lib().orc_render(spheres.ctypes.data_as(C.c_void_p), len(spheres), C.byref(cam), w, h, row_begin,
                          row_count, samps, seed, flags, threads, out.ctypes.data_as(C.c_void_p), C.byref(st))
    if rc:
        raise RuntimeError(f"orc_render failed rc={rc}")
    return out, {"samples": int(st.samples), "bounces": int(st.bounces), "max_depth_kills": int(st.max_depth_kills)}


# ---- triangle meshes (any object with .positions/.normals/.indices numpy arrays, e.g. the product's TriMesh container) ----
def make_sphere_trimesh(origin, radius, subdiv=32):
    n = int(subdiv)
    pos = np.zeros(((n + 1) * (2 * n + 1), 3), dtype=np.float32)
    nor = np.zeros_like(pos)
    idx = np.zeros((4 * n * n, 3), dtype=np.uint32)
    nt = lib().orc_make_sphere_trimesh(f3(*origin), float(radius), n, pos.ctypes.data, nor.ctypes.data, idx.ctypes.data)
    assert nt == len(idx)
    return pos, nor, idx


def _mesh_args(meshes, materials):
    ms = (OrcMesh * max(1, len(meshes)))()
    mats = (OrcMaterial * max(1, len(meshes)))()
    keep = []
    for i, (m, (e, col, refl)) in enumerate(zip(meshes, materials)):
        p = np.ascontiguousarray(m.positions, dtype=np.float32); nn = np.ascontiguousarray(m.normals, dtype=np.float32)
        ix = np.ascontiguousarray(m.indices, dtype=np.uint32)
        keep += [p, nn, ix]
        ms[i].positions, ms[i].normals, ms[i].indices = p.ctypes.data, nn.ctypes.data, ix.ctypes.data
        ms[i].nverts, ms[i].ntris = len(p.reshape(-1, 3)), len(ix.reshape(-1, 3))
        mats[i].emission = f3(*[float(v) for v in e]); mats[i].color = f3(*[float(v) for v in col]); mats[i].refl = int(refl)
    return ms, mats, keep


def trace_rays(meshes, rays):
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    ms, _, keep = _mesh_args(meshes, [((0, 0, 0), (0, 0, 0), 0)] * len(meshes))
    hits = np.zeros(len(rays), dtype=HIT_DTYPE)
    lib().orc_trace_rays(ms, len(meshes), rays.ctypes.data, len(rays), hits.ctypes.data)
    return hits


def render_meshes(meshes, materials, w, h, samps, seed=0, normalise=False, row_begin=0, row_count=None, threads=0, camera=None):
    if row_count is None:
        row_count = h - row_begin
    cam = camera if camera is not None else camera_smallpt(w, h)
    if not isinstance(cam, OrcCamera):
        cam = camera_from(cam)
    ms, mats, keep = _mesh_args(meshes, materials)
    out = np.zeros((row_count, w, 3), dtype=np.float32)
    st = OrcStats()
    rc = lib().orc_render_meshes(ms, len(meshes), mats, C.byref(cam), w, h, row_begin, row_count, samps, seed,
                                 FLAG_NORMALISE if normalise else 0, threads, out.ctypes.data, C.byref(st))
    if rc:
        raise RuntimeError(f"orc_render_meshes failed rc={rc}")
    return out, {"samples": int(st.samples), "bounces": int(st.bounces), "max_depth_kills": int(st.max_depth_kills)}
