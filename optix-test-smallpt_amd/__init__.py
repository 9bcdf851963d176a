"""optix-test-smallpt_amd: MI355X-native replacement for the radiance()/intersect() hot path of
Celeborn2BeAlive/optix-test-smallpt.  The compute path is the gfx950 megakernel behind the C-ABI of
include/smallpt_mi355x.h (csrc/); this package is the thin host-side mirror of the reference's
scene structs and render entry points.  Import name: ``optix_test_smallpt_amd`` (see the shim at
the repository root)."""
from ._lib import (INTERNAL_SYMBOLS, LIB_PATH, MULTI_SYMBOLS, SYMBOLS, SptCamera, SptMaterial, SptMesh, SptMultiStats, SptSphere, SptStats,  # noqa: F401
                   load_library, load_multi_library)
from .renderer import (ACCEL_AUTO, ACCEL_BVH, ACCEL_BVH_FAST, ACCEL_EXHAUSTIVE, ACCEL_GRID, FLAG_NORMALISE, MultiRenderer, ProgressiveRenderer, Renderer, SptError, pinhole_camera,  # noqa: F401
                       smallpt_camera, to_int, write_ppm)
from .scene import (DIFF, HIT_DTYPE, RAY_DTYPE, REFR, SPEC, SPHERE_DTYPE, TriMesh, cornell9, make_sphere_trimesh,  # noqa: F401
                    make_spheres, meshes_from_json, meshes_to_json, random_spheres, single_triangle_scene, spheres_from_json,
                    spheres_to_json)

__all__ = ["Renderer", "MultiRenderer", "ProgressiveRenderer", "SptError", "smallpt_camera", "pinhole_camera", "cornell9", "random_spheres", "make_spheres",
           "spheres_from_json", "spheres_to_json", "SPHERE_DTYPE", "DIFF", "SPEC", "REFR",
           "load_library", "to_int", "write_ppm", "FLAG_NORMALISE", "ACCEL_EXHAUSTIVE", "ACCEL_BVH", "ACCEL_BVH_FAST", "ACCEL_AUTO", "ACCEL_GRID"]
