"""Exhaustive device checks of sqrt_rsq and rcp_exact (csrc/spt_device.h): prints the mismatch counts for the specified
ranges, for the negative controls and for the unspecified sqrt range below 2^-96."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import optix_test_smallpt_amd as pkg
r = pkg.Renderer(0)
lib = r._lib
m, fb = C.c_uint64(), C.c_uint32()
lo, hi = (127 - 96) << 23, 0x7F800000
rlo, rhi = (127 - 100) << 23, (127 + 100) << 23
for name, first, count, op in (("sqrt_rsq [2^-96, inf)", lo, hi - lo, 0), ("sqrt_rsq [2^-96, inf) control", lo, hi - lo, 1), ("sqrt_rsq [0, 2^-96)", 0, lo, 0),
                               ("rcp_exact [2^-100, 2^100)", rlo, rhi - rlo, 2), ("rcp_exact control", rlo, rhi - rlo, 3)):
    assert lib.spt_selftest_range(r._h, op, first, count, C.byref(m), C.byref(fb)) == 0
    print(f"range {name}: {count} inputs, mismatches {m.value}, first bad bits {fb.value:#x}", flush=True)
sp = np.array([0.0, -0.0, -1.0, np.nan, np.inf, 1e-30, 2.0**-96, 2.0**-97], dtype=np.float32)
print("special:", sp, r.selftest_math(10, sp))
