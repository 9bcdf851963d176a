"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the header declares,
its host-only helpers agree with the oracle, and it refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smallpt_mi355x.h")


INTERNAL_HEADER = os.path.join(ROOT, "optix-test-smallpt_amd", "csrc", "spt_internal.h")


def _declared_symbols(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/smallpt_mi355x.h but not exported"
    assert sorted(pkg.SYMBOLS) == declared          # the Python binding covers the whole header
    # test/tuning hooks live in an internal header, not in the drop-in boundary
    internal = _declared_symbols(INTERNAL_HEADER)
    multi_hooks = [n for n in internal if n.startswith("spt_multi_")]       # hooks of libsmallpt_mi355x_multi.so
    internal = [n for n in internal if n not in multi_hooks]
    from optix_test_smallpt_amd._lib import MULTI_INTERNAL_SYMBOLS
    assert sorted(MULTI_INTERNAL_SYMBOLS) == multi_hooks
    assert sorted(pkg.INTERNAL_SYMBOLS) == internal and not set(internal) & set(declared)
    for name in internal:
        assert hasattr(lib, name)
    assert lib.spt_api_version() == 1


def test_struct_layouts(pkg):
    assert C.sizeof(pkg.SptSphere) == 48 == pkg.SPHERE_DTYPE.itemsize
    assert C.sizeof(pkg.SptCamera) == 56
    assert pkg.SptSphere.radius.offset == 12 and pkg.SptSphere.refl.offset == 40
    hdr = open(HEADER).read()
    assert "SPT_MAX_DEPTH      4096u" in hdr


def test_camera_helper_matches_oracle(pkg, oracle):
    for w, h in ((256, 256), (1024, 768), (4096, 4096), (33, 17)):
        a = pkg.smallpt_camera(w, h)
        b = oracle.camera_smallpt(w, h)
        assert bytes(a) == bytes(b)


def test_pinhole_camera_helper_matches_oracle(pkg, oracle):
    """Camera{vx, vy, vz, org, near} of the interactive driver (smallpt.cpp:607-624,885-899)."""
    a = pkg.pinhole_camera()
    b = oracle.OrcCamera()
    vy = [float(v) for v in a.cy]
    oracle.lib().orc_camera_pinhole(oracle.f3(1, 0, 0), oracle.f3(*vy), oracle.f3(0, 0, -1), oracle.f3(0, -1, 0), C.c_float(1.0), C.byref(b))
    assert bytes(a) == bytes(b) and a.sampler == 1 and a.push == 0.0
    assert list(a.cy) == [0.0, 1.0, 0.0] and list(a.dir) == [0.0, 0.0, -1.0]     # vy = normalize(cross(vx, vz))


def test_to_int_matches_oracle(pkg, oracle):
    xs = np.concatenate([np.linspace(-0.5, 1.5, 2001), [0.0, 1.0, 0.5]]).astype(np.float32)
    assert [pkg.to_int(x) for x in xs] == [oracle.lib().orc_to_int(float(x)) for x in xs]


def test_write_ppm_flips_rows(pkg, tmp_path):
    img = np.zeros((2, 3, 3), dtype=np.float32)
    img[0, :, 0] = 1.0                 # bottom row red
    p = tmp_path / "t.ppm"
    pkg.write_ppm(p, img)
    tok = open(p).read().split()
    assert tok[:4] == ["P3", "3", "2", "255"]
    vals = list(map(int, tok[4:]))
    assert vals[:9] == [0] * 9 and vals[9:] == [255, 0, 0] * 3      # file row 0 = top = black


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback(pkg):
    with pytest.raises(pkg.SptError, match="no CPU fallback"):
        pkg.Renderer(0)


def test_product_never_references_oracle():
    """The product path must not import, link or call anything under oracle/."""
    bad = []
    pkgdir = os.path.join(ROOT, "optix-test-smallpt_amd")
    for d, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile", ".txt")):
                t = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"liboracle|smallpt_oracle|oracle_binding|orc_render", t):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
