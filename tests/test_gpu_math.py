"""GPU numerics tests of the exact-math device helpers against correctly rounded CPU results (numpy float32
sqrt / division are IEEE correctly rounded; the oracle supplies sincos and the RNG)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def _inputs():
    rs = np.random.RandomState(11)
    bits = rs.randint(0, 0x7F800000, size=1 << 22, dtype=np.uint32)                    # all positive finite floats
    rnd = bits.view(np.float32)
    edge = np.array([0.0, 1.0, 2.0, 4.0, 0.25, 3.0, 1e-4, 1e20, 1e-30, 1e-38, 2.0**-96, 2.0**-97, 2.0**-126,
                     1e-45, 1.4e-45, 3.4028235e38, 0.99999994, 1.0000001, 16777216.0, 1.9999999, 3.9999998],
                    dtype=np.float32)
    squares = (np.arange(1, 5000, dtype=np.float32) ** 2)
    near_sq = np.concatenate([np.nextafter(squares, f32(0)), np.nextafter(squares, f32(np.inf))])
    uni = (rs.randint(0, 1 << 24, size=1 << 20).astype(np.float32) * f32(2.0**-24))   # RNG-shaped values
    return np.concatenate([edge, squares, near_sq, uni, f32(1) - uni, f32(2) * uni, rnd])


def test_sqrt_helpers_are_correctly_rounded(renderer):
    x = _inputs()
    ref = np.sqrt(x)
    big = x >= f32(2.0**-96)
    for op, name in ((0, "sqrt_fix"), (10, "sqrt_rsq")):
        y = renderer.selftest_math(op, x)
        ok = (y == ref) | ~(big | (x == 0))          # un-guarded forms are specified for x = 0 or x >= 2^-96
        assert ok.all(), (name, x[~ok][:5], y[~ok][:5], ref[~ok][:5])
    y = renderer.selftest_math(2, x)                  # guarded form: every non-negative finite input
    assert (y == ref).all(), (x[y != ref][:5], y[y != ref][:5])
    # special values: +0 stays +0, inf stays inf, negatives and NaN give NaN
    sp = np.array([0.0, np.inf, -1.0, np.nan], dtype=np.float32)
    for op in (0, 2, 10):
        y = renderer.selftest_math(op, sp)
        assert y[0] == 0 and not np.signbit(y[0]) and np.isnan(y[2]) and np.isnan(y[3])
        if op != 10:            # the rsq form is specified for finite arguments only (the API selects the
            assert y[1] == np.inf   # guarded kernel build for scenes whose coordinates could overflow)


def test_sqrt_rsq_exhaustive(renderer):
    """The kernels' square root (v_rsq_f32 + one exact-residual FMA correction, csrc/spt_device.h sqrt_rsq) depends on the
    hardware's reciprocal-square-root table, so its correct rounding is established by enumeration ON THE DEVICE: every
    binary32 value in its specified range 2^-96 <= x < inf (1 879 048 192 inputs) must equal the CPU-proven sqrt_fix bit
    for bit.  The uncorrected estimate run through the same comparison is the negative control."""
    lo, hi = (127 - 96) << 23, 0x7F800000
    m, first_bad = C.c_uint64(), C.c_uint32()
    lib, h = renderer._lib, renderer._h
    assert lib.spt_selftest_range(h, 0, lo, hi - lo, C.byref(m), C.byref(first_bad)) == 0
    assert m.value == 0 and first_bad.value == 0xFFFFFFFF, (m.value, hex(first_bad.value))
    assert lib.spt_selftest_range(h, 1, lo, hi - lo, C.byref(m), C.byref(first_bad)) == 0
    assert m.value > 1_000_000 and lo <= first_bad.value < hi, (m.value, hex(first_bad.value))
    x = np.array([0.0, -0.0], dtype=np.float32)
    y = renderer.selftest_math(10, x)
    assert y[0] == 0 and not np.signbit(y[0]) and y[1] == 0 and np.signbit(y[1])


def test_rcp_exact_exhaustive(renderer):
    """rcp_exact<false> (v_rcp_f32 + one FMA Newton step) against the compiler's IEEE division for every binary32 value in
    2^-100 <= y < 2^100: as for sqrt_rsq the property belongs to the hardware table and is enumerated on the device."""
    lo, hi = (127 - 100) << 23, (127 + 100) << 23
    m, first_bad = C.c_uint64(), C.c_uint32()
    lib, h = renderer._lib, renderer._h
    assert lib.spt_selftest_range(h, 2, lo, hi - lo, C.byref(m), C.byref(first_bad)) == 0
    assert m.value == 0 and first_bad.value == 0xFFFFFFFF, (m.value, hex(first_bad.value))
    assert lib.spt_selftest_range(h, 3, lo, hi - lo, C.byref(m), C.byref(first_bad)) == 0
    assert m.value > 1_000_000 and lo <= first_bad.value < hi, (m.value, hex(first_bad.value))


def test_rcp_exact_is_correctly_rounded(renderer):
    x = _inputs()
    x = x[(x > 0) & np.isfinite(x)]
    allones = ((np.arange(1, 255, dtype=np.uint32) << 23) | 0x7FFFFF).view(np.float32)   # mantissa 0x7FFFFF, every exponent
    x = np.concatenate([x, allones, np.array([2.0**-100, 2.0**100, 2.0**-101, 2.0**101, 2.0**-126, 2.0**126], dtype=np.float32)])
    with np.errstate(over="ignore", divide="ignore"):
        ref = f32(1) / x
    y = renderer.selftest_math(3, x)
    assert (y == ref).all(), (x[y != ref][:5], y[y != ref][:5], ref[y != ref][:5])


def test_double_division_sequence(renderer):
    rs = np.random.RandomState(5)
    for w in (1024, 768, 4096, 1000, 333, 3, 1):
        a = (rs.randint(0, w * 2, size=1 << 18) + rs.randint(0, 1 << 24, size=1 << 18) * 2.0**-24).astype(np.float32)
        ref = (a.astype(np.float64) / np.float64(w)).astype(np.float32)
        assert np.array_equal(renderer.selftest_math(4, a, w=w), ref)


def test_sincos_and_rng_match_oracle(renderer, oracle):
    L = oracle.lib()
    u = (np.random.RandomState(2).randint(0, 1 << 24, size=20000).astype(np.float32) * f32(2.0**-24))
    u = np.concatenate([u, np.array([0.0, 0.25, 0.5, 0.75, 0.24999999, 0.99999994], dtype=np.float32)])
    sn, cs = renderer.selftest_math(5, u), renderer.selftest_math(6, u)
    s, c = C.c_float(), C.c_float()
    for i, v in enumerate(u):
        L.orc_sincos2pi(float(v), C.byref(s), C.byref(c))
        assert (f32(s.value), f32(c.value)) == (sn[i], cs[i])
    # the kernel's bit-level variant (q, f taken from the raw draw) must equal the float route for every draw value
    bits = np.concatenate([np.random.RandomState(9).randint(0, 1 << 32, size=1 << 20, dtype=np.uint64).astype(np.uint32),
                           np.array([0, 0xFF, 0x100, 0x3FFFFFFF, 0x40000000, 0x7FFFFFFF, 0x80000000, 0xBFFFFFFF, 0xC0000000,
                                     0xFFFFFFFF, 0xFFFFFF00], dtype=np.uint32)])
    uu = ((bits >> np.uint32(8)).astype(np.float32) * f32(2.0**-24))
    assert np.array_equal(renderer.selftest_math(8, bits.view(np.float32)).view(np.uint32), renderer.selftest_math(5, uu).view(np.uint32))
    assert np.array_equal(renderer.selftest_math(9, bits.view(np.float32)).view(np.uint32), renderer.selftest_math(6, uu).view(np.uint32))
    keys = np.random.RandomState(3).randint(0, 1 << 31, size=5000, dtype=np.uint32)
    y = renderer.selftest_math(7, keys.view(np.float32))
    for i, k in enumerate(keys[:2000]):
        # rng_draw(x, k1) = mix of the pre-added counter word: equals orc_rng_uniform(k0 = x, k1, ctr = 0)
        assert L.orc_rng_uniform(int(k), 0x9ABCDEF0, 0) == float(y[i])
