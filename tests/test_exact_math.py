"""CPU proof-by-enumeration that the hardware-independent GPU sequences (sqrt fix-up from any 1-ulp estimate, two-step
Newton reciprocal model, Markstein double division) are correctly rounded; the kernels' faster sqrt_rsq / one-step rcp_exact
are enumerated on the device instead (tests/test_gpu_math.py).  This builds and runs tools/verify_exact_math.c (quick mode: every 7th mantissa,
2e6 divisions; the full run is `tools/verify_exact_math` without arguments)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exact_math_sequences(tmp_path):
    exe = str(tmp_path / "verify_exact_math")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tools", "verify_exact_math.c"),
                           "-lm", "-o", exe])
    out = subprocess.run([exe, "quick"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "sqrt mismatches 0, rcp exceptions 0, div mismatches 0" in out.stdout


def test_reciprocal_model_exception_is_the_all_ones_mantissa(tmp_path):
    """Under the hardware-independent model (ANY starting value within 1 ulp, two Newton steps) the full enumeration finds
    exactly one failing mantissa, 0x7FFFFF.  The kernels' rcp_exact() does not rest on that model: it uses ONE step and the
    actual v_rcp_f32 table, enumerated on the device for every input (tests/test_gpu_math.py::test_rcp_exact_exhaustive);
    this test keeps the model's result on record and checks that the header points at the device enumeration."""
    exe = str(tmp_path / "verify_exact_math_full")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tools", "verify_exact_math.c"),
                           "-lm", "-o", exe])
    src = open(os.path.join(ROOT, "optix-test-smallpt_amd", "csrc", "spt_device.h")).read()
    assert "tests/test_gpu_math.py compares" in src and "all 1 677 721 600 inputs" in src
    out = subprocess.run([exe, "rcp-only"], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("rcp exception")]
    assert lines and all("mant 0x7fffff" in l for l in lines)
