"""CPU proof-by-enumeration that the short GPU sequences (sqrt fix-up, Newton reciprocal, Markstein double
division) are correctly rounded: builds and runs tools/verify_exact_math.c (quick mode: every 7th mantissa,
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


def test_reciprocal_exception_is_the_all_ones_mantissa(tmp_path):
    """The full enumeration finds exactly one failing mantissa (0x7FFFFF), which rcp_exact() routes to the
    compiler's IEEE division."""
    exe = str(tmp_path / "verify_exact_math_full")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tools", "verify_exact_math.c"),
                           "-lm", "-o", exe])
    src = open(os.path.join(ROOT, "optix-test-smallpt_amd", "csrc", "spt_device.h")).read()
    assert "(u & 0x7FFFFFu) == 0x7FFFFFu" in src
    out = subprocess.run([exe, "rcp-only"], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("rcp exception")]
    assert lines and all("mant 0x7fffff" in l for l in lines)
