"""Short randomised parity run (tools/fuzz_parity.py): random sphere tables incl. overlapping, huge and emissive
spheres, all materials, both cameras, ragged sizes.  Bit-identical images and bounce counts are required."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_scenes_match_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "8", "77"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " 0 mismatches" in out.stdout
