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


@pytest.mark.gpu
def test_random_mesh_scenes_hierarchy_equals_exhaustive():
    """Short run of tools/fuzz_mesh_hierarchy.py: random mesh scenes (tessellated spheres, soups, coplanar soups with slivers / collinear /
    zero-edge triangles, scenes 3e4 from the origin) x random and adversarial rays (in a triangle's plane anywhere in it, across the
    supporting lines of needles, ...): SPT_ACCEL_BVH returns the exhaustive loop's Hit, byte for byte, on every ray."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_mesh_hierarchy.py")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, FUZZ_SECONDS="12", FUZZ_SEED="5"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " 0 differ -> ok" in out.stdout
