"""Compiler-reported resources of EVERY product kernel (cross-compiled for gfx950 here, no GPU needed): path state stays in
registers -- no scalar or vector spills, no scratch -- and the occupancy the launch geometry counts on is available.  Reads the
-Rpass-analysis=kernel-resource-usage remarks of the five device translation units.  Instrumented builds (megakernel DIAG,
gridkernel / gpoolkernel STATS: selected by tuning bit 8 only, never timed as the product) are exempt from the spill rule but must not use scratch."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optix-test-smallpt_amd", "csrc")


def _resources(unit, tmp_path):
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-kernel-flags"], capture_output=True, text=True, check=True).stdout.split()
    out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--offload-arch=gfx950", "-S", "--cuda-device-only",
                          "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, unit), "-o", str(tmp_path / (unit + ".s"))],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            kernels[name][m.group(1).strip()] = int(m.group(2))
    return kernels


def test_pool_kernels_do_not_spill(tmp_path):
    kernels = _resources("spt_pool.hip", tmp_path)
    pool = {k: v for k, v in kernels.items() if "poolkernel" in k}
    assert len(pool) >= 24, sorted(kernels)
    for k, r in pool.items():
        assert r["SGPRs Spill"] == 0 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (k, r)
        assert r["VGPRs"] + r["AGPRs"] <= 128 and r["Occupancy"] >= 4, (k, r)


@pytest.mark.parametrize("unit, expect, min_occupancy", [
    ("spt_kernel.hip", ("megakernel", "finalize", "accumulate"), 4),     # 4 waves/SIMD: four 256-thread workgroups per CU
    ("spt_mesh.hip", ("meshkernel", "trace_rays"), 4),
    ("spt_grid.hip", ("gridkernel",), 4),                                # one 1024-thread workgroup per CU = 4 waves/SIMD
    ("spt_gpool.hip", ("gpoolkernel",), 4),                              # the same geometry
])
def test_product_kernels_do_not_spill(unit, expect, min_occupancy, tmp_path):
    kernels = _resources(unit, tmp_path)
    for stem in expect:
        assert any(stem in k for k in kernels), (stem, sorted(kernels))
    for k, r in kernels.items():
        # Itanium mangling of the template arguments: megakernel<MAT_LDS, GUARD, DIAG, BIGN, BLOCK> -> ...ILb?ELb?ELb1E...; gridkernel<STATS> -> ILb1E
        instrumented = ("megakernel" in k and re.search(r"megakernelILb[01]ELb[01]ELb1E", k)) or ("gridkernel" in k and "gridkernelILb1E" in k) or "gpoolkernelILb1E" in k
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, (k, r)
        if not instrumented:
            assert r["SGPRs Spill"] == 0, (k, r)
            assert r["VGPRs"] + r["AGPRs"] <= 128 and r["Occupancy"] >= min_occupancy, (k, r)
