"""Compiler-reported resources of the pool kernels (cross-compiled for gfx950 here, no GPU needed): every poolkernel<P, NG>
instantiation must keep its state in registers -- no scalar or vector spills, no scratch -- and leave room for four
waves per SIMD (16 per CU).  Reads `make -C csrc asm`'s -Rpass-analysis=kernel-resource-usage remarks."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optix-test-smallpt_amd", "csrc")


def test_pool_kernels_do_not_spill(tmp_path):
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-kernel-flags"], capture_output=True, text=True, check=True).stdout.split()
    out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--offload-arch=gfx950", "-S", "--cuda-device-only",
                          "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, "spt_pool.hip"), "-o", str(tmp_path / "pool.s")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels = {}
    name = None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            kernels[name][m.group(1).strip()] = int(m.group(2))
    pool = {k: v for k, v in kernels.items() if "poolkernel" in k}
    assert len(pool) >= 24, sorted(kernels)
    for k, r in pool.items():
        assert r["SGPRs Spill"] == 0 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, (k, r)
        assert r["VGPRs"] + r["AGPRs"] <= 128 and r["Occupancy"] >= 4, (k, r)
