"""Generates the golden image fixtures in this directory from the CPU oracle (oracle/liboracle.so).

The reference itself cannot produce images here (it needs OptiX/GLFW and its path tracer is switched
off), so these are ORACLE outputs: they pin the oracle against regressions and
give the GPU tests a fixture that does not need the oracle at run time.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import optix_test_smallpt_amd as pkg  # noqa: E402  (scene tables only; no GPU needed)
import oracle_binding as orc  # noqa: E402

CASES = {
    # name: (scene factory, w, h, samps_per_cell, seed, normalise)
    "cornell9_32x24_s2_seed1": (lambda: pkg.cornell9(), 32, 24, 2, 1, True),
    "cornell9_e12_40x30_s1_seed0_sum": (lambda: pkg.cornell9(12.0), 40, 30, 1, 0, False),
    "rand64_33x17_s3_seed9": (lambda: pkg.random_spheres(64, 3), 33, 17, 3, 9, True),
    "rand1024_24x18_s1_seed2": (lambda: pkg.random_spheres(1024, 1024), 24, 18, 1, 2, True),
    # D9 with several sample blocks per jitter cell (NB = 8 from 128 samples per cell, NB = 2 from 32): pins the block layout,
    # so that a later change of the kernels' scheduling cannot move the summation spec silently
    "cornell9_16x12_s128_seed5": (lambda: pkg.cornell9(), 16, 12, 128, 5, True),
    "rand1024_12x8_s40_seed3": (lambda: pkg.random_spheres(1024, 1024), 12, 8, 40, 3, False),
}


def main():
    for name, (mk, w, h, samps, seed, norm) in CASES.items():
        img, st = orc.render(mk(), w, h, samps, seed=seed, normalise=norm)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img,
                            bounces=np.uint64(st["bounces"]), samples=np.uint64(st["samples"]))
        print(name, img.shape, st)


if __name__ == "__main__":
    main()
