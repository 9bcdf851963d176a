"""Multi-GPU front (include/smallpt_mi355x_multi.h, libsmallpt_mi355x_multi.so): the single-process, thread-per-device
host with the RCCL exchange.  CPU part: the library loads (it links librccl), exports what its header declares, its row
partition equals the Python one.  GPU part (one-GPU box): ndev = 1 gives the image of spt_render, also when the band is
routed through RCCL (grouped self send/recv), which is the exchange step of the 8-GPU configuration at world size 1."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smallpt_mi355x_multi.h")


def test_multi_library_exports_header(pkg):
    lib = pkg.load_multi_library()
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(spt_multi_[a-z_0-9]+)\s*\(", text)))
    assert len(declared) >= 11 and sorted(pkg.MULTI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name)
    assert C.sizeof(pkg.SptMultiStats) == 48


def test_row_bands_equal_python_partition(pkg):
    from optix_test_smallpt_amd.distributed import row_band
    lib = pkg.load_multi_library()
    b, c = C.c_uint32(), C.c_uint32()
    for h in (1, 2, 7, 8, 9, 768, 4096, 4099):
        for world in (1, 2, 3, 4, 8):
            total = 0
            for rank in range(world):
                lib.spt_multi_row_band(h, world, rank, C.byref(b), C.byref(c))
                assert (b.value, c.value) == row_band(h, world, rank)
                assert b.value == total
                total += c.value
            assert total == h


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_multi_create_fails_without_gpu(pkg):
    with pytest.raises(pkg.SptError, match="no CPU fallback"):
        pkg.MultiRenderer((0,))
    lib = pkg.load_multi_library()
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.spt_multi_create(ids, 2, 0, C.byref(h)) != 0 and b"distinct" in lib.spt_multi_last_error(None)


@pytest.mark.gpu
@pytest.mark.parametrize("self_exchange", [False, True])
def test_single_device_multi_equals_spt_render(pkg, renderer, self_exchange):
    w, h, samps, seed = 96, 70, 3, 11
    sc = pkg.cornell9()
    renderer.set_scene(sc)
    ref, rst = renderer.render(w, h, samps, seed=seed, normalise=True)
    with pkg.MultiRenderer((0,), self_exchange=self_exchange) as m:
        m.set_scene(sc)
        img, st = m.render(w, h, samps, seed=seed, normalise=True)
        assert np.array_equal(img, ref)
        assert st["bounces"] == rst["bounces"] and st["samples"] == rst["samples"] and st["ndev"] == 1
        if self_exchange:
            assert st["gather_ms"] > 0          # the band really went through ncclSend / ncclRecv
        # second call reuses the buffers; device-resident variant
        img2, _ = m.render(w, h, samps, seed=seed + 1, normalise=True, to_host=False)
        assert img2 is None and m.framebuffer_ptr()


@pytest.mark.gpu
def test_multi_rejects_missing_device(pkg):
    n = torch.cuda.device_count()
    with pytest.raises(pkg.SptError, match="out of range"):
        pkg.MultiRenderer((0, n + 3))


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,contiguous", [(2, False), (3, False), (8, False), (3, True)])
def test_several_ranks_on_one_device_assemble_the_same_image(pkg, renderer, ranks, contiguous):
    """The whole multi-rank machinery of spt_multi_render on the one-GPU box: `ranks` host threads + contexts that share
    device 0 (SPT_MULTI_COPY_EXCHANGE: peer copies instead of RCCL, which needs one device per rank), rows dealt out
    round-robin in blocks of 16 (or contiguous bands), packed rows pulled to the root's staging buffer and scattered into
    the framebuffer.  Ragged height: short last block, ranks with different row counts."""
    w, h, samps, seed = 40, 75, 2, 3
    sc = pkg.cornell9()
    renderer.set_scene(sc)
    ref, rst = renderer.render(w, h, samps, seed=seed, normalise=True)
    with pkg.MultiRenderer((0,) * ranks, contiguous=contiguous, copy_exchange=True) as m:
        m.set_scene(sc)
        img, st = m.render(w, h, samps, seed=seed, normalise=True)
        assert np.array_equal(img, ref), f"{int((img != ref).any(axis=-1).sum())} pixels differ"
        assert st["bounces"] == rst["bounces"] and st["samples"] == rst["samples"] and st["ndev"] == ranks
        img2, _ = m.render(w, h + 6, samps, seed=seed, normalise=True)             # buffers grow, other partition
        ref2, _ = renderer.render(w, h + 6, samps, seed=seed, normalise=True)
        assert np.array_equal(img2, ref2)


@pytest.mark.gpu
def test_multi_progressive_loop_accumulates_oracle_frames(pkg, oracle):
    """The viewer's render loop through the multi-GPU front (spt_multi_progressive_*; three ranks share device 0): two frames,
    a camera change -- the next frame replaces accumBuffer (smallpt.cpp:924-930) -- and two more; accumBuffer on the root equals
    the sum of the oracle's frames, and the front refuses a frame before _begin."""
    w, h, samps = 40, 27, 1
    sc = pkg.cornell9()
    cam, cam2 = pkg.pinhole_camera(), pkg.pinhole_camera(org=(0, -0.99, 0))
    with pkg.MultiRenderer((0, 0, 0), copy_exchange=True) as m:
        m.set_scene(sc)
        with pytest.raises(pkg.SptError, match="progressive_begin"):
            m._check(m._lib.spt_multi_progressive_snapshot(m._h, None))
        m.progressive_begin(w, h)
        for seed in (0, 1):
            st = m.progressive_frame(samps, seed, clear=False, camera=cam)
            assert st["ndev"] == 3 and st["samples"] == w * h * 4 * samps
        acc = sum(oracle.render(sc, w, h, samps, seed=s, normalise=False, camera=cam)[0] for s in (0, 1))
        assert np.array_equal(m.progressive_snapshot(), acc)
        m.progressive_frame(samps, 2, clear=True, camera=cam2)
        m.progressive_frame(samps, 1, clear=False, camera=cam2)
        acc = oracle.render(sc, w, h, samps, seed=2, normalise=False, camera=cam2)[0] + oracle.render(sc, w, h, samps, seed=1, normalise=False, camera=cam2)[0]
        assert np.array_equal(m.progressive_snapshot(), acc)
        m.progressive_end()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["copy3", "self"])
def test_failed_rank_returns_its_error_and_the_object_stays_usable(pkg, renderer, mode):
    """One rank's render is made to fail (kernel watchdog of its context at 0.1 us): spt_multi_render must return non-zero with
    that device's message -- every rank finishes its rows before any part of the exchange is enqueued, so nobody is left waiting in
    ncclRecv -- and the same object renders the correct image once the watchdog is lifted.  Three ranks sharing device 0 (copy
    transport) and one rank whose band goes through RCCL (grouped self send/recv)."""
    w, h, samps, seed = 64, 50, 2, 5
    sc = pkg.cornell9()
    renderer.set_scene(sc)
    ref, rst = renderer.render(w, h, samps, seed=seed, normalise=True)
    kw = dict(copy_exchange=True) if mode == "copy3" else dict(self_exchange=True)
    ids, bad = ((0, 0, 0), 1) if mode == "copy3" else ((0,), 0)
    with pkg.MultiRenderer(ids, **kw) as m:
        m.set_scene(sc)
        m.set_rank_watchdog(bad, 1e-7)
        with pytest.raises(pkg.SptError, match="watchdog"):
            m.render(w, h, samps, seed=seed, normalise=True)
        m.set_rank_watchdog(bad, 60.0)
        img, st = m.render(w, h, samps, seed=seed, normalise=True)
        assert np.array_equal(img, ref) and st["bounces"] == rst["bounces"]


@pytest.mark.gpu
def test_failure_inside_the_exchange_aborts_the_communicators_and_the_next_render_rebuilds_them(pkg, renderer):
    """A failure INSIDE phase 2 (every rank's rows are complete, the send / receives are being enqueued): the failing rank must take every
    communicator down (ncclCommAbort) before it returns, so that no peer stays blocked in a send / receive whose partner never comes;
    spt_multi_render returns that rank's error, and the next call builds new communicators (ncclCommInitAll) and renders the correct
    image.  One rank whose band goes through RCCL (grouped self send / receive) -- the only RCCL exchange a one-GPU box can run --, the
    failure injected by the hook of csrc/spt_internal.h; twice in a row, so that an abort after a rebuild is covered too."""
    w, h, samps, seed = 48, 40, 2, 9
    sc = pkg.cornell9()
    renderer.set_scene(sc)
    ref, rst = renderer.render(w, h, samps, seed=seed, normalise=True)
    with pkg.MultiRenderer((0,), self_exchange=True) as m:
        m.set_scene(sc)
        img, st = m.render(w, h, samps, seed=seed, normalise=True)
        assert np.array_equal(img, ref)
        for _ in range(2):
            m.inject_exchange_failure(0)
            with pytest.raises(pkg.SptError, match="injected exchange failure"):
                m.render(w, h, samps, seed=seed, normalise=True)
            img, st = m.render(w, h, samps, seed=seed, normalise=True)
            assert np.array_equal(img, ref) and st["bounces"] == rst["bounces"]


@pytest.mark.gpu
def test_multi_mesh_scene_and_accel_pass_throughs(pkg, renderer):
    """spt_multi_set_meshes / set_mesh_accel / set_sphere_accel: the triangle scene the reference ships and a large sphere table
    over three ranks sharing the device -- the same images as the single-context renderer."""
    meshes, mats = pkg.single_triangle_scene()
    cam = pkg.pinhole_camera()
    renderer.set_meshes(meshes, mats)
    ref, _ = renderer.render(48, 36, 1, seed=2, camera=cam)
    big = pkg.random_spheres(300, 5)
    with pkg.MultiRenderer((0, 0, 0), copy_exchange=True) as m:
        m.set_mesh_accel(pkg.ACCEL_BVH)
        m.set_meshes(meshes, mats)
        img, _ = m.render(48, 36, 1, seed=2, camera=cam)
        assert np.array_equal(img, ref)
        for accel in (pkg.ACCEL_GRID, pkg.ACCEL_BVH, pkg.ACCEL_EXHAUSTIVE):
            m.set_sphere_accel(accel)
            m.set_scene(big)
            img, st = m.render(40, 33, 1, seed=3, normalise=True)
            renderer.set_scene(big)
            ref2, rst = renderer.render(40, 33, 1, seed=3, normalise=True)
            assert np.array_equal(img, ref2) and st["bounces"] == rst["bounces"], accel
    renderer.set_scene(pkg.cornell9())


@pytest.mark.gpu
def test_config4_full_size_eight_ranks_on_one_device(pkg, oracle):
    """BASELINE config 4 at FULL size -- Cornell-9, 4096 x 4096, 4096 spp, eight ranks -- through the C++ multi-GPU front with the
    eight ranks sharing this box's one device (copy transport; RCCL needs a device per rank, which the one-GPU box cannot give):
    rows dealt out round-robin in blocks of 16, eight packed bands pulled to the root and scattered into the 201 MB framebuffer.
    Rows at block and rank boundaries are compared with full oracle rows.  68.7 G samples: about 7 s of GPU time."""
    w = h = 4096
    samps = 1024
    sc = pkg.cornell9()
    with pkg.MultiRenderer((0,) * 8, copy_exchange=True) as m:
        m.set_scene(sc)
        img, st = m.render(w, h, samps, seed=0, normalise=True)
    assert st["ndev"] == 8 and st["samples"] == w * h * 4 * samps
    for row in (0, 15, 16, 127, 128, 2047, 4095):          # first / last row of a block, of a round of eight blocks, of the image
        ref, _ = oracle.render(sc, w, h, samps, seed=0, normalise=True, row_begin=row, row_count=1)
        assert np.array_equal(img[row:row + 1], ref), row
    rate = st["samples"] / (st["total_ms"] * 1e-3) / 1e9
    print(f"config 4 on one device, 8 ranks sharing it: {st['total_ms'] / 1e3:.2f} s, {rate:.2f} Gsamples/s, exchange {st['gather_ms']:.2f} ms")
