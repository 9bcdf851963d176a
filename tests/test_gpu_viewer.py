"""f3 viewer glue in C++ (host/viewer.hpp: the render thread of smallpt.cpp:895-942 with its JSON request queue, accumBuffer
in HBM, mutex-guarded snapshot + display weight of :955-962) and the C++ multi-GPU host (host/renderer.hpp MultiRenderer),
driven headless through the CLI and compared with summed oracle frames."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(os.path.dirname(HERE), "optix-test-smallpt_amd", "host", "smallpt_mi355x")


def _scene_file(pkg, tmp_path):
    from test_gpu_parity import pinhole_scene
    sc = pinhole_scene(pkg)
    p = tmp_path / "viewer_scene.json"
    p.write_text(pkg.spheres_to_json(sc))
    return sc, p


def _run(args):
    r = subprocess.run([CLI] + [str(a) for a in args], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()
    return r.stderr.decode()


def test_render_thread_requests_and_snapshot(pkg, oracle, tmp_path):
    """Two frames, one update_camera request, two more frames (deterministic stepOnce mode): the frame after the request
    is rendered with the running sampleCount (2) as seed and REPLACES accumBuffer (:922-937), then sampleCount = 1 (:938)."""
    from test_gpu_parity import expected_ppm
    w, h, samps = 64, 36, 1
    sc, scene = _scene_file(pkg, tmp_path)
    raw, ppm = tmp_path / "accum.bin", tmp_path / "image.ppm"
    err = _run([4 * samps, "--viewer", "--scene", scene, "--size", f"{w}x{h}", "--frames", 2,
                "--request", '{"action": "update_camera", "org": [0, -0.99, 0]}', "--request", '{"action": "noop"}',
                "--frames-after", 2, "--dump-raw", raw, "--out", ppm])
    m = re.search(r"frames rendered (\d+), sampleCount (\d+), weight ([0-9.eE+-]+)", err)
    assert m and int(m.group(1)) == 4 and int(m.group(2)) == 2
    cam2 = pkg.pinhole_camera(org=(0, -0.99, 0))
    f2, _ = oracle.render(sc, w, h, samps, seed=2, normalise=False, camera=cam2)
    f1, _ = oracle.render(sc, w, h, samps, seed=1, normalise=False, camera=cam2)
    accum = np.fromfile(raw, dtype=np.float32).reshape(h, w, 3)
    assert np.array_equal(accum, f2 + f1)
    weight = np.float32(1.0) / np.float32(2 * 4 * samps)                      # :957
    assert np.float32(m.group(3)) == weight
    # exit path :995-1004: accumBuffer /= sampleCount*spp (multiplication by the reciprocal), flipY, writeImage
    assert ppm.read_bytes() == expected_ppm(oracle, (f2 + f1) * weight)


@pytest.mark.parametrize("threaded", [False, True])
def test_two_frames_in_flight_accumulate_the_same_buffer(pkg, oracle, tmp_path, threaded):
    """--pipeline L (2 and 4 here): further contexts on streams of other priorities, frame k issued on lane k % L without waiting for frame k-1
    (spt_progressive_frame_async), accumulations chained by events.  accumBuffer must be byte-identical to the serial loop's --
    five frames, a camera request, four more: the clearing frame is rendered at the running sampleCount (:922-939) on whichever lane
    is due -- and identical to the oracle's summed frames."""
    w, h, samps = 64, 36, 1
    sc, scene = _scene_file(pkg, tmp_path)
    req = '{"action": "update_camera", "org": [0, -0.99, 0]}'
    raws = {}
    for pipe in (1, 2, 4):
        raw = tmp_path / f"accum{pipe}.bin"
        args = [4 * samps, "--viewer", "--scene", scene, "--size", f"{w}x{h}", "--pipeline", pipe, "--dump-raw", raw, "--out", tmp_path / f"p{pipe}.ppm"]
        if threaded:
            err = _run(args + ["--threaded", "--frames", 6])
            n = int(re.search(r"frames rendered (\d+)", err).group(1))
            assert n >= 6
            acc = np.zeros((h, w, 3), dtype=np.float32)
            cam = pkg.pinhole_camera()
            for frame in range(n):
                acc = acc + oracle.render(sc, w, h, samps, seed=frame, normalise=False, camera=cam)[0]
            assert np.array_equal(np.fromfile(raw, dtype=np.float32).reshape(h, w, 3), acc), pipe
            continue
        err = _run(args + ["--frames", 5, "--request", req, "--frames-after", 4])
        m = re.search(r"frames rendered (\d+), sampleCount (\d+)", err)
        assert int(m.group(1)) == 9 and int(m.group(2)) == 4
        raws[pipe] = raw.read_bytes()
    if not threaded:
        assert raws[1] == raws[2] == raws[4]
        cam2 = pkg.pinhole_camera(org=(0, -0.99, 0))
        acc = oracle.render(sc, w, h, samps, seed=5, normalise=False, camera=cam2)[0]        # the clearing frame: seed = running sampleCount
        for seed in (1, 2, 3):
            acc = acc + oracle.render(sc, w, h, samps, seed=seed, normalise=False, camera=cam2)[0]
        assert np.array_equal(np.frombuffer(raws[2], dtype=np.float32).reshape(h, w, 3), acc)


def test_render_thread_failure_is_reported_not_fatal(pkg, tmp_path):
    """An exception in the render thread (here: the kernel watchdog at 0.1 us makes every frame fail) must end the loop with its
    message -- exit code 1 and "render thread: ..." on stderr -- not std::terminate (SIGABRT); a malformed request is refused on
    the posting thread."""
    sc, scene = _scene_file(pkg, tmp_path)
    r = subprocess.run([CLI, "4", "--viewer", "--threaded", "--scene", str(scene), "--size", "640x360", "--frames", "2", "--watchdog", "1e-7",
                        "--out", str(tmp_path / "x.ppm")], capture_output=True, timeout=300)
    assert r.returncode == 1 and b"watchdog" in r.stderr, (r.returncode, r.stderr[-500:])
    r = subprocess.run([CLI, "4", "--viewer", "--scene", str(scene), "--size", "32x18", "--frames", "1", "--request", '{"action": "update_camera", "org": [1, 2',
                        "--out", str(tmp_path / "y.ppm")], capture_output=True, timeout=300)
    assert r.returncode == 1 and r.stderr, (r.returncode, r.stderr[-500:])


def test_render_thread_runs_concurrently(pkg, oracle, tmp_path):
    """Threaded mode (:895-901): the thread renders until stopped; whatever number of frames N it got to, accumBuffer is
    the sum of the oracle's frames 0..N-1 and the weight is 1/(N*spp)."""
    w, h, samps = 48, 27, 1
    sc, scene = _scene_file(pkg, tmp_path)
    raw = tmp_path / "accum.bin"
    err = _run([4, "--viewer", "--threaded", "--scene", scene, "--size", f"{w}x{h}", "--frames", 3, "--dump-raw", raw,
                "--out", tmp_path / "t.ppm"])
    m = re.search(r"frames rendered (\d+), sampleCount (\d+), weight ([0-9.eE+-]+)", err)
    n = int(m.group(1))
    assert n >= 3 and int(m.group(2)) == n
    cam = pkg.pinhole_camera()
    acc = np.zeros((h, w, 3), dtype=np.float32)
    for frame in range(n):
        acc = acc + oracle.render(sc, w, h, samps, seed=frame, normalise=False, camera=cam)[0]
    assert np.array_equal(np.fromfile(raw, dtype=np.float32).reshape(h, w, 3), acc)
    assert np.float32(m.group(3)) == np.float32(1.0) / np.float32(n * 4 * samps)


@pytest.mark.parametrize("extra", [[], ["--self-exchange"]])
def test_cpp_multi_renderer_cli(pkg, oracle, tmp_path, extra):
    """C++ MultiRenderer (thread per device + RCCL exchange) behind the cpuRender-shaped CLI; one device here."""
    from test_gpu_parity import expected_ppm
    out = tmp_path / "multi.ppm"
    err = _run([16, "--size", "70x51", "--seed", 5, "--devices", "0", "--out", out] + extra)
    assert "1 device(s)" in err and "RCCL exchange" in err
    ref, _ = oracle.render(pkg.cornell9(), 70, 51, 4, seed=5, normalise=True)
    assert out.read_bytes() == expected_ppm(oracle, ref)


def test_main_single_triangle_scene_through_cpp_intersector(pkg, oracle, tmp_path):
    """main()'s actual scene (SingleTriangleScene, smallpt.cpp:818-838) through the C++ Intersector seam
    (Renderer::setMeshes = addTriangleMesh + build, traceRays) and the progressive loop: 2 frames summed."""
    w, h = 160, 90
    raw = tmp_path / "tri.bin"
    err = _run([4, "--viewer", "--single-triangle", "--size", f"{w}x{h}", "--frames", 2, "--dump-raw", raw, "--out", tmp_path / "tri.ppm"])
    assert "traceRays probe: dist 2 uv (0.25, 0.5) hit 1" in err                  # the SURVEY.md 8(c) triIntersect KAT
    meshes, mats = pkg.single_triangle_scene()
    cam = pkg.pinhole_camera()
    acc = sum(oracle.render_meshes(meshes, mats, w, h, 1, seed=s, normalise=False, camera=cam)[0] for s in range(2))
    got = np.fromfile(raw, dtype=np.float32).reshape(h, w, 3)
    assert np.array_equal(got, acc) and got[:, :, 0].max() > 0
