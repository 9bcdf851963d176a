"""ASan + UBSan over the code that runs on the CPU: the host C++ scene/JSON/request reader and the
oracle.  GPU AddressSanitizer is not available on the pool, so device code is covered by the bit-exact parity tests."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _sanitizers_work(tmp_path):
    src = tmp_path / "probe.c"
    src.write_text("int main(void){return 0;}\n")
    exe = tmp_path / "probe"
    r = subprocess.run(["gcc", *SAN, str(src), "-o", str(exe)], capture_output=True)
    return r.returncode == 0 and subprocess.run([str(exe)], env=ENV).returncode == 0


def test_host_json_reader_under_asan_ubsan(pkg, tmp_path):
    if not _sanitizers_work(tmp_path):
        pytest.skip("libasan/libubsan not usable in this environment")
    exe = tmp_path / "host_json"
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off", *SAN, os.path.join(ROOT, "tests", "sanitize", "host_json_main.cpp"),
                           os.path.join(ROOT, "optix-test-smallpt_amd", "host", "scene.cpp"), "-o", str(exe)])
    good = pkg.spheres_to_json(pkg.random_spheres(64, 3), camera={"origin": [1, 2, 3], "direction": [0, 0, -1], "fov": 0.5, "push": 1})
    texts = [good, good[:-1], good[:len(good) // 2], "", "{", "[]", '{"spheres": 3}', '{"spheres": [{"radius": 1}]}',
             '{"spheres": [{"radius": "x", "center": [1,2,3], "emission": [0,0,0], "color": [1,1,1], "refl": "DIFF"}]}',
             '{"spheres": [], "camera": {"origin": [1, 2]}}', '{"action": "update_camera", "org": [0, -0.99, 0]}',
             '{"action": "update_camera", "org": [0, "a", 0]}', '{"a": "\\u12', '{"a": "\\', "nul", "-", '{"spheres": [' * 2000,
             '{"spheres": [], "x": "' + "\\n" * 5000 + '"}']
    mesh = ('{"meshes": [{"emission": [0,0,0], "color": [1,1,1], "refl": "DIFF", "positions": [[0,0,0],[1,0,0],[0,1,0]], '
            '"normals": [[0,0,1],[0,0,1],[0,0,1]], "indices": [[0, 1, %s]]}]}')
    # triangle indices are read as integers in double: 2.0000001 (which binary32 would round to 2), 3 (= vertex count) and
    # 16777217 (2^24 + 1) are refused, 2 is accepted
    texts += [mesh % "2", mesh % "2.0000001", mesh % "3", mesh % "16777217", mesh % "-1"]
    files = []
    for i, t in enumerate(texts):
        p = tmp_path / f"in{i}.json"
        p.write_text(t)
        files.append(str(p))
    r = subprocess.run([str(exe), *files], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "scenes ok 3 rejected" in r.stdout          # the well-formed sphere scene, the empty-spheres one and the mesh with index 2


def test_oracle_under_asan_ubsan(tmp_path):
    if not _sanitizers_work(tmp_path):
        pytest.skip("libasan/libubsan not usable in this environment")
    exe = tmp_path / "oracle_san"
    subprocess.check_call(["gcc", "-std=c11", "-ffp-contract=off", "-fno-fast-math", *SAN,
                           os.path.join(ROOT, "tests", "sanitize", "oracle_main.c"), os.path.join(ROOT, "oracle", "smallpt_oracle.c"),
                           "-lm", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "oracle sanitizer run ok" in r.stdout


def test_bvh_builder_under_asan_ubsan(tmp_path):
    """The host-side hierarchy builder of SPT_ACCEL_BVH (csrc/spt_bvh.cpp is plain C++: compiled here with g++ against the HIP
    vector-type headers only)."""
    if not _sanitizers_work(tmp_path):
        pytest.skip("libasan/libubsan not usable in this environment")
    exe = tmp_path / "bvh_san"
    subprocess.check_call(["g++", "-std=c++17", *SAN, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "sanitize", "bvh_main.cpp"),
                           os.path.join(ROOT, "optix-test-smallpt_amd", "csrc", "spt_bvh.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "bvh sanitizer run ok 44" in r.stdout


def _tribvh_harness(tmp_path, flags, name):
    exe = tmp_path / name
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", *flags, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "sanitize", "tribvh_main.cpp"),
                           os.path.join(ROOT, "optix-test-smallpt_amd", "csrc", "spt_bvh.cpp"), "-o", str(exe)])
    return exe


def test_triangle_hierarchy_equals_the_exhaustive_loop_on_the_cpu(tmp_path):
    """SPT_ACCEL_BVH of a mesh scene is exhaustive-equivalent for EVERY ray (csrc/spt_tribvh.h): the host builder and the very walk /
    node-test functions the gfx950 kernel calls, against the reference's exhaustive loop on seven scenes x 17 families of rays built to
    break a hierarchy (in a triangle's plane anywhere in it, tilted out of it by 2^-6 ... 2^-26, along edges, across the supporting
    lines of needles, from 300 scene sizes away, ...): 0 mismatches; and the harness is sensitive -- without the plane tree or without
    the line tree it reports mismatches."""
    exe = _tribvh_harness(tmp_path, ["-O2"], "tribvh")
    r = subprocess.run([str(exe), "2000"], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatches 0, tribvh harness ok" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    for knob in ("TRIBVH_NO_PLANES", "TRIBVH_NO_LINES"):
        r = subprocess.run([str(exe), "1000"], capture_output=True, text=True, env=dict(os.environ, **{knob: "1"}))
        assert r.returncode == 1 and "tribvh harness FAILED" in r.stdout, knob


def test_triangle_hierarchy_walks_under_asan_ubsan(tmp_path):
    """The same harness under ASan + UBSan (fewer rays): the ball-tree builder, the validator and the three walks."""
    if not _sanitizers_work(tmp_path):
        pytest.skip("libasan/libubsan not usable in this environment")
    exe = _tribvh_harness(tmp_path, SAN, "tribvh_san")
    r = subprocess.run([str(exe), "150"], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0 and "mismatches 0, tribvh harness ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_sphere_grid_builder_and_walk_under_asan_ubsan(tmp_path):
    """The uniform grid of large sphere tables: host builder (csrc/spt_grid.cpp) and the traversal functions the kernel calls
    (csrc/spt_grid.h) against the exhaustive loop, under ASan/UBSan (the -O2 run of the same harness is tests/test_sphere_accel.py)."""
    if not _sanitizers_work(tmp_path):
        pytest.skip("libasan/libubsan not usable in this environment")
    exe = tmp_path / "grid_san"
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", *SAN, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "sanitize", "grid_main.cpp"),
                           os.path.join(ROOT, "optix-test-smallpt_amd", "csrc", "spt_grid.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=ENV)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "mismatches 0," in r.stdout and "grid harness ok" in r.stdout


def test_task_dealing_is_a_bijection(tmp_path):
    """csrc/spt_deal.h deal_task (which task a queue position stands for in the grid-pool, grid, mega and mesh kernels): every task id exactly
    once over the valid positions, "no task" for the holes of the last stride, for positions at or beyond the end and for the grid-pool
    kernel's nothing-left position -- the property whose violation (a sentinel inside the valid range) made a launch loop for ever
    once during round 4 --, and deal_task_tiles (the triangle hierarchy's kernel: a chunk = an 8 x 8 tile of pixels with one sub-index): every task
    exactly once below deal_tiles_end(), holes only for a tile's part beyond the image (tests/sanitize/deal_main.cpp, under ASan + UBSan)."""
    exe = tmp_path / "deal_main"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "sanitize", "deal_main.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "deal_task ok" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
