// smallpt_cli.cpp -- offline renderer with the role of the reference's cpuRender(argc, argv)
// (smallpt.cpp:269-379): argv[1] = spp (divided by 4 into samples per jitter cell, :276), renders the scene,
// prints the reference's "Elapsed time" line (:373) and writes ./image.ppm through flipY + writeImage
// (:375-376).  Extra options select the scene file (JSON), the image size and the device.
//
//   smallpt_mi355x [spp] [--scene file.json | shipped-meshes] [--size WxH] [--seed N] [--out image.ppm] [--device D]
//                  [--dump-scene out.json] [--parse-only]
//                  [--accel grid|bvh|bvh-fast|exhaustive]             closest hit of sphere tables above 24 (default grid) / mesh scenes (default bvh)
//                  [--devices 0,1,...] [--self-exchange]      row bands over several GPUs + RCCL exchange (MultiRenderer)
//   smallpt_mi355x [spp] --viewer [--frames N] [--request JSON] [--frames-after M] [--threaded] [--org x,y,z]
//                  [--pipeline L] [--bench-frames N]           L frames in flight (one context each); frames/s of N frames as JSON
//                  [--dump-raw accum.bin]                      main()'s progressive loop (smallpt.cpp:840-1005) without the
//                                                              window: N frames, then the request(s), then M frames; writes the
//                                                              normalised image like the exit path (:995-1004)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "renderer.hpp"
#include "../csrc/spt_internal.h"
#include "viewer.hpp"

using namespace spt_host;

int main(int argc, char* argv[])
{
    int spp = 4, w = 256, h = 256, device = 0;       // smallpt.cpp:274-276 defaults
    unsigned long long seed = 0;
    int accel = -1;                                  // -1: the library's defaults (spheres: grid; meshes: bvh)
    int pipeline = 1, bench_frames = 0;
    double watchdog = 0.0;                           // test hook: kernel watchdog in seconds (csrc/spt_internal.h)
    std::string scene_path, out_path = "image.ppm", dump_path;
    bool single_triangle = false;
    bool parse_only = false, viewer = false, threaded = false, self_exchange = false;
    int frames = 1, frames_after = 0;
    std::vector<int> devices;
    std::vector<std::string> requests;
    std::string dump_raw;
    float org[3] = {0, -1, 0};
    bool have_org = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value after %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--scene") scene_path = next();
        else if (a == "--size") { if (std::sscanf(next(), "%dx%d", &w, &h) != 2 || w <= 0 || h <= 0) { std::fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 10);
        else if (a == "--out") out_path = next();
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--dump-scene") dump_path = next();
        else if (a == "--parse-only") parse_only = true;
        else if (a == "--parse-request") {   // host-only: one message of the viewer's request queue through the JSON reader
            try {
                float3 o3;
                if (parse_update_camera_request(next(), &o3)) std::printf("update_camera %.9g %.9g %.9g\n", o3.x, o3.y, o3.z);
                else std::printf("ignored\n");
                return 0;
            } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
        }
        else if (a == "--accel") { const std::string m = next(); if (m == "bvh") accel = SPT_ACCEL_BVH; else if (m == "bvh-fast") accel = SPT_ACCEL_BVH_FAST; else if (m == "exhaustive") accel = SPT_ACCEL_EXHAUSTIVE; else if (m == "grid") accel = SPT_ACCEL_GRID; else { std::fprintf(stderr, "--accel grid|bvh|bvh-fast|exhaustive\n"); return 2; } }
        else if (a == "--pipeline") { pipeline = std::atoi(next()); if (pipeline < 1 || pipeline > 8) { std::fprintf(stderr, "--pipeline 1..8\n"); return 2; } }
        else if (a == "--bench-frames") bench_frames = std::atoi(next());
        else if (a == "--watchdog") watchdog = std::atof(next());
        else if (a == "--single-triangle") single_triangle = true;   // SingleTriangleScene of main(), smallpt.cpp:818-832
        else if (a == "--viewer") viewer = true;
        else if (a == "--threaded") threaded = true;
        else if (a == "--self-exchange") self_exchange = true;
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--frames-after") frames_after = std::atoi(next());
        else if (a == "--request") requests.push_back(next());
        else if (a == "--dump-raw") dump_raw = next();
        else if (a == "--org") { if (std::sscanf(next(), "%f,%f,%f", &org[0], &org[1], &org[2]) != 3) { std::fprintf(stderr, "--org x,y,z\n"); return 2; } have_org = true; }
        else if (a == "--devices") { const char* p = next(); while (*p) { devices.push_back((int)std::strtol(p, const_cast<char**>(&p), 10)); if (*p == ',') ++p; } }
        else if (a[0] != '-') spp = std::atoi(a.c_str());
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        Scene scene = scene_path.empty() ? cornell9() : (scene_path == "shipped-meshes" ? shipped_two_sphere_mesh_scene() : load_scene_file(scene_path));
        realize_meshes(scene);
        auto upload = [&](Renderer& rr) {          // spheres, or the Intersector seam for a mesh scene
            if (scene.meshes.empty()) { if (accel >= 0) rr.setSphereAccel(accel); rr.setScene(scene.spheres); return; }
            if (accel == SPT_ACCEL_GRID) throw std::runtime_error("--accel grid applies to sphere scenes");
            if (accel >= 0) rr.setMeshAccel(accel);
            std::vector<TriMesh> ms; std::vector<Material> mats;
            for (const MeshInstance& m : scene.meshes) { ms.push_back(m.mesh); mats.push_back(m.material); }
            rr.setMeshes(ms, mats);
        };
        if (!dump_path.empty()) {
            std::ofstream f(dump_path);
            f << scene_to_json(scene) << "\n";
        }
        if (parse_only) {   // host-only path (no GPU): used by the CPU tests of the JSON loader
            const std::vector<spt_sphere> abi = to_abi(scene.spheres);
            std::fwrite(abi.data(), sizeof(spt_sphere), abi.size(), stdout);
            return 0;
        }
        const int samps = spp / 4 > 0 ? spp / 4 : 1;                               // :276
        if (viewer) {
            // main() of the reference (smallpt.cpp:840-1005) without GLFW/GL: render thread + request queue + accumulation
            Renderer renderer(device);
            std::vector<std::unique_ptr<Renderer>> extra;           // --pipeline N: one more context (same device, same scene) per further frame in flight
            for (int k = 1; k < pipeline; ++k) extra.emplace_back(new Renderer(device));
            auto setup = [&](Renderer& rr, bool probe_it) {
                if (!single_triangle) { upload(rr); return; }
                TriMesh triangle;                                                     // smallpt.cpp:826-828
                triangle.positionBuffer = {make_float3(-0.5f, -0.5f, -2), make_float3(0.5f, -0.5f, -2), make_float3(0, 0.5f, -2)};
                triangle.normalBuffer = {make_float3(1, 0, 0), make_float3(0, 1, 0), make_float3(0, 0, 1)};
                triangle.indexBuffer = {0, 1, 2};
                rr.setMeshes({triangle}, {Material{make_float3(1, 0, 0), make_float3(0, 0, 0), DIFF}});   // :821, :830-831
                if (!probe_it) return;
                const Ray probe{make_float3(0, 0, 0), make_float3(0, 0, -1)};
                const std::vector<Hit> hit = rr.traceRays(&probe, 1);
                std::fprintf(stderr, "traceRays probe: dist %.9g uv (%.9g, %.9g) hit %d\n", hit[0].dist, hit[0].uv[0], hit[0].uv[1], (int)(bool)hit[0]);
            };
            setup(renderer, true);
            if (watchdog > 0) spt_set_watchdog(renderer.handle(), watchdog);
            std::vector<Renderer*> lanes;
            for (auto& e : extra) { setup(*e, false); lanes.push_back(e.get()); }
            Camera camera = defaultViewerCamera();
            if (have_org) camera.org = make_float3(org[0], org[1], org[2]);
            ProgressiveRenderer prog(renderer, (size_t)w, (size_t)h, (size_t)samps, camera, lanes);
            auto run_frames = [&](int n) {
                if (n <= 0) return;
                if (threaded) {
                    const size_t target = prog.framesRendered() + (size_t)n;
                    prog.start();
                    while (prog.framesRendered() < target && prog.lastError().empty()) std::this_thread::yield();
                    prog.stop();
                    if (!prog.lastError().empty()) throw std::runtime_error("render thread: " + prog.lastError());
                } else {
                    for (int k = 0; k < n; ++k) prog.stepOnce();
                }
            };
            if (bench_frames > 0) {      // frames per second of the loop (warm-up 10 frames), for bench.py's interactive row
                for (int k = 0; k < 10; ++k) prog.stepOnce();
                prog.flush();
                const auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < bench_frames; ++k) prog.stepOnce();
                prog.flush();
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("{\"frames\": %d, \"pipeline\": %d, \"width\": %d, \"height\": %d, \"spp_per_frame\": %d, \"frames_per_s\": %.1f}\n", bench_frames, pipeline, w, h, 4 * samps, bench_frames / dt);
                return 0;
            }
            run_frames(frames);
            for (const std::string& r : requests) prog.postRequest(r);
            run_frames(frames_after);
            std::vector<float3> image;
            float weight3[3];
            prog.snapshot(image, weight3);       // what the GL loop hands to drawWeightedRGBImage(image, w, h, weight3), :955-962
            std::fprintf(stderr, "viewer: frames rendered %zu, sampleCount %zu, weight %.9g\n", prog.framesRendered(), prog.sampleCount(), weight3[0]);
            if (!dump_raw.empty()) {
                std::ofstream f(dump_raw, std::ios::binary);
                f.write(reinterpret_cast<const char*>(image.data()), (std::streamsize)(image.size() * sizeof(float3)));
            }
            const std::vector<float3> fin = prog.finalImage();                          // :995-1001
            if (spt_write_ppm(out_path.c_str(), reinterpret_cast<const float*>(fin.data()), (uint32_t)w, (uint32_t)h)) {   // :1003-1004
                std::fprintf(stderr, "cannot write %s\n", out_path.c_str());
                return 1;
            }
            return 0;
        }
        const spt_camera cam = make_camera(scene.camera, (uint32_t)w, (uint32_t)h);  // :277-279
        std::fprintf(stderr, "Starting rendering\n");                               // :272
        const auto start = std::chrono::high_resolution_clock::now();
        if (!devices.empty()) {
            MultiRenderer multi(devices, self_exchange);
            multi.setScene(scene.spheres);
            std::vector<float3> c = multi.render(cam, (size_t)w, (size_t)h, (size_t)samps, (size_t)seed, /*normalise=*/true);
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start).count();
            const spt_multi_stats& st = multi.stats();
            std::fprintf(stderr, "Rendering (%d spp) 100.00%%\nElapsed time: %lld ms\n", samps * 4, (long long)ms);
            std::fprintf(stderr, "%u device(s): render %.3f ms, RCCL exchange %.3f ms, %.1f Msamples/s, %.3f bounces/sample\n", st.ndev,
                         st.render_ms, st.gather_ms, st.samples / (st.total_ms * 1e3), (double)st.bounces / (double)st.samples);
            if (spt_write_ppm(out_path.c_str(), reinterpret_cast<const float*>(c.data()), (uint32_t)w, (uint32_t)h)) {
                std::fprintf(stderr, "cannot write %s\n", out_path.c_str());
                return 1;
            }
            return 0;
        }
        Renderer renderer(device);
        upload(renderer);
        renderer.setOneShot(true);                               // cpuRender renders its view once
        std::vector<float3> c = renderer.render(cam, (size_t)w, (size_t)h, (size_t)samps, (size_t)seed, /*normalise=*/true);
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start).count();
        const spt_stats& st = renderer.stats();
        std::fprintf(stderr, "Rendering (%d spp) 100.00%%\nElapsed time: %lld ms\n", samps * 4, (long long)ms);   // :368,373
        std::fprintf(stderr, "kernel %.3f ms, %.1f Msamples/s, %.3f bounces/sample, grid %u x %u\n", st.kernel_ms,
                     st.samples / (st.kernel_ms * 1e3), (double)st.bounces / (double)st.samples, st.grid_blocks, st.block_threads);
        if (spt_write_ppm(out_path.c_str(), reinterpret_cast<const float*>(c.data()), (uint32_t)w, (uint32_t)h)) {  // :375-376
            std::fprintf(stderr, "cannot write %s\n", out_path.c_str());
            return 1;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
