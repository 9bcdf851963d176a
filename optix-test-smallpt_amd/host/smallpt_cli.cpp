// smallpt_cli.cpp -- offline renderer with the role of the reference's cpuRender(argc, argv)
// (smallpt.cpp:269-379): argv[1] = spp (divided by 4 into samples per jitter cell, :276), renders the scene,
// prints the reference's "Elapsed time" line (:373) and writes ./image.ppm through flipY + writeImage
// (:375-376).  Extra options select the scene file (JSON), the image size and the device.
//
//   smallpt_mi355x [spp] [--scene file.json] [--size WxH] [--seed N] [--out image.ppm] [--device D]
//                  [--dump-scene out.json] [--parse-only]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "renderer.hpp"

using namespace spt_host;

int main(int argc, char* argv[])
{
    int spp = 4, w = 256, h = 256, device = 0;       // smallpt.cpp:274-276 defaults
    unsigned long long seed = 0;
    std::string scene_path, out_path = "image.ppm", dump_path;
    bool parse_only = false;
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
        else if (a[0] != '-') spp = std::atoi(a.c_str());
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        Scene scene = scene_path.empty() ? cornell9() : load_scene_file(scene_path);
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
        const spt_camera cam = make_camera(scene.camera, (uint32_t)w, (uint32_t)h);  // :277-279
        std::fprintf(stderr, "Starting rendering\n");                               // :272
        const auto start = std::chrono::high_resolution_clock::now();
        Renderer renderer(device);
        renderer.setScene(scene.spheres);
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
