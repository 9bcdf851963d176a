// Sanitizer harness for the host C++ that runs without a GPU: the JSON scene reader/writer, the render-request reader,
// the Cornell table and the camera constants (host/scene.cpp).  Built with -fsanitize=address,undefined by
// tests/test_sanitizers.py ("TSan/ASan on host code"); argv[1..] = files with JSON text; every
// text is parsed as a scene and as a request, well-formed scenes are written back and re-read.
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

#include "../../optix-test-smallpt_amd/host/scene.hpp"

using namespace spt_host;

int main(int argc, char** argv)
{
    int ok = 0, bad = 0;
    const Scene c9 = cornell9();
    const std::string text9 = scene_to_json(c9);
    const Scene back = load_scene_json(text9);
    if (back.spheres.size() != 9 || to_abi(back.spheres).size() != 9) return 2;
    const spt_camera cam = make_camera(c9.camera, 1024, 768);
    if (!(cam.push == 140.0f)) return 3;
    for (int i = 1; i < argc; ++i) {
        std::ifstream f(argv[i], std::ios::binary);
        std::ostringstream ss;
        ss << f.rdbuf();
        const std::string text = ss.str();
        try {
            const Scene s = load_scene_json(text);
            const Scene again = load_scene_json(scene_to_json(s));
            if (again.spheres.size() != s.spheres.size()) return 4;
            ++ok;
        } catch (const std::exception&) { ++bad; }
        try {
            float3 org;
            (void)parse_update_camera_request(text, &org);
        } catch (const std::exception&) {}
    }
    std::printf("scenes ok %d rejected %d\n", ok, bad);
    return 0;
}
