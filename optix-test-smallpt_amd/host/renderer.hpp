// renderer.hpp -- C++ wrapper with the shape of the reference's render entry points, on top of the C-ABI.
//
//   reference                                                        here
//   Vector<float3> Renderer::render(camera, intersector, materials,  std::vector<float3> Renderer::render(camera, w, h,
//       w, h, sampleCountPerJitterCell, threadCount, seed)               sampleCountPerJitterCell, seed)
//       (smallpt.cpp:679-680,692-814; caller :922)                    -> un-normalised sum, row 0 = bottom
//   Intersector::addTriangleMesh / build (smallpt.cpp:489-530)        Renderer::setScene(spheres)  (scene upload)
//   int cpuRender(argc, argv) (smallpt.cpp:269-379)                   spt_host::offlineRender(...) (normalised)
//
// Errors become std::runtime_error (the reference ignores every rtp* return code, smallpt.cpp:381-393).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "scene.hpp"

namespace spt_host {

class Renderer {
public:
    explicit Renderer(int device = 0)
    {
        if (spt_create(device, &ctx_)) throw std::runtime_error(spt_last_error(nullptr));
    }
    ~Renderer() { spt_destroy(ctx_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void setScene(const std::vector<Sphere>& spheres)
    {
        const std::vector<spt_sphere> abi = to_abi(spheres);
        check(spt_set_scene(ctx_, abi.data(), (uint32_t)abi.size()));
    }

    // Same contract as the reference's Renderer::render: image by value, row-major w*h packed float3,
    // row 0 = bottom, UN-NORMALISED sum of 4*sampleCountPerJitterCell samples per pixel; `seed` is the
    // frame counter of the progressive viewer loop (smallpt.cpp:893,922,926).
    std::vector<float3> render(const spt_camera& camera, size_t imageWidth, size_t imageHeight,
                               size_t sampleCountPerJitterCell, size_t seed, bool normalise = false)
    {
        std::vector<float3> out(imageWidth * imageHeight);
        check(spt_render(ctx_, &camera, (uint32_t)imageWidth, (uint32_t)imageHeight, (uint32_t)sampleCountPerJitterCell,
                         (uint64_t)seed, normalise ? SPT_FLAG_NORMALISE : 0u, reinterpret_cast<float*>(out.data()), &stats_));
        return out;
    }

    const spt_stats& stats() const { return stats_; }
    spt_ctx* handle() { return ctx_; }

private:
    void check(int rc)
    {
        if (rc) throw std::runtime_error(spt_last_error(ctx_));
    }
    spt_ctx* ctx_ = nullptr;
    spt_stats stats_{};
};

}  // namespace spt_host
