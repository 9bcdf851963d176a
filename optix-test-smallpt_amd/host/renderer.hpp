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
#include "../../include/smallpt_mi355x_multi.h"

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
                         (uint64_t)seed, (normalise ? SPT_FLAG_NORMALISE : 0u) | (oneShot_ ? SPT_FLAG_ONE_SHOT : 0u), reinterpret_cast<float*>(out.data()), &stats_));
        return out;
    }

    // accumBuffer of the viewer loop in HBM (spt_progressive_*, smallpt.cpp:881-883,922-937,955-959)
    void progressiveBegin(size_t w, size_t h) { check(spt_progressive_begin(ctx_, (uint32_t)w, (uint32_t)h)); }
    void progressiveFrame(const spt_camera& camera, size_t sampleCountPerJitterCell, size_t seed, bool clear)
    {
        check(spt_progressive_frame(ctx_, &camera, (uint32_t)sampleCountPerJitterCell, (uint64_t)seed, clear ? 1 : 0, &stats_));
    }
    // several frames in flight (spt_progressive_attach / _frame_async / _wait): this context as a lane of `owner`'s accumBuffer
    void progressiveAttach(Renderer& owner) { check(spt_progressive_attach(ctx_, owner.ctx_)); }
    void progressiveFrameAsync(Renderer& owner, const spt_camera& camera, size_t sampleCountPerJitterCell, size_t seed, bool clear)
    {
        check(spt_progressive_frame_async(ctx_, owner.ctx_, &camera, (uint32_t)sampleCountPerJitterCell, (uint64_t)seed, clear ? 1 : 0));
    }
    void progressiveWait() { check(spt_progressive_wait(ctx_, &stats_)); }
    void progressiveSnapshot(std::vector<float3>& image) { check(spt_progressive_snapshot(ctx_, reinterpret_cast<float*>(image.data()))); }
    void progressiveEnd() { check(spt_progressive_end(ctx_)); }

    // The Intersector seam of the reference (smallpt.cpp:427-473 CPUIntersector / :475-603 OptixIntersector):
    //   addTriangleMesh(mesh) for every instance + build()  ->  setMeshes(meshes, materials)   (materials[i] <-> instance i, :170)
    //   Vector<Hit> traceRays(const PathContrib*, size_t)    ->  traceRays(rays, n)
    void setMeshes(const std::vector<TriMesh>& meshes, const std::vector<Material>& materials)
    {
        if (meshes.size() != materials.size()) throw std::runtime_error("setMeshes: one material per mesh instance");
        std::vector<spt_mesh> ms(meshes.size());
        std::vector<spt_material> mats(meshes.size());
        for (size_t i = 0; i < meshes.size(); ++i) {
            ms[i].positions = reinterpret_cast<const float*>(meshes[i].positionBuffer.data());
            ms[i].normals = reinterpret_cast<const float*>(meshes[i].normalBuffer.data());
            ms[i].indices = meshes[i].indexBuffer.data();
            ms[i].nverts = (uint32_t)meshes[i].positionBuffer.size();
            ms[i].ntris = (uint32_t)meshes[i].triangleCount();
            const Material& m = materials[i];
            mats[i] = spt_material{{m.emission.x, m.emission.y, m.emission.z}, {m.color.x, m.color.y, m.color.z}, (int32_t)m.refl, 0u};
        }
        check(spt_set_meshes(ctx_, ms.data(), (uint32_t)ms.size(), mats.data()));
    }
    // the OptixIntersector's acceleration structure (rtpModelUpdate, smallpt.cpp:520-530: SPT_ACCEL_BVH, the default since round 4 -- the same
    // Hit as the loop for every ray) or the CPUIntersector's loop over every triangle (SPT_ACCEL_EXHAUSTIVE); contract in include/smallpt_mi355x.h
    void setMeshAccel(int accel) { check(spt_set_mesh_accel(ctx_, accel)); }
    // the same switch for sphere tables above 24 spheres (exhaustive-equivalent by construction, include/smallpt_mi355x.h)
    void setSphereAccel(int accel) { check(spt_set_sphere_accel(ctx_, accel)); }
    std::vector<Hit> traceRays(const Ray* rays, size_t n)
    {
        std::vector<Hit> hits(n);
        check(spt_trace_rays(ctx_, reinterpret_cast<const spt_ray*>(rays), (uint64_t)n, reinterpret_cast<spt_hit*>(hits.data())));
        return hits;
    }
    // the same query on device buffers (n Ray in, n Hit out, this context's device), enqueued on `hipStream` (nullptr: the context's
    // stream) without waiting: what RTP_BUFFER_TYPE_CUDA_LINEAR buffers are to the reference's Prime query (smallpt.cpp:571-575)
    void traceRaysDevice(const void* dRays, size_t n, void* dHits, void* hipStream = nullptr)
    {
        check(spt_trace_rays_device(ctx_, dRays, (uint64_t)n, dHits, hipStream));
    }

    const spt_stats& stats() const { return stats_; }
    spt_ctx* handle() { return ctx_; }
    // a caller that renders a view once (cpuRender, smallpt.cpp:269-379): the launch records no dispatch order for a repetition (SPT_FLAG_ONE_SHOT)
    void setOneShot(bool on) { oneShot_ = on; }

private:
    void check(int rc)
    {
        if (rc) throw std::runtime_error(spt_last_error(ctx_));
    }
    spt_ctx* ctx_ = nullptr;
    spt_stats stats_{};
    bool oneShot_ = false;
};

// The same call spread over the GPUs of one node (include/smallpt_mi355x_multi.h): one host thread + context per
// device, contiguous row bands, one RCCL exchange into the root device's framebuffer.  The reference is single-device
// (smallpt.cpp:480-481); the image is bit-identical for every device count.
class MultiRenderer {
public:
    explicit MultiRenderer(const std::vector<int>& devices, bool selfExchange = false)
    {
        if (spt_multi_create(devices.data(), (int)devices.size(), selfExchange ? SPT_MULTI_SELF_EXCHANGE : 0u, &m_))
            throw std::runtime_error(spt_multi_last_error(nullptr));
    }
    ~MultiRenderer() { spt_multi_destroy(m_); }
    MultiRenderer(const MultiRenderer&) = delete;
    MultiRenderer& operator=(const MultiRenderer&) = delete;

    void setScene(const std::vector<Sphere>& spheres)
    {
        const std::vector<spt_sphere> abi = to_abi(spheres);
        check(spt_multi_set_scene(m_, abi.data(), (uint32_t)abi.size()));
    }
    // the triangle seam and the closest-hit modes on every device (cf. Renderer::setMeshes / setMeshAccel / setSphereAccel)
    void setMeshes(const std::vector<TriMesh>& meshes, const std::vector<Material>& materials)
    {
        if (meshes.size() != materials.size()) throw std::runtime_error("setMeshes: one material per mesh instance");
        std::vector<spt_mesh> ms(meshes.size());
        std::vector<spt_material> mats(meshes.size());
        for (size_t i = 0; i < meshes.size(); ++i) {
            ms[i].positions = reinterpret_cast<const float*>(meshes[i].positionBuffer.data());
            ms[i].normals = reinterpret_cast<const float*>(meshes[i].normalBuffer.data());
            ms[i].indices = meshes[i].indexBuffer.data();
            ms[i].nverts = (uint32_t)meshes[i].positionBuffer.size();
            ms[i].ntris = (uint32_t)meshes[i].triangleCount();
            const Material& m = materials[i];
            mats[i] = spt_material{{m.emission.x, m.emission.y, m.emission.z}, {m.color.x, m.color.y, m.color.z}, (int32_t)m.refl, 0u};
        }
        check(spt_multi_set_meshes(m_, ms.data(), (uint32_t)ms.size(), mats.data()));
    }
    void setMeshAccel(int accel) { check(spt_multi_set_mesh_accel(m_, accel)); }
    void setSphereAccel(int accel) { check(spt_multi_set_sphere_accel(m_, accel)); }
    std::vector<float3> render(const spt_camera& camera, size_t imageWidth, size_t imageHeight,
                               size_t sampleCountPerJitterCell, size_t seed, bool normalise = false)
    {
        std::vector<float3> out(imageWidth * imageHeight);
        check(spt_multi_render(m_, &camera, (uint32_t)imageWidth, (uint32_t)imageHeight, (uint32_t)sampleCountPerJitterCell,
                               (uint64_t)seed, (normalise ? SPT_FLAG_NORMALISE : 0u) | (oneShot_ ? SPT_FLAG_ONE_SHOT : 0u), reinterpret_cast<float*>(out.data()), &stats_));
        return out;
    }
    const spt_multi_stats& stats() const { return stats_; }
    void setOneShot(bool on) { oneShot_ = on; }                  // as Renderer::setOneShot, on every device
    // the render thread's loop over all devices (smallpt.cpp:895-942; spt_multi_progressive_*): accumBuffer on the root device
    void progressiveBegin(size_t imageWidth, size_t imageHeight)
    {
        check(spt_multi_progressive_begin(m_, (uint32_t)imageWidth, (uint32_t)imageHeight));
        pw_ = imageWidth; ph_ = imageHeight;
    }
    void progressiveFrame(const spt_camera& camera, size_t sampleCountPerJitterCell, size_t seed, bool clear)
    {
        check(spt_multi_progressive_frame(m_, &camera, (uint32_t)sampleCountPerJitterCell, (uint64_t)seed, clear ? 1 : 0, &stats_));
    }
    std::vector<float3> progressiveSnapshot()
    {
        std::vector<float3> out(pw_ * ph_);
        check(spt_multi_progressive_snapshot(m_, reinterpret_cast<float*>(out.data())));
        return out;
    }
    void progressiveEnd() { check(spt_multi_progressive_end(m_)); }

private:
    void check(int rc)
    {
        if (rc) throw std::runtime_error(spt_multi_last_error(m_));
    }
    size_t pw_ = 0, ph_ = 0;
    spt_multi* m_ = nullptr;
    spt_multi_stats stats_{};
    bool oneShot_ = false;
};

}  // namespace spt_host
