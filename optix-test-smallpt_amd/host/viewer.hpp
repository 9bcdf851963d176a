// viewer.hpp -- the viewer glue of the reference's main() (smallpt.cpp:840-1005) without the window:
//
//   reference                                                       here
//   struct Camera{vx, vy, vz, org, nearPlaneDistance} :607-641        spt_host::Camera (same fields; sampleRay runs in the kernel)
//   std::thread renderThread (:895-942): drain renderRequests,        ProgressiveRenderer::start()/stop(), or stepOnce()
//     render(seed = sampleCount), accumBuffer (+)= outImage            for a deterministic single frame
//   Vector<json> renderRequests + mutex (:889-891, :905-920)          postRequest(json text) -- {"action":"update_camera","org":[x,y,z]}
//   GL loop: weight = 1/(sampleCount*sampleCountPerPixel), image =    snapshot(image, &weight3) -> hand to
//     accumBuffer under the mutex, drawWeightedRGBImage (:955-962)      drawWeightedRGBImage(const float*, w, h, weight[3]) (glutils.h:153)
//   keys UP/DOWN move org.y by 0.01 and post a request (:968-985)     moveCamera(dy)
//   exit: accumBuffer /= sampleCount*spp, flipY, writeImage (:995-1004)  finalImage()
//
// accumBuffer lives in HBM (spt_progressive_*); the mutex guards the frame counter and the request queue like the
// reference's two mutexes, and serialises the snapshot against a frame in flight.
//
// Two frames in flight: the reference overlaps rendering with display (:895-962); on the GPU the end of a 4-spp frame is a few
// long specular chains that leave most of the chip idle, so with extra lanes (further Renderer contexts holding the same scene)
// stepOnce() issues frame k on lane k % lanes without waiting for frame k-1 (spt_progressive_frame_async): the accumulations stay
// in frame order, accumBuffer is bit-identical to the serial loop's, a camera request still replaces the buffer with the frame
// rendered at the running sampleCount (:922-939).
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "renderer.hpp"

namespace spt_host {

struct Camera {                      // smallpt.cpp:607-624: columns of localToWorld
    float3 vx{1, 0, 0}, vy{0, 1, 0}, vz{0, 0, -1}, org{0, -1, 0};
    float nearPlaneDistance = 1.f;
    Camera() = default;
    Camera(float3 vx_, float3 vy_, float3 vz_, float3 org_, float near_) : vx(vx_), vy(vy_), vz(vz_), org(org_), nearPlaneDistance(near_) {}
    spt_camera abi() const;
};

// main()'s camera (smallpt.cpp:885-888,899): vx=(1,0,0), vz=(0,0,-1), vy=normalize(cross(vx,vz)), org=(0,-1,0), near 1
Camera defaultViewerCamera();

// Parses one render request; returns true and fills org for {"action":"update_camera","org":[x,y,z]} (:909-916),
// false for any other well-formed request (ignored like the reference does); throws std::runtime_error on malformed JSON.
bool parseUpdateCamera(const std::string& json, float3* org);

class ProgressiveRenderer {
public:
    // extraLanes: further contexts on the same device with the same scene, one per additional frame in flight
    ProgressiveRenderer(Renderer& renderer, size_t imageWidth, size_t imageHeight, size_t sampleCountPerJitterCell, const Camera& camera,
                        const std::vector<Renderer*>& extraLanes = {});
    ~ProgressiveRenderer();

    void start();                                   // spawns the render thread (:895)
    void stop();                                    // renderDone = true; join (:992-993)
    void stepOnce();                                // one iteration of the thread's while-body (:903-941)

    void postRequest(const std::string& json);      // GL thread side, :978-985; throws std::runtime_error on malformed JSON (caller's thread)
    void flush();                                   // waits for every frame in flight
    std::string lastError();                        // message of the exception that ended the render thread ("" if none)
    void moveCamera(float dy);                      // keys UP (+0.01) / DOWN (-0.01), :968-985

    // :955-959: copies accumBuffer and returns the display weight in weight3 (all three equal, :961)
    void snapshot(std::vector<float3>& image, float weight3[3]);
    std::vector<float3> finalImage();               // :995-1001 (before flipY / writeImage)
    size_t sampleCount();                           // frames accumulated since the last clear
    size_t framesRendered() const { return framesRendered_; }

private:
    Renderer& renderer_;
    std::vector<Renderer*> lanes_;                  // lanes_[0] = &renderer_ (owner of accumBuffer)
    std::vector<char> inFlight_;
    size_t issued_ = 0;
    std::string lastError_;
    size_t w_, h_, samps_;
    Camera camera_;            // render-thread copy (:899)
    float3 org_;               // GL-thread copy moved by the keys (:887)
    std::mutex requestsMutex_, accumMutex_;
    std::vector<std::string> requests_;
    size_t sampleCount_ = 0;   // :893
    std::atomic<size_t> framesRendered_{0};
    std::atomic<bool> renderDone_{false};      // the reference uses a plain float here (:894), a data race
    std::thread thread_;
};

}  // namespace spt_host
