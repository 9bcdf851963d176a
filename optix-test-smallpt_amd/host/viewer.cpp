// viewer.cpp -- see viewer.hpp.
#include "viewer.hpp"

#include <cmath>
#include <cstdio>

namespace spt_host {

spt_camera Camera::abi() const
{
    spt_camera c;
    const float x[3] = {vx.x, vx.y, vx.z}, y[3] = {vy.x, vy.y, vy.z}, z[3] = {vz.x, vz.y, vz.z}, o[3] = {org.x, org.y, org.z};
    if (spt_camera_pinhole(x, y, z, o, nearPlaneDistance, &c)) throw std::runtime_error("spt_camera_pinhole failed");
    return c;
}

Camera defaultViewerCamera()
{
    const float3 vx = make_float3(1, 0, 0), vz = make_float3(0, 0, -1);
    const float3 c = make_float3(vx.y * vz.z - vx.z * vz.y, vx.z * vz.x - vx.x * vz.z, vx.x * vz.y - vx.y * vz.x);   // cross(vx, vz), :886
    const float inv = 1.0f / std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z);
    return Camera(vx, make_float3(c.x * inv, c.y * inv, c.z * inv), vz, make_float3(0, -1, 0), 1.f);
}

bool parseUpdateCamera(const std::string& json, float3* org) { return parse_update_camera_request(json, org); }

ProgressiveRenderer::ProgressiveRenderer(Renderer& renderer, size_t imageWidth, size_t imageHeight, size_t sampleCountPerJitterCell, const Camera& camera,
                                         const std::vector<Renderer*>& extraLanes)
    : renderer_(renderer), w_(imageWidth), h_(imageHeight), samps_(sampleCountPerJitterCell), camera_(camera), org_(camera.org)
{
    renderer_.progressiveBegin(w_, h_);             // accumBuffer.resize(w*h, 0), :881-883
    lanes_.push_back(&renderer_);
    for (Renderer* r : extraLanes) {
        r->progressiveAttach(renderer_);            // own frame buffer + a stream of another priority
        lanes_.push_back(r);
    }
    inFlight_.assign(lanes_.size(), 0);
}

ProgressiveRenderer::~ProgressiveRenderer()
{
    stop();
    try { flush(); } catch (...) {}
    for (size_t i = 1; i < lanes_.size(); ++i) { try { lanes_[i]->progressiveEnd(); } catch (...) {} }
    try { renderer_.progressiveEnd(); } catch (...) {}
}

void ProgressiveRenderer::start()
{
    if (thread_.joinable()) return;
    renderDone_ = false;
    thread_ = std::thread([this] {                   // :895-901
        // an exception must not leave the thread (std::terminate): keep its message, end the loop
        try {
            while (!renderDone_) stepOnce();
            flush();
        } catch (const std::exception& e) {
            std::unique_lock<std::mutex> l{requestsMutex_};
            lastError_ = e.what();
            renderDone_ = true;
        }
    });
}

void ProgressiveRenderer::stop()
{
    renderDone_ = true;                              // :992
    if (thread_.joinable()) thread_.join();          // :993
}

std::string ProgressiveRenderer::lastError()
{
    std::unique_lock<std::mutex> l{requestsMutex_};
    return lastError_;
}

void ProgressiveRenderer::flush()
{
    std::unique_lock<std::mutex> l{accumMutex_};
    for (size_t i = 0; i < lanes_.size(); ++i)
        if (inFlight_[i]) { inFlight_[i] = 0; lanes_[i]->progressiveWait(); }
}

void ProgressiveRenderer::postRequest(const std::string& json)
{
    float3 ignored;
    (void)parseUpdateCamera(json, &ignored);         // malformed text is refused HERE, on the caller's thread, not in the render thread
    std::unique_lock<std::mutex> l{requestsMutex_};  // :980
    requests_.emplace_back(json);
}

void ProgressiveRenderer::moveCamera(float dy)
{
    org_.y += dy;                                    // :969 / :974
    char buf[160];
    std::snprintf(buf, sizeof buf, "{\"action\": \"update_camera\", \"org\": [%.9g, %.9g, %.9g]}", org_.x, org_.y, org_.z);   // :981-984
    postRequest(buf);
}

void ProgressiveRenderer::stepOnce()
{
    bool needClearBuffer = false;                    // :903
    {
        std::unique_lock<std::mutex> l{requestsMutex_};   // :906
        for (const std::string& request : requests_) {    // :909
            float3 newOrg;
            if (parseUpdateCamera(request, &newOrg)) {    // :911-913
                camera_ = Camera{camera_.vx, camera_.vy, camera_.vz, newOrg, camera_.nearPlaneDistance};   // :914
                needClearBuffer = true;                   // :915
            }
        }
        requests_.clear();                           // :918
    }
    size_t seed;
    {
        std::unique_lock<std::mutex> l{accumMutex_};
        seed = sampleCount_;                         // :922 passes the running sampleCount as the seed
    }
    // :922 render + :927-937 accumulate, both on the device; the lock keeps a snapshot from reading a half-added frame
    std::unique_lock<std::mutex> l{accumMutex_};     // :925
    if (lanes_.size() == 1) {
        renderer_.progressiveFrame(camera_.abi(), samps_, seed, needClearBuffer);
    } else {
        const size_t k = issued_++ % lanes_.size();  // frame in flight on lane k: wait for the one it rendered before, issue, return
        if (inFlight_[k]) { inFlight_[k] = 0; lanes_[k]->progressiveWait(); }
        lanes_[k]->progressiveFrameAsync(renderer_, camera_.abi(), samps_, seed, needClearBuffer);
        inFlight_[k] = 1;
    }
    ++sampleCount_;                                  // :926
    if (needClearBuffer) sampleCount_ = 1;           // :938-939
    ++framesRendered_;
}

void ProgressiveRenderer::snapshot(std::vector<float3>& image, float weight3[3])
{
    std::unique_lock<std::mutex> l{accumMutex_};     // :956
    const size_t sampleCountPerPixel = 4 * samps_;   // jitterSize^2 * sampleCountPerJitterCell, :847-848
    const float weight = 1.f / (sampleCount_ * sampleCountPerPixel);   // :957
    image.resize(w_ * h_);
    renderer_.progressiveSnapshot(image);            // image = accumBuffer, :958
    weight3[0] = weight3[1] = weight3[2] = weight;   // :961
}

std::vector<float3> ProgressiveRenderer::finalImage()
{
    std::vector<float3> image;
    float w3[3];
    snapshot(image, w3);
    const float div = (float)(sampleCount_ * 4 * samps_);
    const float inv = 1.0f / div;                    // operator/=(float3, float) multiplies by the reciprocal, :999
    for (float3& p : image) { p.x *= inv; p.y *= inv; p.z *= inv; }
    return image;
}

size_t ProgressiveRenderer::sampleCount()
{
    std::unique_lock<std::mutex> l{accumMutex_};
    return sampleCount_;
}

}  // namespace spt_host
