// spt_multi.cpp -- include/smallpt_mi355x_multi.h: one host thread + context + stream per device, row bands, and
// one RCCL exchange step (ncclSend per band, grouped ncclRecv into the root's framebuffer slices).
#include "../../include/smallpt_mi355x_multi.h"
#include "spt_internal.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_multi_create_error;

// One persistent worker per device: jobs are closures run with that device current (the reference spawns detached
// threads per parallel section, ThreadUtils.h:29-48; here the threads live as long as the spt_multi).
class Worker {
public:
    Worker() : thread_([this] { loop(); }) {}
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }
    void submit(std::function<void()> job)        // by value: nothing dangles (cf. the capture bug at ThreadUtils.h:133)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = std::move(job);
            busy_ = true;
        }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return !busy_; });
    }

private:
    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return quit_ || (busy_ && job_); });
                if (quit_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            job();
            {
                std::lock_guard<std::mutex> l(m_);
                busy_ = false;
            }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false;
    std::thread thread_;
};

struct Rank {
    int device = 0;
    spt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    ncclComm_t comm = nullptr;
    float* d_band = nullptr;       // non-root ranks (and the root in self-exchange mode): this rank's rows
    size_t band_cap = 0;           // in floats
    std::string error;             // set by the rank's job
    spt_stats stats{};
    float gather_ms = 0.f;
    Worker* worker = nullptr;
};

}  // namespace

struct spt_multi {
    std::vector<Rank> ranks;
    uint32_t flags = 0;
    bool use_rccl = false;
    float* d_frame = nullptr;      // root device: w*h*3 floats
    size_t frame_cap = 0;
    float* d_staging = nullptr;    // root device, interleaved partition: the other ranks' packed rows before the scatter
    size_t staging_cap = 0;
    float* d_accum = nullptr;      // root device: accumBuffer of the progressive loop (spt_multi_progressive_*), w*h*3 floats
    uint32_t prog_w = 0, prog_h = 0;
    std::string error;
    // failure inside the exchange (phase 2 of spt_multi_render): the first rank that sees an error aborts EVERY communicator so that
    // no peer stays blocked in a send / receive whose partner will never come; the next render builds new communicators
    std::vector<int> device_ids;
    std::mutex comm_mutex;
    bool comms_alive = false;
    int fail_exchange_rank = -1;   // test hook (spt_internal.h spt_multi_inject_exchange_failure): this rank fails inside its next exchange

    void abort_comms()
    {
        std::lock_guard<std::mutex> l(comm_mutex);
        if (!comms_alive) return;
        comms_alive = false;
        for (auto& r : ranks)
            if (r.comm) { (void)ncclCommAbort(r.comm); r.comm = nullptr; }      // frees the communicator; in-flight operations end with an error
    }
    int init_comms()
    {
        std::lock_guard<std::mutex> l(comm_mutex);
        if (comms_alive) return 0;
        std::vector<ncclComm_t> comms(ranks.size());
        const ncclResult_t e = ncclCommInitAll(comms.data(), (int)ranks.size(), device_ids.data());
        if (e != ncclSuccess) return fail("ncclCommInitAll over %d device(s): %s", (int)ranks.size(), ncclGetErrorString(e));
        for (size_t i = 0; i < ranks.size(); ++i) ranks[i].comm = comms[i];
        comms_alive = true;
        return 0;
    }

    int fail(const char* fmt, ...)
    {
        char buf[768];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        return 1;
    }
    // runs fn(rank index) on every rank's worker thread and collects the first error
    int on_all(const std::function<void(int)>& fn)
    {
        for (size_t i = 0; i < ranks.size(); ++i) {
            ranks[i].error.clear();
            ranks[i].worker->submit([&fn, i] { fn((int)i); });
        }
        for (auto& r : ranks) r.worker->wait();
        for (size_t i = 0; i < ranks.size(); ++i)
            if (!ranks[i].error.empty()) return fail("device %d: %s", ranks[i].device, ranks[i].error.c_str());
        return 0;
    }
};

#define RK_HIP(r, call)                                                                        \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) { (r).error = std::string(#call) + ": " + hipGetErrorString(e__); return; } \
    } while (0)
#define RK_NCCL(r, call)                                                                       \
    do {                                                                                       \
        ncclResult_t e__ = (call);                                                             \
        if (e__ != ncclSuccess) { (r).error = std::string(#call) + ": " + ncclGetErrorString(e__); return; } \
    } while (0)
// inside the exchange: a failing rank takes every communicator down before it returns (spt_multi::abort_comms)
#define XK_HIP(m, r, call)                                                                     \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) { (r).error = std::string(#call) + ": " + hipGetErrorString(e__); (m)->abort_comms(); return; } \
    } while (0)
#define XK_NCCL(m, r, call)                                                                    \
    do {                                                                                       \
        ncclResult_t e__ = (call);                                                             \
        if (e__ != ncclSuccess) { (r).error = std::string(#call) + ": " + ncclGetErrorString(e__); (m)->abort_comms(); return; } \
    } while (0)

extern "C" {

const char* spt_multi_last_error(const spt_multi* m) { return m ? m->error.c_str() : g_multi_create_error.c_str(); }
int spt_multi_device_count(const spt_multi* m) { return m ? (int)m->ranks.size() : 0; }

void spt_multi_row_band(uint32_t h, uint32_t world, uint32_t rank, uint32_t* row_begin, uint32_t* row_count)
{
    const uint32_t base = h / world, extra = h % world;
    if (row_count) *row_count = base + (rank < extra ? 1u : 0u);
    if (row_begin) *row_begin = rank * base + (rank < extra ? rank : extra);
}

void spt_multi_destroy(spt_multi* m)
{
    if (!m) return;
    if (!m->ranks.empty() && m->ranks[0].worker) {
        m->on_all([m](int i) {
            Rank& r = m->ranks[(size_t)i];
            (void)hipSetDevice(r.device);
            if (r.stream) (void)hipStreamSynchronize(r.stream);
            if (r.comm) (void)ncclCommDestroy(r.comm);             // (null after an aborted exchange)
            if (r.d_band) (void)hipFree(r.d_band);
            if (i == 0 && m->d_frame) (void)hipFree(m->d_frame);
            if (i == 0 && m->d_staging) (void)hipFree(m->d_staging);
            if (i == 0 && m->d_accum) (void)hipFree(m->d_accum);
            if (r.ev_a) (void)hipEventDestroy(r.ev_a);
            if (r.ev_b) (void)hipEventDestroy(r.ev_b);
            if (r.stream) (void)hipStreamDestroy(r.stream);
            if (r.ctx) spt_destroy(r.ctx);
        });
    }
    for (auto& r : m->ranks) delete r.worker;
    delete m;
}

int spt_multi_create(const int* device_ids, int ndev, uint32_t flags, spt_multi** out)
{
    if (!out) { g_multi_create_error = "spt_multi_create: out is NULL"; return 1; }
    *out = nullptr;
    if (!device_ids || ndev < 1) { g_multi_create_error = "spt_multi_create: need at least one device id"; return 1; }
    if (!(flags & SPT_MULTI_COPY_EXCHANGE))      // RCCL needs one device per rank; the copy transport also runs ranks that share a device
        for (int i = 0; i < ndev; ++i)
            for (int j = 0; j < i; ++j)
                if (device_ids[i] == device_ids[j]) { g_multi_create_error = "spt_multi_create: device ids must be distinct"; return 1; }
    spt_multi* m = nullptr;
    try {                                   // no exception may cross the C boundary (thread creation, allocations)
    m = new spt_multi;
    m->flags = flags;
    m->use_rccl = !(flags & SPT_MULTI_COPY_EXCHANGE) && (ndev > 1 || (flags & SPT_MULTI_SELF_EXCHANGE));
    m->ranks.resize((size_t)ndev);
    for (int i = 0; i < ndev; ++i) {
        m->ranks[(size_t)i].device = device_ids[i];
        m->ranks[(size_t)i].worker = new Worker;
    }
    // contexts, streams, events -- each on its own thread with its device current
    int rc = m->on_all([m](int i) {
        Rank& r = m->ranks[(size_t)i];
        if (spt_create(r.device, &r.ctx)) { r.error = spt_last_error(nullptr); return; }
        RK_HIP(r, hipSetDevice(r.device));
        RK_HIP(r, hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
        RK_HIP(r, hipEventCreate(&r.ev_a));
        RK_HIP(r, hipEventCreate(&r.ev_b));
    });
    m->device_ids.assign(device_ids, device_ids + ndev);
    if (rc == 0 && m->use_rccl) rc = m->init_comms();       // single-process communicators over all devices (rank i = device_ids[i]); ncclCommInitAll handles the grouping
    if (rc) {
        g_multi_create_error = "spt_multi_create: " + m->error;
        spt_multi_destroy(m);
        return 1;
    }
    *out = m;
    return 0;
    } catch (const std::exception& e) {
        g_multi_create_error = std::string("spt_multi_create: ") + e.what();
        if (m) { for (auto& r : m->ranks) delete r.worker; delete m; }
        return 1;
    }
}

int spt_multi_set_scene(spt_multi* m, const spt_sphere* spheres, uint32_t n)
{
    if (!m) return 1;
    try {
    return m->on_all([m, spheres, n](int i) {
        Rank& r = m->ranks[(size_t)i];
        if (spt_set_scene(r.ctx, spheres, n)) r.error = spt_last_error(r.ctx);
    });
    } catch (const std::exception& e) {
        return m->fail("spt_multi_set_scene: %s", e.what());
    }
}

// The triangle seam and the closest-hit modes on every device (spt_set_meshes / spt_set_mesh_accel / spt_set_sphere_accel)
int spt_multi_set_meshes(spt_multi* m, const spt_mesh* meshes, uint32_t nmesh, const spt_material* materials)
{
    if (!m) return 1;
    try {
    return m->on_all([m, meshes, nmesh, materials](int i) {
        Rank& r = m->ranks[(size_t)i];
        if (spt_set_meshes(r.ctx, meshes, nmesh, materials)) r.error = spt_last_error(r.ctx);
    });
    } catch (const std::exception& e) {
        return m->fail("spt_multi_set_meshes: %s", e.what());
    }
}

int spt_multi_set_mesh_accel(spt_multi* m, int accel)
{
    if (!m) return 1;
    try {
    return m->on_all([m, accel](int i) {
        Rank& r = m->ranks[(size_t)i];
        if (spt_set_mesh_accel(r.ctx, accel)) r.error = spt_last_error(r.ctx);
    });
    } catch (const std::exception& e) {
        return m->fail("spt_multi_set_mesh_accel: %s", e.what());
    }
}

int spt_multi_set_sphere_accel(spt_multi* m, int accel)
{
    if (!m) return 1;
    try {
    return m->on_all([m, accel](int i) {
        Rank& r = m->ranks[(size_t)i];
        if (spt_set_sphere_accel(r.ctx, accel)) r.error = spt_last_error(r.ctx);
    });
    } catch (const std::exception& e) {
        return m->fail("spt_multi_set_sphere_accel: %s", e.what());
    }
}

// Test hook (csrc/spt_internal.h): kernel watchdog of ONE rank's context -- lets a test make exactly one rank's render fail
int spt_multi_set_rank_watchdog(spt_multi* m, uint32_t rank, double seconds)
{
    if (!m || rank >= m->ranks.size()) return 1;
    return spt_set_watchdog(m->ranks[rank].ctx, seconds);
}

// Test hook (csrc/spt_internal.h): `rank` fails inside its part of the next RCCL exchange, after every rank's rows are complete
int spt_multi_inject_exchange_failure(spt_multi* m, uint32_t rank)
{
    if (!m || rank >= m->ranks.size()) return 1;
    m->fail_exchange_rank = (int)rank;
    return 0;
}

void* spt_multi_framebuffer(spt_multi* m) { return m ? m->d_frame : nullptr; }

int spt_multi_render(spt_multi* m, const spt_camera* cam, uint32_t w, uint32_t h, uint32_t samps, uint64_t seed,
                     uint32_t flags, float* out_rgb, spt_multi_stats* stats)
{
    if (!m) return 1;
    try {
    if (!cam) return m->fail("spt_multi_render: camera is NULL");
    if (w == 0 || h == 0 || samps == 0) return m->fail("spt_multi_render: empty image or samps == 0");
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t world = (uint32_t)m->ranks.size();
    const size_t nfl = (size_t)w * h * 3;
    const bool self_exchange = world == 1 && m->use_rccl;

    // Row partition: contiguous bands (SPT_MULTI_CONTIGUOUS, or a single device) or -- default -- rows dealt out round-robin
    // in blocks of kBlockRows rows, which balances the ranks (contiguous bands of a Cornell-like image differ by up to 1.34x).
    constexpr uint32_t kBlockRows = 16;
    const bool interleaved = world > 1 && !(m->flags & SPT_MULTI_CONTIGUOUS);
    auto rows_of = [=](uint32_t p, uint32_t* begin) -> uint32_t {
        uint32_t b = 0, c = 0;
        if (interleaved) c = spt_interleaved_row_count(h, kBlockRows, world, p);
        else spt_multi_row_band(h, world, p, &b, &c);
        if (begin) *begin = b;
        return c;
    };
    // staging offsets (in floats) of the ranks' packed rows on the root, interleaved mode
    std::vector<size_t> stage_off(world + 1, 0);
    for (uint32_t p = 0; p < world; ++p) stage_off[p + 1] = stage_off[p] + (interleaved && p != 0 ? (size_t)rows_of(p, nullptr) * w * 3 : 0);

    // 1. buffers + band render, every device on its own thread (ranks without rows -- more devices than rows -- skip).  Every rank
    // finishes its rows (stream synchronised, spt_sync) BEFORE any rank enqueues its part of the exchange: a rank whose render
    // fails would otherwise leave its peers waiting in ncclRecv / ncclSend for ever.
    int rc = m->on_all([=, &stage_off](int i) {
        Rank& r = m->ranks[(size_t)i];
        uint32_t begin = 0;
        const uint32_t count = rows_of((uint32_t)i, &begin);
        RK_HIP(r, hipSetDevice(r.device));
        if (i == 0 && nfl > m->frame_cap) {
            if (m->d_frame) (void)hipFree(m->d_frame);
            m->d_frame = nullptr; m->frame_cap = 0;
            RK_HIP(r, hipMalloc(reinterpret_cast<void**>(&m->d_frame), nfl * sizeof(float)));
            m->frame_cap = nfl;
        }
        if (i == 0 && stage_off[world] > m->staging_cap) {
            if (m->d_staging) (void)hipFree(m->d_staging);
            m->d_staging = nullptr; m->staging_cap = 0;
            RK_HIP(r, hipMalloc(reinterpret_cast<void**>(&m->d_staging), stage_off[world] * sizeof(float)));
            m->staging_cap = stage_off[world];
        }
        const size_t band_fl = (size_t)count * w * 3;
        float* dst;
        if (i == 0 && !self_exchange && !interleaved) {
            dst = m->d_frame + (size_t)begin * w * 3;          // contiguous: the root's band is rendered in place
        } else {
            if (band_fl > r.band_cap) {
                if (r.d_band) (void)hipFree(r.d_band);
                r.d_band = nullptr; r.band_cap = 0;
                RK_HIP(r, hipMalloc(reinterpret_cast<void**>(&r.d_band), band_fl * sizeof(float)));
                r.band_cap = band_fl;
            }
            dst = r.d_band;
        }
        r.stats = spt_stats{};
        r.gather_ms = 0.f;
        if (count) {
            const int e = interleaved
                ? spt_render_interleaved_device(r.ctx, cam, w, h, kBlockRows, world, (uint32_t)i, samps, seed, flags, dst, r.stream)
                : spt_render_rows_device(r.ctx, cam, w, h, begin, count, samps, seed, flags, dst, r.stream);
            if (e) { r.error = spt_last_error(r.ctx); return; }
        }
        RK_HIP(r, hipStreamSynchronize(r.stream));
        if (count && spt_sync(r.ctx, &r.stats)) { r.error = spt_last_error(r.ctx); return; }
    });
    if (rc) return rc;                                         // nothing of the exchange has been enqueued: no peer is left waiting

    // 2. the exchange step (RCCL): every rank's rows are complete
    if (m->use_rccl) {
        if (m->init_comms()) return 1;                        // (new communicators after an exchange that was aborted)
        rc = m->on_all([=, &stage_off](int i) {
            Rank& r = m->ranks[(size_t)i];
            const uint32_t count = rows_of((uint32_t)i, nullptr);
            const size_t band_fl = (size_t)count * w * 3;
            XK_HIP(m, r, hipSetDevice(r.device));
            if (m->fail_exchange_rank == i) { m->fail_exchange_rank = -1; r.error = "injected exchange failure (test hook)"; m->abort_comms(); return; }
            XK_HIP(m, r, hipEventRecord(r.ev_a, r.stream));
            if (i == 0) {
                XK_NCCL(m, r, ncclGroupStart());
                for (uint32_t p = self_exchange ? 0u : 1u; p < world; ++p) {
                    uint32_t pb = 0;
                    const uint32_t pc = rows_of(p, &pb);
                    float* to = interleaved ? m->d_staging + stage_off[p] : m->d_frame + (size_t)pb * w * 3;
                    if (pc) XK_NCCL(m, r, ncclRecv(to, (size_t)pc * w * 3, ncclFloat, (int)p, r.comm, r.stream));
                }
                if (self_exchange && count) XK_NCCL(m, r, ncclSend(r.d_band, band_fl, ncclFloat, 0, r.comm, r.stream));
                XK_NCCL(m, r, ncclGroupEnd());
                if (interleaved) {
                    // scatter every rank's packed row blocks to their rows: block k of rank p starts at row (k * world + p) * B
                    for (uint32_t p = 0; p < world; ++p) {
                        const uint32_t pc = rows_of(p, nullptr);
                        if (!pc) continue;
                        const float* from = p == 0 ? r.d_band : m->d_staging + stage_off[p];
                        const size_t block_bytes = (size_t)kBlockRows * w * 3 * sizeof(float);
                        const uint32_t full = pc / kBlockRows, rest = pc % kBlockRows;
                        if (full)
                            XK_HIP(m, r, hipMemcpy2DAsync(m->d_frame + (size_t)p * kBlockRows * w * 3, block_bytes * world, from, block_bytes,
                                                       block_bytes, full, hipMemcpyDeviceToDevice, r.stream));
                        if (rest)
                            XK_HIP(m, r, hipMemcpyAsync(m->d_frame + ((size_t)full * world + p) * kBlockRows * w * 3, from + (size_t)full * kBlockRows * w * 3,
                                                     (size_t)rest * w * 3 * sizeof(float), hipMemcpyDeviceToDevice, r.stream));
                    }
                }
            } else if (count) {
                XK_NCCL(m, r, ncclSend(r.d_band, band_fl, ncclFloat, 0, r.comm, r.stream));
            }
            XK_HIP(m, r, hipEventRecord(r.ev_b, r.stream));
            XK_HIP(m, r, hipStreamSynchronize(r.stream));
            XK_HIP(m, r, hipEventElapsedTime(&r.gather_ms, r.ev_a, r.ev_b));
        });
        if (rc) { m->abort_comms(); return rc; }              // (a rank that failed has aborted them already; idempotent)
    }

    // 2'. copy transport (SPT_MULTI_COPY_EXCHANGE): every rank has finished its rows (stream synchronised above); the root
    // pulls the packed rows with peer copies and scatters / places them exactly like the RCCL path does.
    if (!m->use_rccl && world > 1) {
        Rank& r0 = m->ranks[0];
        r0.error.clear();
        r0.worker->submit([&] {
            Rank& r = r0;
            RK_HIP(r, hipSetDevice(r.device));
            RK_HIP(r, hipEventRecord(r.ev_a, r.stream));
            for (uint32_t p = 0; p < world; ++p) {
                uint32_t pb = 0;
                const uint32_t pc = rows_of(p, &pb);
                if (!pc) continue;
                const Rank& src = m->ranks[p];
                if (!interleaved) {
                    if (p == 0) continue;                      // rendered in place
                    RK_HIP(r, hipMemcpyPeerAsync(m->d_frame + (size_t)pb * w * 3, r.device, src.d_band, src.device, (size_t)pc * w * 3 * sizeof(float), r.stream));
                    continue;
                }
                const float* from = src.d_band;
                if (p != 0) {
                    RK_HIP(r, hipMemcpyPeerAsync(m->d_staging + stage_off[p], r.device, src.d_band, src.device, (size_t)pc * w * 3 * sizeof(float), r.stream));
                    from = m->d_staging + stage_off[p];
                }
                const size_t block_bytes = (size_t)kBlockRows * w * 3 * sizeof(float);
                const uint32_t full = pc / kBlockRows, rest = pc % kBlockRows;
                if (full)
                    RK_HIP(r, hipMemcpy2DAsync(m->d_frame + (size_t)p * kBlockRows * w * 3, block_bytes * world, from, block_bytes,
                                               block_bytes, full, hipMemcpyDeviceToDevice, r.stream));
                if (rest)
                    RK_HIP(r, hipMemcpyAsync(m->d_frame + ((size_t)full * world + p) * kBlockRows * w * 3, from + (size_t)full * kBlockRows * w * 3,
                                             (size_t)rest * w * 3 * sizeof(float), hipMemcpyDeviceToDevice, r.stream));
            }
            RK_HIP(r, hipEventRecord(r.ev_b, r.stream));
            RK_HIP(r, hipStreamSynchronize(r.stream));
            RK_HIP(r, hipEventElapsedTime(&r.gather_ms, r.ev_a, r.ev_b));
        });
        r0.worker->wait();
        if (!r0.error.empty()) return m->fail("device %d: %s", r0.device, r0.error.c_str());
    }

    // 3. framebuffer to the host if asked for
    if (out_rgb) {
        rc = 0;
        Rank& r0 = m->ranks[0];
        r0.error.clear();
        r0.worker->submit([&] {
            RK_HIP(r0, hipSetDevice(r0.device));
            RK_HIP(r0, hipMemcpy(out_rgb, m->d_frame, nfl * sizeof(float), hipMemcpyDeviceToHost));
        });
        r0.worker->wait();
        if (!r0.error.empty()) return m->fail("device %d: %s", r0.device, r0.error.c_str());
    }
    if (stats) {
        *stats = spt_multi_stats{};
        stats->ndev = world;
        for (const Rank& r : m->ranks) {
            stats->samples += r.stats.samples;
            stats->bounces += r.stats.bounces;
            stats->max_depth_kills += r.stats.max_depth_kills;
            const float ms = r.stats.kernel_ms + r.stats.finalize_ms;
            if (ms > stats->render_ms) stats->render_ms = ms;
        }
        stats->gather_ms = m->ranks[0].gather_ms;
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
    } catch (const std::exception& e) {
        return m->fail("spt_multi_render: %s", e.what());
    }
}

// ---- the viewer's render loop over several devices (smallpt.cpp:895-942): accumBuffer lives on the root device ----
static int on_root(spt_multi* m, const std::function<void(Rank&)>& fn)
{
    Rank& r0 = m->ranks[0];
    r0.error.clear();
    r0.worker->submit([&] { fn(r0); });
    r0.worker->wait();
    if (!r0.error.empty()) return m->fail("device %d: %s", r0.device, r0.error.c_str());
    return 0;
}

int spt_multi_progressive_begin(spt_multi* m, uint32_t w, uint32_t h)
{
    if (!m) return 1;
    if (w == 0 || h == 0) return m->fail("spt_multi_progressive_begin: empty image");
    const size_t bytes = (size_t)w * h * 3 * sizeof(float);
    const int rc = on_root(m, [&](Rank& r) {
        RK_HIP(r, hipSetDevice(r.device));
        if (m->d_accum) (void)hipFree(m->d_accum);
        m->d_accum = nullptr;
        RK_HIP(r, hipMalloc(reinterpret_cast<void**>(&m->d_accum), bytes));
        RK_HIP(r, hipMemset(m->d_accum, 0, bytes));
    });
    if (rc == 0) { m->prog_w = w; m->prog_h = h; }
    return rc;
}

int spt_multi_progressive_frame(spt_multi* m, const spt_camera* cam, uint32_t samps, uint64_t seed, int clear, spt_multi_stats* stats)
{
    if (!m) return 1;
    if (!m->d_accum) return m->fail("spt_multi_progressive_frame: call spt_multi_progressive_begin first");
    // outImage = renderer.render(...) (:922) on all devices, assembled on the root; then accumBuffer (+)= outImage (:924-937) there
    int rc = spt_multi_render(m, cam, m->prog_w, m->prog_h, samps, seed, 0u, nullptr, stats);
    if (rc != 0) return rc;
    const uint64_t n = (uint64_t)m->prog_w * m->prog_h * 3;
    return on_root(m, [&](Rank& r) {
        RK_HIP(r, hipSetDevice(r.device));
        if (spt_accumulate_device(r.ctx, m->d_accum, m->d_frame, n, clear, r.stream) != 0) { r.error = spt_last_error(r.ctx); return; }
        RK_HIP(r, hipStreamSynchronize(r.stream));
    });
}

int spt_multi_progressive_snapshot(spt_multi* m, float* out_rgb)
{
    if (!m) return 1;
    if (!m->d_accum) return m->fail("spt_multi_progressive_snapshot: call spt_multi_progressive_begin first");
    if (!out_rgb) return m->fail("spt_multi_progressive_snapshot: out_rgb is NULL");
    const size_t bytes = (size_t)m->prog_w * m->prog_h * 3 * sizeof(float);
    return on_root(m, [&](Rank& r) {
        RK_HIP(r, hipSetDevice(r.device));
        RK_HIP(r, hipMemcpy(out_rgb, m->d_accum, bytes, hipMemcpyDeviceToHost));
    });
}

int spt_multi_progressive_end(spt_multi* m)
{
    if (!m) return 1;
    const int rc = on_root(m, [&](Rank& r) {
        RK_HIP(r, hipSetDevice(r.device));
        if (m->d_accum) RK_HIP(r, hipFree(m->d_accum));
        m->d_accum = nullptr;
    });
    m->prog_w = m->prog_h = 0;
    return rc;
}

}  // extern "C"
