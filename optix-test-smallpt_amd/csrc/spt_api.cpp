// spt_api.cpp -- the C-ABI of include/smallpt_mi355x.h on top of the gfx950 megakernel.
// Host-side only: scene upload, launch geometry, HIP-event timing, statistics, image output.
#include "../../include/smallpt_mi355x.h"
#include "spt_internal.h"
#include "spt_bvh.h"
#include "spt_grid.h"
#include "spt_kernel.h"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct HostF3 { float x, y, z; };
inline HostF3 hscl(HostF3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float hdot(HostF3 a, HostF3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline HostF3 hcross(HostF3 a, HostF3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline HostF3 hnormalize(HostF3 v) { float inv = 1.0f / std::sqrt(hdot(v, v)); return hscl(v, inv); }

// D7 seed hashing (host side; the per-pixel / per-sample part runs in the kernel)
inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}

}  // namespace

struct spt_ctx {
    int device = 0;
    int cu_count = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_stop = nullptr;
    // scene
    uint32_t n = 0;
    float4* d_geom = nullptr;
    float4* d_mat = nullptr;
    uint32_t scene_cap = 0;
    bool needs_guard = false;      // r*r < 2^-60 or coordinates above 1e15: the hot-loop sqrt keeps its range guard
    bool pool_ok = false;          // scene qualifies for the material-sorted pool kernel (spt_pool.hip)
    float* d_stack = nullptr;      // pool kernel: global-memory stack of pending transmitted children
    size_t stack_cap = 0;          // in floats
    bool last_was_pool = false;
    int last_kernel = 0;           // 0 megakernel, 1 pool kernel, 2 mesh kernel (triangles; 6 / 7: through the exact / the plain hierarchy), 3 mesh kernel over a sphere hierarchy, 4 grid kernel (lanes own paths), 5 grid kernel with path pools
    // triangle-mesh scene (spt_set_meshes); mesh_scene selects it for spt_render*
    bool mesh_scene = false;
    bool mesh_specular = false;            // a mesh material is SPEC or REFR: long mirror / glass chains are possible (task dealing of the hierarchy kernel)
    float4* d_tris = nullptr; uint4* d_tri_index = nullptr; float4* d_verts = nullptr; uint32_t* d_inst_first = nullptr; float4* d_mesh_mats = nullptr;
    float* d_trace_rays = nullptr; float* d_trace_hits = nullptr; uint64_t trace_cap = 0;   // spt_trace_rays staging (rays)
    std::vector<float4> h_geom;      // host copy of the sphere table {centre, r*r} and the radii: its hierarchy is built on demand
    std::vector<float> h_radius;
    int sphere_accel = SPT_ACCEL_GRID;
    // uniform grid over the sphere table (spt_grid.h): built in spt_set_scene for tables above the pool kernel's limit
    bool grid_ready = false;         // the tables below belong to the current sphere scene and the scene qualifies
    int grid_global = 0;             // ... 1: every table stays in global memory, 2: the sphere records do, the grid is staged in LDS (spt_grid.hip WHERE)
    spt::GridParams grid{};
    uint32_t* d_grid_cells = nullptr; uint16_t* d_grid_refs = nullptr; uint32_t* d_grid_always = nullptr;
    std::string grid_why;            // why the current scene does not run on the grid kernel
    bool sbvh_ready = false;
    float4* d_sbvh_nodes = nullptr; float4* d_sbvh_geom = nullptr; uint32_t* d_sbvh_index = nullptr; uint32_t* d_sbvh_always = nullptr;
    uint32_t sbvh_nalways = 0, sbvh_depth = 0;
    std::vector<float4> h_tris;      // host copy of the triangle records: the hierarchy is built from it on demand
    int accel = SPT_ACCEL_AUTO;            // mesh scenes (spt_set_mesh_accel): one of the two modes that return the exhaustive loop's Hit for every ray (spt_tribvh.h)
    float mesh_ratio = -1.f;               // closest-hit queries per sample of the last synchronised launch of this mesh scene (-1: none yet): SPT_ACCEL_AUTO
    int last_mesh_mode = SPT_ACCEL_EXHAUSTIVE;   // what the last mesh launch / query ran through
    bool bvh_ready = false;          // the hierarchy below belongs to the current mesh scene
    float4* d_bvh_nodes = nullptr; float4* d_bvh_tris = nullptr; uint32_t* d_bvh_index = nullptr;
    float4* d_flat_lines = nullptr; uint32_t* d_flat_line_index = nullptr; uint32_t nline_slots = 0; bool bvh_flat = false;     // thin triangles as a table (spt_tribvh.h (3))
    uint32_t* d_cam_planes = nullptr; uint32_t ncam = 0, cam_cap = 0; float cam_key[4] = {0, 0, 0, 0}; bool cam_valid = false;   // spt_bvh.h camera_planes of the last pinhole origin
    float4* d_bvh_cones = nullptr; float4* d_plane_nodes = nullptr; float4* d_line_nodes = nullptr; bool have_planes = false, have_lines = false;   // spt_tribvh.h
    uint32_t bvh_nodes = 0, bvh_depth = 0, bvh_leaves = 0;
    uint32_t ntris = 0, ninst = 0;
    float* d_accum = nullptr;      // spt_progressive_*: accumBuffer (smallpt.cpp:881-883) and the current frame, w*h*3 floats each
    float* d_frame = nullptr;
    uint32_t prog_w = 0, prog_h = 0;
    hipEvent_t ev_acc = nullptr;   // owner of an accumBuffer: completion of the most recent accumulation (any lane's stream)
    bool acc_recorded = false;
    bool frame_in_flight = false;  // a spt_progressive_frame_async of this lane has not been waited for
    uint32_t lanes_attached = 0;   // owner: lanes attached right now (spreads them over the stream priorities, sizes short launches)
    spt_ctx* attached_to = nullptr; // lane: the owner it is attached to (its count is given back when the lane ends or re-attaches)
    uint32_t frames_in_flight_hint = 1;   // set by spt_progressive_frame_async for its launch: lanes of the loop (sizes a short launch's grid)
    unsigned long long pool_stats[24] = {};  // batches per class [3], lanes per class [3], watchdog hits, tail batches, tail lanes, full batches
    // scratch
    float4* d_cells = nullptr;
    size_t cells_cap = 0;          // in float4
    float* d_out = nullptr;        // image buffer for spt_render
    size_t out_cap = 0;            // in floats
    uint32_t* d_queue = nullptr;   // 1 x u32 queue head + 2 x u64 counters (one 32-byte allocation)
    unsigned long long* d_counters = nullptr;
    // tuning
    uint32_t blocks_per_cu = 0;
    uint32_t variant = 0;
    // grid kernels: 0 = wave-private path pools (spt_gpool.hip) whenever the LDS has room for them, 1 = lanes own their path (spt_grid.hip);
    // pool geometry {slots per wave, begun walks per wave, drain, smallest batch, walk iterations behind a batch's loads} (spt_set_grid_pools)
    int grid_lane_owned = 0;
    int grid_force_global = 0;
    uint32_t gq[5] = {192u, 96u, 24u, 32u, 4u};
    // pool kernel, cost-ordered dispatch (spt_kernel.h KParams::chunk_order): tables of the last pool launch and the view they belong to
    uint32_t* d_chunk_tables = nullptr;   // order[cap] | clock[2 * cap] | 512 words of the sorting kernels
    size_t chunk_cap = 0;
    bool order_valid = false;
    std::vector<unsigned char> order_key;  // camera, image, band, samples, scene generation, seed: an identical next launch reuses the order
    std::vector<unsigned char> last_pool_key;   // ... of the last pool launch, recorded or not: a launch records only when it repeats its predecessor
    uint64_t scene_gen = 0;
    uint32_t last_nchunks = 0;             // chunks of the launch the order table was derived from (0: that launch recorded none)
    hipEvent_t ev_order = nullptr;         // the order kernel of the last pool launch has run (the next launch may come on another stream)
    bool order_pending = false;
    unsigned long long watchdog_ticks = 0;   // pool kernel: s_memtime ticks (shader cycles) per launch; 0 = no watchdog
    // last launch
    bool pending = false;
    spt_stats last{};
    unsigned long long diag[24] = {};   // DIAG build only: phase wave-times and lane counts (pool kernel: its statistics)
    std::string error;

    int fail(const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        return 1;
    }
};

#define SPT_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) return (ctx)->fail("%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

extern "C" {

int spt_api_version(void) { return SPT_API_VERSION; }

int spt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* spt_last_error(const spt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int spt_create(int device_id, spt_ctx** out)
{
    if (!out) { g_create_error = "spt_create: out is NULL"; return 1; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = std::string("spt_create: no HIP device available (") + hipGetErrorString(e) +
                         "); this library has no CPU fallback";
        return 1;
    }
    if (device_id < 0 || device_id >= ndev) { g_create_error = "spt_create: device_id out of range"; return 1; }
    spt_ctx* c = new spt_ctx;
    c->device = device_id;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string("spt_create: ") + what + ": " + hipGetErrorString(err);
        spt_destroy(c);
        return 1;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("spt_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        spt_destroy(c);
        return 1;
    }
    c->cu_count = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev_start)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev_mid)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev_stop)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    void* p = nullptr;
    if ((e = hipMalloc(&p, 256)) != hipSuccess) return bail("hipMalloc", e);
    c->d_queue = static_cast<uint32_t*>(p);
    c->d_counters = reinterpret_cast<unsigned long long*>(static_cast<char*>(p) + 16);
    *out = c;
    return 0;
}

void spt_destroy(spt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_chunk_tables) (void)hipFree(c->d_chunk_tables);
    if (c->d_geom) (void)hipFree(c->d_geom);
    if (c->d_mat) (void)hipFree(c->d_mat);
    if (c->d_cells) (void)hipFree(c->d_cells);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_stack) (void)hipFree(c->d_stack);
    if (c->d_accum) (void)hipFree(c->d_accum);
    if (c->d_frame) (void)hipFree(c->d_frame);
    if (c->d_tris) (void)hipFree(c->d_tris);
    if (c->d_tri_index) (void)hipFree(c->d_tri_index);
    if (c->d_trace_rays) (void)hipFree(c->d_trace_rays);
    if (c->d_trace_hits) (void)hipFree(c->d_trace_hits);
    if (c->d_grid_cells) (void)hipFree(c->d_grid_cells);
    if (c->d_grid_refs) (void)hipFree(c->d_grid_refs);
    if (c->d_grid_always) (void)hipFree(c->d_grid_always);
    if (c->d_sbvh_nodes) (void)hipFree(c->d_sbvh_nodes);
    if (c->d_sbvh_geom) (void)hipFree(c->d_sbvh_geom);
    if (c->d_sbvh_index) (void)hipFree(c->d_sbvh_index);
    if (c->d_sbvh_always) (void)hipFree(c->d_sbvh_always);
    if (c->d_bvh_nodes) (void)hipFree(c->d_bvh_nodes);
    if (c->d_bvh_tris) (void)hipFree(c->d_bvh_tris);
    if (c->d_bvh_index) (void)hipFree(c->d_bvh_index);
    if (c->d_bvh_cones) (void)hipFree(c->d_bvh_cones);
    if (c->d_cam_planes) (void)hipFree(c->d_cam_planes);
    if (c->d_flat_line_index) (void)hipFree(c->d_flat_line_index);
    if (c->d_flat_lines) (void)hipFree(c->d_flat_lines);
    if (c->d_plane_nodes) (void)hipFree(c->d_plane_nodes);
    if (c->d_line_nodes) (void)hipFree(c->d_line_nodes);
    if (c->d_verts) (void)hipFree(c->d_verts);
    if (c->d_inst_first) (void)hipFree(c->d_inst_first);
    if (c->d_mesh_mats) (void)hipFree(c->d_mesh_mats);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->ev_acc) (void)hipEventDestroy(c->ev_acc);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int spt_set_grid_pools(spt_ctx* c, int lane_owned, uint32_t slots, uint32_t ready, uint32_t drain, uint32_t min_batch, uint32_t walk_iters)
{
    if (!c) return 1;
    if (slots > 256u || (slots & 15u) || ready > 0xFFFFu || drain > 64u)
        return c->fail("spt_set_grid_pools: slots must be a multiple of 16 up to 256, ready <= 65535, drain <= 64");
    c->grid_lane_owned = lane_owned ? 1 : 0;
    c->grid_force_global = lane_owned >= 2 ? lane_owned - 1 : 0;   // 2: every table in global memory, 3: the sphere records only                      // (A/B: tables in global memory although they would fit the LDS; read by the next spt_set_scene)
    const uint32_t def[5] = {192u, 96u, 24u, 32u, 4u}, in[5] = {slots, ready & ~3u, drain, min_batch, walk_iters};
    for (int i = 0; i < 5; ++i) c->gq[i] = in[i] ? in[i] : def[i];
    return 0;
}

int spt_set_tuning(spt_ctx* c, uint32_t blocks_per_cu, uint32_t variant)
{
    if (!c) return 1;
    const uint32_t psel = (variant >> 11) & 3u;                   // pool slots per wave: 1 -> 96 and 2 -> 192 exist in -DSPT_POOL_SIZES builds only
    const int pool = psel == 1 ? 96 : (psel == 2 ? 192 : (psel == 3 ? 128 : spt_pool_default_slots()));
    if (!spt_pool_has_size(pool)) return c->fail("spt_set_tuning: this build carries no pool kernel with %d slots per wave (bits 12:11 = %u)", pool, psel);
    c->blocks_per_cu = blocks_per_cu;
    c->variant = variant;
    return 0;
}

// Builds the device tables from the reference-shaped sphere records.  All derived values are single
// IEEE operations on the host, bit-identical to evaluating them per bounce:
//   r*r (scene.cpp:133), pmax = fmaxf(color) (smallpt.cpp:177), color*(1/pmax) (smallpt.cpp:192).
static int set_scene_impl(spt_ctx* c, const spt_sphere* s, uint32_t n);
static int build_sphere_accel(spt_ctx* c);
static int build_default_sphere_structure(spt_ctx* c);
static int build_sphere_grid_tables(spt_ctx* c);

int spt_set_scene(spt_ctx* c, const spt_sphere* s, uint32_t n)
{
    if (!c) return 1;
    try {
        return set_scene_impl(c, s, n);
    } catch (const std::exception& e) {
        return c->fail("spt_set_scene: %s", e.what());
    }
}

static int set_scene_impl(spt_ctx* c, const spt_sphere* s, uint32_t n)
{
    if (n > SPT_MAX_SPHERES_ACCEL) return c->fail("spt_set_scene: %u spheres > SPT_MAX_SPHERES_ACCEL (%u)", n, SPT_MAX_SPHERES_ACCEL);
    if (n && !s) return c->fail("spt_set_scene: spheres is NULL");
    for (uint32_t i = 0; i < n; ++i)
        if (s[i].refl < SPT_DIFF || s[i].refl > SPT_REFR) return c->fail("spt_set_scene: sphere %u has refl=%d", i, s[i].refl);
    // The un-guarded square root (sqrt_rsq) in the closest-hit loop is exact for det = 0 or 2^-96 <= det < inf.  That holds
    // whenever r*r >= 2^-60 and no coordinate can overflow b*b / dot(op,op); other scenes get the guarded build.
    bool needs_guard = false;
    for (uint32_t i = 0; i < n; ++i) {
        const float big = std::fmax(std::fmax(std::fabs(s[i].center[0]), std::fabs(s[i].center[1])),
                                    std::fmax(std::fabs(s[i].center[2]), std::fabs(s[i].radius)));
        if (!(s[i].radius * s[i].radius >= 0x1p-60f) || !(big <= 1e15f)) needs_guard = true;
    }
    // The exhaustive kernels stage the whole table in LDS (SPT_MAX_SPHERES); a larger table exists only behind a structure -- the
    // grid while its tables fit one CU's LDS, the hierarchy (records in global memory) beyond -- whose error bounds exclude the
    // degenerate scenes of the guarded build.
    if (n > SPT_MAX_SPHERES) {
        if (c->sphere_accel == SPT_ACCEL_EXHAUSTIVE)
            return c->fail("spt_set_scene: %u spheres > SPT_MAX_SPHERES (%u): the exhaustive kernels stage the table in LDS; larger tables need SPT_ACCEL_GRID or SPT_ACCEL_BVH", n, SPT_MAX_SPHERES);
        if (needs_guard)
            return c->fail("spt_set_scene: %u spheres > SPT_MAX_SPHERES (%u) with radii below 2^-30 or coordinates beyond 1e15, which only the exhaustive kernels take", n, SPT_MAX_SPHERES);
    }
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) SPT_HIP(c, hipEventSynchronize(c->ev_stop));
    const uint32_t cap = n ? n : 1;
    if (cap > c->scene_cap) {
        if (c->d_geom) (void)hipFree(c->d_geom);
        if (c->d_mat) (void)hipFree(c->d_mat);
        c->d_geom = c->d_mat = nullptr;
        c->scene_cap = 0;
        SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_geom), sizeof(float4) * cap));
        SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_mat), sizeof(float4) * 3 * cap));
        c->scene_cap = cap;
    }
    std::vector<float4> geom(cap), mat(3 * (size_t)cap);
    for (uint32_t i = 0; i < n; ++i) {
        const spt_sphere& sp = s[i];
        geom[i] = make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius * sp.radius);
        const float pmax = std::fmax(std::fmax(sp.color[0], sp.color[1]), sp.color[2]);
        const float inv = 1.0f / pmax;
        // refl in bits 1:0; bit 2 = "emission is not exactly zero" (the pool kernel skips the + weight*0 of smallpt.cpp:179
        // for non-emissive hits, which is exact for finite weights)
        const bool emissive = !(sp.emission[0] == 0.f && sp.emission[1] == 0.f && sp.emission[2] == 0.f);
        const int32_t rb = sp.refl | (emissive ? 4 : 0);
        float reflbits;
        std::memcpy(&reflbits, &rb, 4);
        mat[3 * i + 0] = make_float4(sp.emission[0], sp.emission[1], sp.emission[2], reflbits);
        mat[3 * i + 1] = make_float4(sp.color[0], sp.color[1], sp.color[2], pmax);
        mat[3 * i + 2] = make_float4(sp.color[0] * inv, sp.color[1] * inv, sp.color[2] * inv, 0.0f);
    }
    SPT_HIP(c, hipMemcpy(c->d_geom, geom.data(), sizeof(float4) * cap, hipMemcpyHostToDevice));
    SPT_HIP(c, hipMemcpy(c->d_mat, mat.data(), sizeof(float4) * 3 * cap, hipMemcpyHostToDevice));
    c->n = n;
    ++c->scene_gen;
    c->mesh_scene = false;
    c->h_geom.assign(geom.begin(), geom.begin() + n);
    c->h_radius.resize(n);
    for (uint32_t i = 0; i < n; ++i) c->h_radius[i] = s[i].radius;
    c->sbvh_ready = false;
    c->grid_ready = false;
    c->needs_guard = needs_guard;
    // Pool kernel: unrolled closest hit (<= 24 spheres), un-guarded sqrt, and path weights that only the glass factors
    // can push out of the finite range (colours in [0,1], finite emission) -- it tracks that case with a flag.
    c->pool_ok = n <= (uint32_t)spt_pool_max_spheres() && !c->needs_guard;
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            if (!(s[i].color[k] >= 0.f && s[i].color[k] <= 1.f) || !(std::fabs(s[i].emission[k]) <= 3e38f)) c->pool_ok = false;
    if (c->sphere_accel == SPT_ACCEL_BVH) return build_sphere_accel(c);
    return c->sphere_accel == SPT_ACCEL_GRID ? build_default_sphere_structure(c) : 0;   // (n > SPT_MAX_SPHERES: always one of the two)
}

// LDS the grid kernel may spend on cell headers + references + the always-list, beside the 16-byte sphere records (one workgroup per CU)
static size_t grid_table_budget(uint32_t n) { return (size_t)150 * 1024 - (size_t)(n ? n : 1u) * 16u; }

// Uniform grid over the current sphere table (spt_grid.h); the caller holds the C-boundary try block.  Tables the pool kernel
// takes, scenes that need the range-guarded square root and tables that do not fit the LDS keep the other kernels (grid_why says which).
static int build_sphere_grid_tables(spt_ctx* c)
{
    c->grid_ready = false;
    if (c->n <= (uint32_t)spt_pool_max_spheres()) { c->grid_why = "table small enough for the unrolled closest hit"; return 0; }
    if (c->needs_guard) { c->grid_why = "scene needs the range-guarded square root"; return 0; }
    // Tables that fit one CU's LDS are staged there (both grid kernels).  Larger ones -- sphere records or grid beyond the LDS -- keep the
    // grid with their tables in global memory (spt_grid.hip GLOBAL_TABLES, round 4) up to kGridGlobalMax spheres: every lookup of the walk
    // is then a 64-lane gather through the texture path instead of an LDS read, which the walk's ~40 dependent lookups per ray pay for
    // (16 384 random spheres: 404 Msamples/s against 230 through the hierarchy; at 24 576 the two are level, at 32 768 the hierarchy's
    // log N wins 326 : 236, profiles/r04_big_tables.txt), so beyond that the hierarchy keeps the scene as in round 3.
    constexpr uint32_t kGridGlobalMax = 24576;
    spt::SphereGrid g;
    const uint32_t dsel = (c->variant >> 24) & 0xFFu;
    const bool records_fit = (size_t)c->n * 16u + 8192u <= (size_t)150 * 1024;
    c->grid_global = 0;
    if (records_fit && c->grid_force_global == 0) spt::build_sphere_grid(c->h_geom.data(), c->h_radius.data(), c->n, dsel ? (double)dsel : 4.0, grid_table_budget(c->n), g);
    // (an LDS grid that had to shrink below a quarter of a cell per sphere to fit -- from about 6 500 random spheres on -- tests too many
    // spheres per cell: 8 000 spheres, 4 x 4 x 7 cells: 336 Msamples/s from LDS against 596 from global memory at the full resolution;
    // 6 000 spheres, 0.36 cells per sphere: 888 against 666; profiles/r04_big_tables.txt)
    if (g.usable && c->n <= kGridGlobalMax) {
        const double interior = (double)g.P.dim[0] * g.P.dim[1] * g.P.dim[2], in_grid_n = (double)c->n - (double)g.always.size();
        if (interior < 0.25 * in_grid_n) g = spt::SphereGrid();
    }
    if (!g.usable) {
        const std::string lds_why = g.why;
        if (c->n > 0xFFFFu) { c->grid_why = "more spheres than the grid's 16-bit references address"; return 0; }
        // the sphere records in global memory and the grid in LDS (two of the walk's three lookups per sphere stay LDS reads), if a grid of
        // at least a quarter of a cell per sphere fits there; else everything in global memory at the full resolution, up to kGridGlobalMax
        g = spt::SphereGrid();
        if (c->grid_force_global != 1) spt::build_sphere_grid(c->h_geom.data(), c->h_radius.data(), c->n, dsel ? (double)dsel : 4.0, (size_t)150 * 1024, g);
        c->grid_global = 2;
        const double interior = g.usable ? (double)g.P.dim[0] * g.P.dim[1] * g.P.dim[2] : 0.0;
        if (!g.usable || interior < 0.25 * ((double)c->n - (double)g.always.size())) {
            if (c->n > kGridGlobalMax) { c->grid_why = records_fit ? lds_why : "sphere records alone exceed the LDS, and the table is beyond the size up to which the global-memory grid beats the hierarchy"; return 0; }
            g = spt::SphereGrid();
            spt::build_sphere_grid(c->h_geom.data(), c->h_radius.data(), c->n, dsel ? (double)dsel : 4.0, (size_t)256 << 20, g);
            c->grid_global = 1;
        }
    }
    if (!g.usable) { c->grid_why = g.why; return 0; }
    // A cell that lists a third of the table means nearly everything shares a cell (the extent is set by a few large spheres that
    // are not large enough for the always-tested list): the walk would test the whole table per lane with LDS gathers, slower than
    // the exhaustive kernel's broadcast loop (measured 2.5x on such a table), which then keeps the scene.
    const size_t in_grid = (size_t)c->n - g.always.size();
    if (in_grid > 96 && (size_t)g.max_cell * 3 > in_grid) { c->grid_why = "a single cell lists more than a third of the spheres"; return 0; }
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) SPT_HIP(c, hipEventSynchronize(c->ev_stop));
    auto upload = [&](auto*& dptr, const void* src, size_t bytes) -> hipError_t {
        if (dptr) (void)hipFree(dptr);
        dptr = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dptr), bytes ? bytes : 16);
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice);
    };
    SPT_HIP(c, upload(c->d_grid_cells, g.cells.data(), g.cells.size() * sizeof(uint32_t)));
    SPT_HIP(c, upload(c->d_grid_refs, g.refs.data(), g.refs.size() * sizeof(uint16_t)));
    SPT_HIP(c, upload(c->d_grid_always, g.always.data(), g.always.size() * sizeof(uint32_t)));
    c->grid = g.P;
    c->grid_why.clear();
    c->grid_ready = true;
    return 0;
}

// SPT_ACCEL_GRID on a table the grid does not take (beyond the LDS, everything in one cell; not: degenerate radii / coordinates): from
// kSphereBvhFrom spheres on the hierarchy -- exhaustive-equivalent as well -- takes the scene instead of the exhaustive kernel
// (measured exhaustive / hierarchy: 1024 spheres 375 / 483, 4096 spheres 62 / 315 Msamples/s; below ~1000 the exhaustive kernel wins).
constexpr uint32_t kSphereBvhFrom = 1024;
static int build_default_sphere_structure(spt_ctx* c)
{
    const int rc = build_sphere_grid_tables(c);
    if (rc != 0 || c->grid_ready || c->n < kSphereBvhFrom || c->needs_guard) return rc;
    return c->sbvh_ready ? 0 : build_sphere_accel(c);
}

// Hierarchy over the current sphere table (spt_bvh.h build_sphere_bvh); the caller holds the C-boundary try block.
static int build_sphere_accel(spt_ctx* c)
{
    spt::Bvh bvh;
    spt::build_sphere_bvh(c->h_geom.data(), c->h_radius.data(), c->n, bvh);
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) SPT_HIP(c, hipEventSynchronize(c->ev_stop));
    auto upload = [&](auto*& dptr, const void* src, size_t bytes) -> hipError_t {
        if (dptr) (void)hipFree(dptr);
        dptr = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dptr), bytes ? bytes : 16);
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice);
    };
    SPT_HIP(c, upload(c->d_sbvh_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(float4)));
    SPT_HIP(c, upload(c->d_sbvh_geom, bvh.tris.data(), bvh.tris.size() * sizeof(float4)));
    SPT_HIP(c, upload(c->d_sbvh_index, bvh.index.data(), bvh.index.size() * sizeof(uint32_t)));
    SPT_HIP(c, upload(c->d_sbvh_always, bvh.always.data(), bvh.always.size() * sizeof(uint32_t)));
    c->sbvh_nalways = (uint32_t)bvh.always.size(); c->sbvh_depth = bvh.depth;
    c->sbvh_ready = true;
    return 0;
}

int spt_set_sphere_accel(spt_ctx* c, int accel)
{
    if (!c) return 1;
    if (accel != SPT_ACCEL_EXHAUSTIVE && accel != SPT_ACCEL_BVH && accel != SPT_ACCEL_GRID) return c->fail("spt_set_sphere_accel: unknown mode %d", accel);
    if (accel == SPT_ACCEL_EXHAUSTIVE && !c->mesh_scene && c->d_geom && c->n > SPT_MAX_SPHERES)
        return c->fail("spt_set_sphere_accel: the current table has %u spheres > SPT_MAX_SPHERES (%u), which the exhaustive kernels cannot stage in LDS", c->n, SPT_MAX_SPHERES);
    c->sphere_accel = accel;
    if (accel == SPT_ACCEL_EXHAUSTIVE || c->mesh_scene || !c->d_geom) return 0;
    try {
        if (accel == SPT_ACCEL_BVH) return c->sbvh_ready ? 0 : build_sphere_accel(c);
        return c->grid_ready ? 0 : build_default_sphere_structure(c);
    } catch (const std::exception& e) {
        return c->fail("spt_set_sphere_accel: %s", e.what());
    }
}

// Host-only self-test of the grid builder (no device call).  out8 = {dim x, dim y, dim z, references, always-tested spheres, table bytes, usable, most references in one cell}.
int spt_selftest_sphere_grid(const spt_sphere* s, uint32_t n, uint32_t cells_per_sphere, uint32_t* out8, char* why, uint32_t why_len)
{
    try {
        std::vector<float4> geom(n);
        std::vector<float> radius(n);
        for (uint32_t i = 0; i < n; ++i) { geom[i] = make_float4(s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius * s[i].radius); radius[i] = s[i].radius; }
        spt::SphereGrid g;
        spt::build_sphere_grid(geom.data(), radius.data(), n, cells_per_sphere ? (double)cells_per_sphere : 4.0, grid_table_budget(n), g);
        std::string reason = g.why;
        const bool ok = g.usable && spt::validate_sphere_grid(geom.data(), radius.data(), n, g, reason);
        if (out8) {
            out8[0] = (uint32_t)g.P.dim[0]; out8[1] = (uint32_t)g.P.dim[1]; out8[2] = (uint32_t)g.P.dim[2]; out8[3] = g.P.nrefs;
            out8[4] = (uint32_t)g.always.size(); out8[5] = (uint32_t)g.lds_bytes(); out8[6] = g.usable ? 1u : 0u; out8[7] = g.max_cell;
        }
        if (why && why_len) std::snprintf(why, why_len, "%s", reason.c_str());
        return ok ? 0 : 2;
    } catch (const std::exception& e) {
        if (why && why_len) std::snprintf(why, why_len, "%s", e.what());
        return 1;
    }
}

// Host-only self-test of the sphere hierarchy builder (no device call).  out4 = {nodes, leaves, depth, always-tested spheres}.
int spt_selftest_sphere_bvh(const spt_sphere* s, uint32_t n, uint32_t* out4, char* why, uint32_t why_len)
{
    try {
        std::vector<float4> geom(n);
        std::vector<float> radius(n);
        for (uint32_t i = 0; i < n; ++i) { geom[i] = make_float4(s[i].center[0], s[i].center[1], s[i].center[2], s[i].radius * s[i].radius); radius[i] = s[i].radius; }
        spt::Bvh bvh;
        spt::build_sphere_bvh(geom.data(), radius.data(), n, bvh);
        std::string reason;
        const bool ok = spt::validate_sphere_bvh(geom.data(), radius.data(), n, bvh, reason);
        if (out4) { out4[0] = (uint32_t)(bvh.nodes.size() / 4); out4[1] = bvh.leaves; out4[2] = bvh.depth; out4[3] = (uint32_t)bvh.always.size(); }
        if (why && why_len) std::snprintf(why, why_len, "%s", reason.c_str());
        return ok ? 0 : 2;
    } catch (const std::exception& e) {
        if (why && why_len) std::snprintf(why, why_len, "%s", e.what());
        return 1;
    }
}

// ---- triangle meshes (smallpt.cpp:427-473, scene.cpp:3-116) ----
namespace {
inline HostF3 hsub(HostF3 a, HostF3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline HostF3 hld(const float* p) { return {p[0], p[1], p[2]}; }
// material rows as for spheres: {emission, refl | emissive << 2} {color, pmax} {color * (1/pmax), 0}
inline void material_rows(const float e[3], const float col[3], int32_t refl, float4* rows)
{
    const float pmax = std::fmax(std::fmax(col[0], col[1]), col[2]);     // smallpt.cpp:177
    const float inv = 1.0f / pmax;                                       // :192
    const bool emissive = !(e[0] == 0.f && e[1] == 0.f && e[2] == 0.f);
    const int32_t rb = refl | (emissive ? 4 : 0);
    float reflbits;
    std::memcpy(&reflbits, &rb, 4);
    rows[0] = make_float4(e[0], e[1], e[2], reflbits);
    rows[1] = make_float4(col[0], col[1], col[2], pmax);
    rows[2] = make_float4(col[0] * inv, col[1] * inv, col[2] * inv, 0.0f);
}
}  // namespace

uint32_t spt_make_sphere_trimesh(const float origin[3], float radius, uint32_t subdiv_longitude, float* positions, float* normals, uint32_t* indices)
{
    if (!origin || !positions || !normals || !indices || subdiv_longitude == 0) return 0;
    const uint32_t discLong = subdiv_longitude, discLat = 2 * discLong;                 // scene.cpp:5-6
    const float pi = 3.14159265358979323846f, half_pi = 1.57079632679489661923f;        // maths.h:14-15
    const float rcpLat = 1.f / discLat, rcpLong = 1.f / discLong;                       // :8
    const float dPhi = pi * 2.f * rcpLat, dTheta = pi * rcpLong;                        // :9
    uint32_t nv = 0;
    for (uint32_t j = 0; j <= discLong; ++j) {                                          // :13
        const float cosTheta = std::cos(-half_pi + j * dTheta);                         // :15 (float overloads)
        const float sinTheta = std::sin(-half_pi + j * dTheta);                         // :16
        for (uint32_t i = 0; i <= discLat; ++i) {                                       // :18
            const float cx = std::sin(i * dPhi) * cosTheta, cy = sinTheta, cz = std::cos(i * dPhi) * cosTheta;   // :19-23
            positions[3 * nv + 0] = origin[0] + radius * cx;                            // :25
            positions[3 * nv + 1] = origin[1] + radius * cy;
            positions[3 * nv + 2] = origin[2] + radius * cz;
            normals[3 * nv + 0] = cx; normals[3 * nv + 1] = cy; normals[3 * nv + 2] = cz;   // :26
            ++nv;
        }
    }
    uint32_t ni = 0;
    for (uint32_t j = 0; j < discLong; ++j) {                                           // :32
        const uint32_t offset = j * (discLat + 1);
        for (uint32_t i = 0; i < discLat; ++i) {
            indices[ni++] = offset + i; indices[ni++] = offset + (i + 1); indices[ni++] = offset + discLat + 1 + (i + 1);          // :37-39
            indices[ni++] = offset + i; indices[ni++] = offset + discLat + 1 + (i + 1); indices[ni++] = offset + i + discLat + 1;  // :41-43
        }
    }
    return ni / 3;
}

static int set_meshes_impl(spt_ctx* c, const spt_mesh* meshes, uint32_t nmesh, const spt_material* materials);
static int build_accel(spt_ctx* c);

int spt_set_meshes(spt_ctx* c, const spt_mesh* meshes, uint32_t nmesh, const spt_material* materials)
{
    if (!c) return 1;
    try {                                   // host-side tables are std::vectors: no exception may cross the C boundary
        return set_meshes_impl(c, meshes, nmesh, materials);
    } catch (const std::exception& e) {
        return c->fail("spt_set_meshes: %s", e.what());
    }
}

static int set_meshes_impl(spt_ctx* c, const spt_mesh* meshes, uint32_t nmesh, const spt_material* materials)
{
    if (nmesh && (!meshes || !materials)) return c->fail("spt_set_meshes: NULL argument");
    uint64_t ntris = 0, nverts = 0;
    for (uint32_t i = 0; i < nmesh; ++i) {
        const spt_mesh& m = meshes[i];
        if ((m.ntris && !m.indices) || (m.nverts && (!m.positions || !m.normals))) return c->fail("spt_set_meshes: mesh %u has NULL buffers", i);
        if (materials[i].refl < SPT_DIFF || materials[i].refl > SPT_REFR) return c->fail("spt_set_meshes: material %u has refl=%d", i, materials[i].refl);
        for (uint64_t k = 0; k < (uint64_t)m.ntris * 3; ++k)
            if (m.indices[k] >= m.nverts) return c->fail("spt_set_meshes: mesh %u index %u out of range (%u vertices)", i, m.indices[k], m.nverts);
        ntris += m.ntris; nverts += m.nverts;
    }
    if (ntris > 0x7FFFFFFFull || nverts > 0x7FFFFFFFull) return c->fail("spt_set_meshes: too many triangles");
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) SPT_HIP(c, hipEventSynchronize(c->ev_stop));
    // flatten: triangle records with the per-call constants of triIntersect (scene.cpp:56-60) evaluated once
    std::vector<float4> tris(3 * (size_t)(ntris ? ntris : 1)), verts(2 * (size_t)(nverts ? nverts : 1)), mats(3 * (size_t)(nmesh ? nmesh : 1));
    std::vector<uint4> tidx((size_t)(ntris ? ntris : 1));
    std::vector<uint32_t> first((size_t)nmesh + 1, 0u);
    size_t t = 0, vbase = 0;
    bool specular = false;
    for (uint32_t i = 0; i < nmesh; ++i) {
        const spt_mesh& m = meshes[i];
        first[i] = (uint32_t)t;
        for (uint32_t v = 0; v < m.nverts; ++v) {
            verts[2 * (vbase + v)] = make_float4(m.positions[3 * v], m.positions[3 * v + 1], m.positions[3 * v + 2], 0.f);
            verts[2 * (vbase + v) + 1] = make_float4(m.normals[3 * v], m.normals[3 * v + 1], m.normals[3 * v + 2], 0.f);
        }
        for (uint32_t k = 0; k < m.ntris; ++k, ++t) {
            const uint32_t i1 = m.indices[3 * k], i2 = m.indices[3 * k + 1], i3 = m.indices[3 * k + 2];
            const HostF3 v0 = hld(m.positions + 3 * i1), v1 = hld(m.positions + 3 * i2), v2 = hld(m.positions + 3 * i3);
            const HostF3 e1 = hsub(v1, v0), e2 = hsub(v2, v0);                          // scene.cpp:56-57
            const HostF3 n = hcross(e1, e2);                                            // :60
            tris[3 * t] = make_float4(v0.x, v0.y, v0.z, n.x);
            tris[3 * t + 1] = make_float4(e1.x, e1.y, e1.z, n.y);
            tris[3 * t + 2] = make_float4(e2.x, e2.y, e2.z, n.z);
            tidx[t] = make_uint4((uint32_t)vbase + i1, (uint32_t)vbase + i2, (uint32_t)vbase + i3, i);
        }
        vbase += m.nverts;
        material_rows(materials[i].emission, materials[i].color, materials[i].refl, &mats[3 * (size_t)i]);
        specular = specular || materials[i].refl != SPT_DIFF;
    }
    first[nmesh] = (uint32_t)t;
    auto upload = [&](auto*& dptr, const void* src, size_t bytes) -> hipError_t {
        if (dptr) (void)hipFree(dptr);
        dptr = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dptr), bytes);
        if (e != hipSuccess) return e;
        return hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice);
    };
    SPT_HIP(c, upload(c->d_tris, tris.data(), tris.size() * sizeof(float4)));
    SPT_HIP(c, upload(c->d_tri_index, tidx.data(), tidx.size() * sizeof(uint4)));
    SPT_HIP(c, upload(c->d_verts, verts.data(), verts.size() * sizeof(float4)));
    SPT_HIP(c, upload(c->d_inst_first, first.data(), first.size() * sizeof(uint32_t)));
    SPT_HIP(c, upload(c->d_mesh_mats, mats.data(), mats.size() * sizeof(float4)));
    c->ntris = (uint32_t)ntris; c->ninst = nmesh;
    c->mesh_scene = true;
    c->mesh_specular = specular;
    c->mesh_ratio = -1.f;
    tris.resize(3 * (size_t)ntris);
    c->h_tris.swap(tris);
    c->bvh_ready = false;
    return c->accel != SPT_ACCEL_EXHAUSTIVE ? build_accel(c) : 0;
}

// Builds and uploads the hierarchy of the current mesh scene (spt_bvh.h); the caller holds the C-boundary try block.
static int build_accel(spt_ctx* c)
{
    spt::Bvh bvh;
    spt::build_bvh(c->h_tris.data(), c->ntris, bvh);
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) SPT_HIP(c, hipEventSynchronize(c->ev_stop));
    auto upload = [&](auto*& dptr, const void* src, size_t bytes) -> hipError_t {
        if (dptr) (void)hipFree(dptr);
        dptr = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dptr), bytes);
        if (e != hipSuccess) return e;
        return hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice);
    };
    SPT_HIP(c, upload(c->d_bvh_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(float4)));
    SPT_HIP(c, upload(c->d_bvh_tris, bvh.tris.data(), bvh.tris.size() * sizeof(float4)));
    SPT_HIP(c, upload(c->d_bvh_index, bvh.index.data(), bvh.index.size() * sizeof(uint32_t)));
    c->have_planes = !bvh.planes.empty(); c->have_lines = !bvh.lines.empty();
    if (c->have_planes) SPT_HIP(c, upload(c->d_plane_nodes, bvh.planes.data(), bvh.planes.size() * sizeof(float4)));
    if (c->have_lines) SPT_HIP(c, upload(c->d_line_nodes, bvh.lines.data(), bvh.lines.size() * sizeof(float4)));
    c->bvh_flat = bvh.flat && bvh.thin_count; c->nline_slots = (uint32_t)bvh.flat_lines.size(); c->cam_valid = false;
    if (c->bvh_flat) {
        SPT_HIP(c, upload(c->d_flat_lines, bvh.flat_lines.data(), bvh.flat_lines.size() * sizeof(float4)));
        SPT_HIP(c, upload(c->d_flat_line_index, bvh.flat_line_index.data(), bvh.flat_line_index.size() * sizeof(uint32_t)));
    }
    SPT_HIP(c, upload(c->d_bvh_cones, bvh.cones.data(), bvh.cones.size() * sizeof(float4)));
    c->bvh_nodes = (uint32_t)(bvh.nodes.size() / 4); c->bvh_depth = bvh.depth; c->bvh_leaves = bvh.leaves;
    c->bvh_ready = true;
    return 0;
}

int spt_set_mesh_accel(spt_ctx* c, int accel)
{
    if (!c) return 1;
    if (accel != SPT_ACCEL_EXHAUSTIVE && accel != SPT_ACCEL_BVH && accel != SPT_ACCEL_BVH_FAST && accel != SPT_ACCEL_AUTO) return c->fail("spt_set_mesh_accel: unknown mode %d", accel);
    c->accel = accel;
    if (accel == SPT_ACCEL_EXHAUSTIVE || !c->mesh_scene || c->bvh_ready) return 0;
    try {
        return build_accel(c);
    } catch (const std::exception& e) {
        return c->fail("spt_set_mesh_accel: %s", e.what());
    }
}

// Host-only self-test of the builder (no device call): builds the hierarchy over the meshes' triangles and validates it.
// out4 = {nodes, leaves, depth, triangles}.
int spt_selftest_bvh(const spt_mesh* meshes, uint32_t nmesh, uint32_t* out4, char* why, uint32_t why_len)
{
    try {
        std::vector<float4> recs;
        for (uint32_t i = 0; i < nmesh; ++i) {
            const spt_mesh& m = meshes[i];
            for (uint32_t k = 0; k < m.ntris; ++k) {
                const uint32_t i1 = m.indices[3 * k], i2 = m.indices[3 * k + 1], i3 = m.indices[3 * k + 2];
                if (i1 >= m.nverts || i2 >= m.nverts || i3 >= m.nverts) throw std::runtime_error("index out of range");
                const HostF3 v0 = hld(m.positions + 3 * i1), v1 = hld(m.positions + 3 * i2), v2 = hld(m.positions + 3 * i3);
                const HostF3 e1 = hsub(v1, v0), e2 = hsub(v2, v0);
                const HostF3 n = hcross(e1, e2);
                recs.push_back(make_float4(v0.x, v0.y, v0.z, n.x));
                recs.push_back(make_float4(e1.x, e1.y, e1.z, n.y));
                recs.push_back(make_float4(e2.x, e2.y, e2.z, n.z));
            }
        }
        const uint32_t ntris = (uint32_t)(recs.size() / 3);
        spt::Bvh bvh;
        spt::build_bvh(recs.data(), ntris, bvh);
        std::string reason;
        const bool ok = spt::validate_bvh(recs.data(), ntris, bvh, reason);
        if (out4) { out4[0] = (uint32_t)(bvh.nodes.size() / 4); out4[1] = bvh.leaves; out4[2] = bvh.depth; out4[3] = bvh.regular_count; }
        if (why && why_len) std::snprintf(why, why_len, "%s", reason.c_str());
        return ok ? 0 : 2;
    } catch (const std::exception& e) {
        if (why && why_len) std::snprintf(why, why_len, "%s", e.what());
        return 1;
    }
}

// Which closest-hit mode a launch (render = true) or a ray query of the current mesh scene takes.  SPT_ACCEL_AUTO picks between the two EXACT
// modes: the hierarchy, unless the scene is tiny (< 256 triangles) or -- for renders -- small (< 8192 triangles) and, in its last launch, more than
// 15 % of the closest-hit queries were bounce rays: those walk the plane tree (only camera rays have their list), and below that size the
// exhaustive loop is then the faster exact mode (profiles/r04_triangle_hierarchy.txt, size sweep).
static int mesh_mode(const spt_ctx* c, bool render)
{
    if (c->accel != SPT_ACCEL_AUTO) return c->bvh_ready || c->accel == SPT_ACCEL_EXHAUSTIVE ? c->accel : SPT_ACCEL_EXHAUSTIVE;
    if (!c->bvh_ready || c->ntris < 256u) return SPT_ACCEL_EXHAUSTIVE;
    if (render && c->ntris < 8192u && c->mesh_ratio > 1.15f) return SPT_ACCEL_EXHAUSTIVE;
    return SPT_ACCEL_BVH;
}

static spt::MParams mesh_params(const spt_ctx* c, int mode)
{
    spt::MParams M{};
    M.tris = c->d_tris; M.tri_index = c->d_tri_index; M.verts = c->d_verts; M.inst_first_tri = c->d_inst_first; M.mats = c->d_mesh_mats;
    M.ntris = c->ntris; M.ninst = c->ninst;
    M.strips = c->mesh_specular ? 0u : 1u;
    if (mode != SPT_ACCEL_EXHAUSTIVE && c->bvh_ready) {
        M.bvh_nodes = c->d_bvh_nodes; M.bvh_tris = c->d_bvh_tris; M.bvh_index = c->d_bvh_index;
        if (mode == SPT_ACCEL_BVH) {                           // (SPT_ACCEL_BVH_FAST: the spatial tree alone, no cones)
            M.bvh_cones = c->d_bvh_cones;
            if (c->have_planes) M.plane_nodes = c->d_plane_nodes;
            if (c->have_lines) M.line_nodes = c->d_line_nodes;
            if (c->bvh_flat) { M.flat_lines = c->d_flat_lines; M.flat_line_index = c->d_flat_line_index; M.nline_slots = c->nline_slots; }
        }
    }
    return M;
}

int spt_trace_rays_device(spt_ctx* c, const void* d_rays, uint64_t n, void* d_hits, void* hip_stream)
{
    if (!c) return 1;
    if (!c->mesh_scene) return c->fail("spt_trace_rays_device: no mesh scene set (call spt_set_meshes)");
    if (n == 0) return 0;
    if (!d_rays || !d_hits) return c->fail("spt_trace_rays_device: NULL argument");
    if (n > 0x7FFFFFFFull * 256ull) return c->fail("spt_trace_rays_device: too many rays for one call");
    SPT_HIP(c, hipSetDevice(c->device));
    c->last_mesh_mode = mesh_mode(c, false);
    const spt::MParams M = mesh_params(c, c->last_mesh_mode);
    SPT_HIP(c, spt_mesh_trace_rays(&M, static_cast<const float*>(d_rays), n, static_cast<float*>(d_hits),
                                   hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream));
    return 0;
}

int spt_trace_rays(spt_ctx* c, const spt_ray* rays, uint64_t n, spt_hit* hits)
{
    if (!c) return 1;
    if (!c->mesh_scene) return c->fail("spt_trace_rays: no mesh scene set (call spt_set_meshes)");
    if (n == 0) return 0;
    if (!rays || !hits) return c->fail("spt_trace_rays: NULL argument");
    if (n > 0x7FFFFFFFull * 256ull) return c->fail("spt_trace_rays: too many rays for one call");
    static_assert(sizeof(spt_ray) == 24 && sizeof(spt_hit) == 44, "Ray / Hit layouts of scene.h");
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) { SPT_HIP(c, hipEventSynchronize(c->ev_stop)); }
    // ray / hit staging buffers are kept between calls (traceRays is called once per bounce by the reference's render loop)
    hipError_t e = hipSuccess;
    if (n > c->trace_cap) {
        if (c->d_trace_rays) (void)hipFree(c->d_trace_rays);
        if (c->d_trace_hits) (void)hipFree(c->d_trace_hits);
        c->d_trace_rays = c->d_trace_hits = nullptr; c->trace_cap = 0;
        e = hipMalloc(reinterpret_cast<void**>(&c->d_trace_rays), n * sizeof(spt_ray));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_trace_hits), n * sizeof(spt_hit));
        if (e == hipSuccess) c->trace_cap = n;
    }
    float* const d_rays = c->d_trace_rays;
    float* const d_hits = c->d_trace_hits;
    c->last_mesh_mode = mesh_mode(c, false);
    const spt::MParams M = mesh_params(c, c->last_mesh_mode);
    // 24 B per ray up and 44 B per hit down through the caller's pageable buffers: the host link is the bound (measured 176 Mrays/s
    // for 1 Mi rays against 1.3 Grays/s of the kernel on the shipped scene's hierarchy; chunks on two streams were tried and are
    // slower, pageable copies do not overlap).  spt_trace_rays_device skips the link.
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n * sizeof(spt_ray), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = spt_mesh_trace_rays(&M, d_rays, n, d_hits, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hits, d_hits, n * sizeof(spt_hit), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return c->fail("spt_trace_rays: %s", hipGetErrorString(e));
    return 0;
}

// smallpt.cpp:277-279 (D10: cx = (w*.5135/h, 0, 0))
int spt_camera_smallpt(uint32_t w, uint32_t h, spt_camera* out)
{
    if (!out || w == 0 || h == 0) return 1;
    const HostF3 o{50, 52, 295.6f};
    const HostF3 dir = hnormalize(HostF3{0, (float)-0.042612, -1});
    const HostF3 cx{(float)((int)w * .5135 / (int)h), 0, 0};
    const HostF3 cy = hscl(hnormalize(hcross(cx, dir)), (float).5135);
    out->origin[0] = o.x; out->origin[1] = o.y; out->origin[2] = o.z;
    out->dir[0] = dir.x; out->dir[1] = dir.y; out->dir[2] = dir.z;
    out->cx[0] = cx.x; out->cx[1] = cx.y; out->cx[2] = cx.z;
    out->cy[0] = cy.x; out->cy[1] = cy.y; out->cy[2] = cy.z;
    out->push = 140.0f;
    out->sampler = SPT_SAMPLER_SMALLPT;
    return 0;
}

// Camera ctor smallpt.cpp:609-618 + sampleRay :635: direction = localToWorld * (clip.x, clip.y, near, 0)
// = (vx*clip.x + vy*clip.y) + vz*near (+ org*0); vz*near is the same product for every sample.
int spt_camera_pinhole(const float vx[3], const float vy[3], const float vz[3], const float org[3], float near_plane_distance, spt_camera* out)
{
    if (!vx || !vy || !vz || !org || !out) return 1;
    for (int i = 0; i < 3; ++i) {
        out->cx[i] = vx[i]; out->cy[i] = vy[i];
        out->dir[i] = vz[i] * near_plane_distance;
        out->origin[i] = org[i];
    }
    out->push = 0.0f;
    out->sampler = SPT_SAMPLER_PINHOLE;
    return 0;
}

static int render_rows_impl(spt_ctx* c, const spt_camera* cam, uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
                            uint32_t rb_log2, uint32_t rb_stride, uint32_t rb_mask, uint32_t samps, uint64_t seed,
                            uint32_t flags, void* d_out_rgb, void* hip_stream);

int spt_render_rows_device(spt_ctx* c, const spt_camera* cam, uint32_t w, uint32_t h,
                           uint32_t row_begin, uint32_t row_count, uint32_t samps, uint64_t seed,
                           uint32_t flags, void* d_out_rgb, void* hip_stream)
{
    if (!c) return 1;
    if (row_count == 0 || (uint64_t)row_begin + row_count > h) return c->fail("spt_render_rows_device: row band [%u,+%u) outside image height %u", row_begin, row_count, h);
    return render_rows_impl(c, cam, w, h, row_begin, row_count, 0u, 1u, 0u, samps, seed, flags, d_out_rgb, hip_stream);
}

// Rows dealt out round-robin in blocks of `block_rows` rows: block t of the image (rows [t*B, (t+1)*B)) belongs to rank t % world.
uint32_t spt_interleaved_row_count(uint32_t h, uint32_t block_rows, uint32_t world, uint32_t rank)
{
    if (!block_rows || !world || rank >= world) return 0;
    const uint32_t nblk = (h + block_rows - 1) / block_rows;            // blocks of the image, the last one may be short
    if (rank >= nblk) return 0;
    const uint32_t mine = (nblk - 1 - rank) / world + 1;                 // blocks rank, rank + world, ...
    uint32_t rows = mine * block_rows;
    const uint32_t last = rank + (mine - 1) * world;                     // this rank's last block
    if (last == nblk - 1) rows -= nblk * block_rows - h;                 // ... is the image's short last block
    return rows;
}

int spt_render_interleaved_device(spt_ctx* c, const spt_camera* cam, uint32_t w, uint32_t h, uint32_t block_rows,
                                  uint32_t world, uint32_t rank, uint32_t samps, uint64_t seed, uint32_t flags,
                                  void* d_out_rgb, void* hip_stream)
{
    if (!c) return 1;
    if (!block_rows || (block_rows & (block_rows - 1))) return c->fail("spt_render_interleaved_device: block_rows must be a power of two");
    if (!world || rank >= world) return c->fail("spt_render_interleaved_device: rank %u of %u", rank, world);
    if ((uint64_t)world * block_rows > 0x7FFFFFFFull) return c->fail("spt_render_interleaved_device: world * block_rows too large");
    const uint32_t rows = spt_interleaved_row_count(h, block_rows, world, rank);
    if (rows == 0) return c->fail("spt_render_interleaved_device: rank %u owns no rows of a %u-row image", rank, h);
    uint32_t lb = 0;
    while ((1u << lb) < block_rows) ++lb;
    return render_rows_impl(c, cam, w, h, rank * block_rows, rows, lb, world * block_rows, block_rows - 1u, samps, seed, flags, d_out_rgb, hip_stream);
}

static int render_rows_impl(spt_ctx* c, const spt_camera* cam, uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_count,
                            uint32_t rb_log2, uint32_t rb_stride, uint32_t rb_mask, uint32_t samps, uint64_t seed,
                            uint32_t flags, void* d_out_rgb, void* hip_stream)
{
    if (!cam || !d_out_rgb) return c->fail("spt_render_rows_device: NULL argument");
    if (w == 0 || h == 0 || samps == 0) return c->fail("spt_render_rows_device: empty image or samps == 0");
    if ((uint64_t)w * h > 0xFFFFFFFFull) return c->fail("spt_render_rows_device: w*h exceeds 2^32-1 pixels");
    if ((uint64_t)samps * 4 > 0xFFFFFFFFull) return c->fail("spt_render_rows_device: spp overflows 32 bits");
    const uint64_t npix = (uint64_t)row_count * w;
    // D9: a jitter cell's samples are accumulated in nb = 1, 2, 4 or 8 blocks (>= 16 samples each); one task = one block
    const uint32_t nb_log2 = samps >= 128u ? 3u : (samps >= 64u ? 2u : (samps >= 32u ? 1u : 0u));
    const uint32_t nb = 1u << nb_log2;
    if (npix * 4 * nb > 0xF0000000ull) return c->fail("spt_render_rows_device: band has more than 15*2^26 sample blocks (%u per pixel); split it", 4u * nb);
    if (!c->d_geom && !c->mesh_scene) return c->fail("spt_render_rows_device: no scene set (call spt_set_scene)");
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) { SPT_HIP(c, hipEventSynchronize(c->ev_stop)); }

    const size_t ntasks = (size_t)npix * 4 * nb;
    if (ntasks > c->cells_cap) {
        if (c->d_cells) (void)hipFree(c->d_cells);
        c->d_cells = nullptr; c->cells_cap = 0;
        SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_cells), ntasks * sizeof(float4)));
        c->cells_cap = ntasks;
    }

    spt::KParams P{};
    std::memcpy(P.cam_o, cam->origin, 12); std::memcpy(P.cam_d, cam->dir, 12);
    std::memcpy(P.cam_cx, cam->cx, 12); std::memcpy(P.cam_cy, cam->cy, 12);
    P.cam_push = cam->push;
    if (cam->sampler > SPT_SAMPLER_PINHOLE) return c->fail("spt_render_rows_device: unknown camera sampler %u", cam->sampler);
    P.sampler = cam->sampler;
    P.inv_wf = 1.f / (float)w; P.inv_hf = 1.f / (float)h;   // pixelSize, smallpt.cpp:746
    P.w = w; P.h = h; P.row_begin = row_begin; P.row_count = row_count;
    P.rb_log2 = rb_log2; P.rb_stride = rb_stride; P.rb_mask = rb_mask;
    P.inv_w = 1.0 / (double)w; P.inv_h = 1.0 / (double)h;
    P.samps = samps; P.ntasks = (uint32_t)ntasks;
    P.nb_log2 = nb_log2; P.sb = (samps + nb - 1u) / nb;
    P.park_threshold = (c->variant & 0xFFu) ? (c->variant & 0xFFu) : 8u;
    P.s0 = mix32((uint32_t)seed + 0x243F6A88u);
    P.s1 = mix32((uint32_t)(seed >> 32) ^ P.s0 ^ 0x85A308D3u);
    P.n = c->n; P.n_pad = c->n ? c->n : 1;
    P.geom = c->d_geom; P.mat = c->d_mat;
    P.cells = c->d_cells; P.queue = c->d_queue; P.counters = c->d_counters;

    const float cam_big = std::fmax(std::fmax(std::fabs(cam->origin[0]), std::fabs(cam->origin[1])),
                                    std::fmax(std::fabs(cam->origin[2]), std::fabs(cam->push)));
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    const float scale = 1.0f / (float)(4u * samps);   // smallpt.cpp:360 operator/=(float3, float)

    // ---- large sphere table through its uniform grid (spt_grid.hip): the default above the pool kernel's limit ----
    if (!c->mesh_scene && c->sphere_accel == SPT_ACCEL_GRID && c->grid_ready && cam_big <= 1e15f && !(c->variant & 0x400u)) {
        // one 1024-thread workgroup per CU shares the LDS tables (tuning: variant bits 15:13 = threads / 128 - 1 ... 0 = 1024; blocks_per_cu)
        const uint32_t tsel = (c->variant >> 13) & 7u;
        const uint32_t threads = tsel ? 128u * (tsel + 1u) : (uint32_t)spt_grid_block_threads();
        const uint32_t blocks = (uint32_t)c->cu_count * (c->blocks_per_cu ? c->blocks_per_cu : 1u);
        const size_t need_stack = spt_grid_stack_floats(blocks, threads);
        if (need_stack > c->stack_cap) {
            if (c->d_stack) (void)hipFree(c->d_stack);
            c->d_stack = nullptr; c->stack_cap = 0;
            SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_stack), need_stack * sizeof(float)));
            c->stack_cap = need_stack;
        }
        P.stack = c->d_stack;
        P.watchdog_ticks = c->watchdog_ticks;
        const uint32_t lsel = (c->variant >> 16) & 0xFFu;
        // ---- round 4: wave-private path pools with register-resident walkers (spt_gpool.hip) whenever the tables leave the LDS for
        // them: R begun walks of 64 bytes + two byte lists per wave beside the grid tables.  spt_set_grid_pools (internal) keeps the
        // lane-owned kernel or changes the pool geometry; SPT_GPOOL="S,R,drain,min_batch[,walk_iters]" overrides it per process (tools). ----
        if (!c->grid_lane_owned && !c->grid_global) {
            static const char* env = std::getenv("SPT_GPOOL");
            uint32_t S = c->gq[0], Rwant = c->gq[1], drain = c->gq[2], minb = c->gq[3], witers = c->gq[4];
            if (env) { unsigned a = 0, b2 = 0, d2 = 0, m2 = 0, w2 = 0; const int got = std::sscanf(env, "%u,%u,%u,%u,%u", &a, &b2, &d2, &m2, &w2); if (got >= 4) { S = a; Rwant = b2; drain = d2; minb = m2; } if (got == 5) witers = w2; }
            const uint32_t waves = threads / 64u;
            const size_t fixed = spt_gpool_lds_bytes(&c->grid, waves, S, 0);
            const size_t room = fixed < (size_t)160 * 1024 ? (size_t)160 * 1024 - fixed : 0;
            uint32_t R = (uint32_t)(room / ((size_t)waves * 64u)) & ~3u;
            if (R > Rwant) R = Rwant & ~3u;
            if (R >= 48u && c->n <= 0xC000u && c->grid.nrefs < 0x7FFEu) {
                const size_t stack_floats = spt_gpool_stack_floats(blocks, waves, S);
                const size_t need = stack_floats + spt_gpool_slot_floats(blocks, waves, S);
                if (need > c->stack_cap) {
                    if (c->d_stack) (void)hipFree(c->d_stack);
                    c->d_stack = nullptr; c->stack_cap = 0;
                    SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_stack), need * sizeof(float)));
                    c->stack_cap = need;
                }
                P.stack = c->d_stack;
                spt::QParams Q{};
                Q.slots = reinterpret_cast<float4*>(c->d_stack + stack_floats);
                Q.S = S; Q.R = R; Q.drain = drain; Q.min_batch = minb; Q.walk_iters = witers ? witers : 1u;
                SPT_HIP(c, hipMemsetAsync(c->d_queue, 0, 256, st));
                SPT_HIP(c, hipEventRecord(c->ev_start, st));
                SPT_HIP(c, spt_gpool_launch(&P, &c->grid, c->d_grid_cells, c->d_grid_refs, c->d_grid_always, &Q, blocks, threads, (c->variant & 0x100u) ? 1 : 0, st));
                SPT_HIP(c, hipEventRecord(c->ev_mid, st));
                SPT_HIP(c, spt_k_finalize(c->d_cells, static_cast<float*>(d_out_rgb), (uint32_t)npix, scale, (flags & SPT_FLAG_NORMALISE) ? 1 : 0, nb, st));
                SPT_HIP(c, hipEventRecord(c->ev_stop, st));
                c->pending = true;
                c->last_was_pool = false;
                c->last_kernel = 5;
                c->last = spt_stats{};
                c->last.samples = npix * 4ull * samps;
                c->last.grid_blocks = blocks;
                c->last.block_threads = threads;
                return 0;
            }
        }
        SPT_HIP(c, hipMemsetAsync(c->d_queue, 0, 256, st));
        SPT_HIP(c, hipEventRecord(c->ev_start, st));
        SPT_HIP(c, spt_grid_launch(&P, &c->grid, c->d_grid_cells, c->d_grid_refs, c->d_grid_always, blocks, threads, lsel ? lsel - 1u : 16u, (c->variant & 0x100u) ? 1 : 0,
                                   c->grid_global, st));
        SPT_HIP(c, hipEventRecord(c->ev_mid, st));
        SPT_HIP(c, spt_k_finalize(c->d_cells, static_cast<float*>(d_out_rgb), (uint32_t)npix, scale, (flags & SPT_FLAG_NORMALISE) ? 1 : 0, nb, st));
        SPT_HIP(c, hipEventRecord(c->ev_stop, st));
        c->pending = true;
        c->last_was_pool = false;
        c->last_kernel = 4;
        c->last = spt_stats{};
        c->last.samples = npix * 4ull * samps;
        c->last.grid_blocks = blocks;
        c->last.block_threads = threads;
        return 0;
    }

    // ---- triangle-mesh scene (spt_mesh.hip), or a sphere table too large for the pool kernel through its hierarchy ----
    const bool sphere_bvh = !c->mesh_scene && c->sbvh_ready && c->n > (uint32_t)spt_pool_max_spheres() &&
                            (c->sphere_accel == SPT_ACCEL_BVH || (c->sphere_accel == SPT_ACCEL_GRID && !c->grid_ready && c->n >= kSphereBvhFrom && !c->needs_guard));
    if (!c->mesh_scene && !sphere_bvh && c->n > SPT_MAX_SPHERES)    // (a grid table whose launch conditions this call does not meet)
        return c->fail("spt_render_rows_device: %u spheres > SPT_MAX_SPHERES (%u) render through the grid only with camera coordinates within 1e15 and without the exhaustive-kernel tuning bit; use SPT_ACCEL_BVH", c->n, SPT_MAX_SPHERES);
    if (c->mesh_scene || sphere_bvh) {
        // (a short launch -- the viewer's frames -- takes three workgroups per CU instead of four: 661 -> 672 frames/s on the shipped scene
        // through the hierarchy at 1280x720 x 4 spp, 949 -> 1033 with two frames in flight, which then share the CUs; tools/ab_mesh_viewer_blocks.py)
        const int mode = c->mesh_scene ? mesh_mode(c, true) : SPT_ACCEL_EXHAUSTIVE;
        if (c->mesh_scene) c->last_mesh_mode = mode;
        const bool through_hierarchy = sphere_bvh || (c->mesh_scene && mode != SPT_ACCEL_EXHAUSTIVE);   // (the exhaustive tile loop keeps four)
        const uint32_t mesh_per_cu = c->blocks_per_cu ? c->blocks_per_cu : (through_hierarchy && npix * 4ull * samps < (4ull << 20) ? 3u : 4u);
        uint64_t blocks = (uint64_t)c->cu_count * mesh_per_cu;
        const uint64_t needed = (ntasks + 255) / 256;
        if (blocks > needed) blocks = needed;
        if (blocks < 1) blocks = 1;
        const size_t need_stack = spt_mesh_stack_floats((uint32_t)blocks);
        if (need_stack > c->stack_cap) {
            if (c->d_stack) (void)hipFree(c->d_stack);
            c->d_stack = nullptr; c->stack_cap = 0;
            SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_stack), need_stack * sizeof(float)));
            c->stack_cap = need_stack;
        }
        P.stack = c->d_stack;
        spt::MParams M{};
        if (sphere_bvh) {
            M.bvh_nodes = c->d_sbvh_nodes; M.bvh_tris = c->d_sbvh_geom; M.bvh_index = c->d_sbvh_index;
            M.always = c->d_sbvh_always; M.nalways = c->sbvh_nalways; M.sphere_mode = 1u;
        } else {
            P.n = 0; P.n_pad = 1; P.geom = nullptr; P.mat = nullptr;
            M = mesh_params(c, mode);
            if (M.plane_nodes) {
                // every ray of depth 0 lies on a line through cam->origin and starts at most |push| |d| from it (push = 0: a pinhole
                // camera, every ray starts there), and a ray can only be reported by a regular triangle through a determinant that is
                // zero to rounding if its origin lies in that triangle's plane (spt_tribvh.h (2), condition (B)): those triangles are
                // listed once per camera (none, as a rule) and the camera rays skip the plane tree
                auto norm3 = [](const float* v) { return std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); };
                const float extra = (float)(std::fabs((double)cam->push) * (norm3(cam->dir) + 1.1 * (norm3(cam->cx) + norm3(cam->cy))) * 1.01);
                const float key[4] = {cam->origin[0], cam->origin[1], cam->origin[2], extra};
                if (!c->cam_valid || std::memcmp(c->cam_key, key, sizeof c->cam_key) != 0) {
                    std::vector<uint32_t> list;
                    spt::camera_planes(c->h_tris.data(), c->ntris, cam->origin, extra, list);
                    if (list.size() > c->cam_cap) {
                        if (c->d_cam_planes) (void)hipFree(c->d_cam_planes);
                        c->d_cam_planes = nullptr; c->cam_cap = 0;
                        SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_cam_planes), list.size() * sizeof(uint32_t)));
                        c->cam_cap = (uint32_t)list.size();
                    }
                    if (!list.empty()) SPT_HIP(c, hipMemcpyAsync(c->d_cam_planes, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    if (!list.empty()) SPT_HIP(c, hipStreamSynchronize(st));           // (the list is a local)
                    c->ncam = (uint32_t)list.size();
                    std::memcpy(c->cam_key, key, sizeof c->cam_key);
                    c->cam_valid = true;
                }
                M.cam_planes = c->d_cam_planes; M.ncam = c->ncam; M.cam_cull = 1u;
            }
        }
        SPT_HIP(c, hipMemsetAsync(c->d_queue, 0, 256, st));
        SPT_HIP(c, hipEventRecord(c->ev_start, st));
        SPT_HIP(c, spt_mesh_launch(&P, &M, (uint32_t)blocks, st));
        SPT_HIP(c, hipEventRecord(c->ev_mid, st));
        SPT_HIP(c, spt_k_finalize(c->d_cells, static_cast<float*>(d_out_rgb), (uint32_t)npix, scale, (flags & SPT_FLAG_NORMALISE) ? 1 : 0, nb, st));
        SPT_HIP(c, hipEventRecord(c->ev_stop, st));
        c->pending = true;
        c->last_was_pool = false;
        c->last_kernel = sphere_bvh ? 3 : (mode == SPT_ACCEL_BVH ? 6 : (mode == SPT_ACCEL_BVH_FAST ? 7 : 2));
        c->last = spt_stats{};
        c->last.samples = npix * 4ull * samps;
        c->last.grid_blocks = (uint32_t)blocks;
        c->last.block_threads = 256;
        return 0;
    }

    // ---- material-sorted pool kernel (spt_pool.hip): small tables, regular scenes; variant bit 10 forces the megakernel ----
    if (c->pool_ok && cam_big <= 1e15f && !(c->variant & 0x500u)) {
        const uint32_t psel = (c->variant >> 11) & 3u;
        const int pool = psel == 1 ? 96 : (psel == 2 ? 192 : (psel == 3 ? 128 : spt_pool_default_slots()));   // default: four workgroups per CU
        const size_t lds = spt_pool_lds_bytes(P.n, pool);
        uint32_t per_cu = c->blocks_per_cu;
        if (per_cu == 0) {
            const uint32_t by_lds = (uint32_t)((160u * 1024u) / lds);
            per_cu = by_lds < 1 ? 1 : (by_lds > 8 ? 8 : by_lds);
            // A short launch (the viewer's frames: 3.7 M samples = 900 per wave of a full grid) is over before the slot pools of four
            // workgroups per CU ever run full: with half the waves the batches are fuller, and the other half of every CU's LDS is free
            // for the next frame's kernel when several frames are in flight.  Measured at 1280x720 x 4 spp, frames/s with 4 / 2 / 1
            // workgroups per CU (profiles/r03_small_launch_ab.txt): one frame at a time 427 / 450 / 411, two in flight 675 / 773 / 730,
            // four 935 / 1155 / 1152, eight 1197 / 1575 / 1756; from 8 spp on (1024x768) a single launch is faster with four again.
            // So: launches below 4 Mi samples take two, one when six or more frames are in flight (the lanes of
            // spt_progressive_frame_async know their number).  SPT_SMALL_LAUNCH_BLOCKS overrides (experiments).
            static const uint32_t forced = [] { const char* e = std::getenv("SPT_SMALL_LAUNCH_BLOCKS"); return e ? (uint32_t)std::atoi(e) : 0u; }();
            const uint32_t small_blocks = forced ? forced : (c->frames_in_flight_hint >= 6u ? 1u : 2u);
            if (npix * 4ull * samps < (4ull << 20) && per_cu > small_blocks) per_cu = small_blocks;
        }
        uint64_t blocks = (uint64_t)c->cu_count * per_cu;
        const uint64_t needed = (ntasks + 255) / 256;
        if (blocks > needed) blocks = needed;
        if (blocks < 1) blocks = 1;
        // one allocation: the children stack followed by the {task, next sample} words of every slot
        const size_t stack_floats = spt_pool_stack_floats((uint32_t)blocks, pool);
        const size_t need_stack = stack_floats + spt_pool_state_bytes((uint32_t)blocks, pool) / sizeof(float);
        if (need_stack > c->stack_cap) {
            if (c->d_stack) (void)hipFree(c->d_stack);
            c->d_stack = nullptr; c->stack_cap = 0;
            SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_stack), need_stack * sizeof(float)));
            c->stack_cap = need_stack;
        }
        P.stack = c->d_stack;
        P.slot_state = reinterpret_cast<uint2*>(c->d_stack + stack_floats);
        P.watchdog_ticks = c->watchdog_ticks;
        // Cost-ordered dispatch: the queue hands out chunks of 64 tasks; every launch records how long each chunk kept its wave busy, and a
        // launch of the SAME view with the SAME seed (a repeated render) starts the expensive chunks first.  Round 3 applied the order to
        // any seed of the view ("a pixel's cost is a property of what it looks at"); measured with the seed stepped every launch that is
        // wrong at the granularity of a chunk -- 80.4-80.7 ms against 79.3-79.7 in the static order, also when only the most expensive
        // 1/64 of the chunks is moved to the front (profiles/r04_cost_order_seeds.txt): a chunk's time is mostly the luck of its 2048
        // samples and of when its wave ran it -- so the seed is part of the key now, and a new seed runs in the static order like a first
        // launch.  Recording is not free either -- the clock stores and the three ordering kernels behind the frame cost 0.6 ms of an 80 ms
        // launch (profiles/r04_cost_order_regions.txt) --, so a launch records only when it repeats its predecessor (same view, same seed)
        // or an order for it exists: a progressive loop never pays, a repeated render runs twice in the static order and is ordered from
        // its third launch on.  Results do not depend on the dispatch order.  Tuning bit 13 switches it off for this kernel (A/B),
        // SPT_FLAG_ONE_SHOT for one launch.
        const uint32_t nchunks = (uint32_t)((ntasks + 63) / 64);
        std::vector<unsigned char> key(sizeof(spt_camera) + 10 * sizeof(uint32_t) + 2 * sizeof(uint64_t));
        {
            unsigned char* k = key.data();
            std::memcpy(k, cam, sizeof(spt_camera)); k += sizeof(spt_camera);
            const uint32_t words[10] = {w, h, row_begin, row_count, rb_log2, rb_stride, rb_mask, samps, c->variant, (uint32_t)blocks};
            std::memcpy(k, words, sizeof words); k += sizeof words;
            std::memcpy(k, &c->scene_gen, sizeof(uint64_t)); k += sizeof(uint64_t);
            std::memcpy(k, &seed, sizeof(uint64_t));
        }
        if (c->order_pending) { SPT_HIP(c, hipStreamWaitEvent(st, c->ev_order, 0)); c->order_pending = false; }
        // (not for the viewer's frames of a few samples per cell: a chunk's time is then the luck of 64 single paths, and the order kernel
        // between two frames costs the frames in flight more than it gains; bit 13 means something else to the grid kernel only)
        // (... and not beyond 4 Mi chunks -- 48 MB of tables; a single band of config 4's size is 1 Mi --: the tail the order removes is a
        // fixed few milliseconds, nothing of a launch that long)
        const bool have_order = c->order_valid && key == c->order_key;
        const bool repeats = key == c->last_pool_key;
        if (!(c->variant & 0x2000u) && !(flags & SPT_FLAG_ONE_SHOT) && samps >= 16u && nchunks <= (4u << 20) && (have_order || repeats)) {
            if (nchunks > c->chunk_cap) {
                if (c->d_chunk_tables) (void)hipFree(c->d_chunk_tables);
                c->d_chunk_tables = nullptr; c->chunk_cap = 0; c->order_valid = false;
                SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_chunk_tables), ((size_t)nchunks * 3 + 512) * sizeof(uint32_t)));
                c->chunk_cap = nchunks;
            }
            uint32_t* const d_order = c->d_chunk_tables;
            uint32_t* const d_clock = c->d_chunk_tables + c->chunk_cap;
            P.chunk_order = (have_order && c->order_valid) ? d_order : nullptr;     // (order_valid: the tables may just have been re-allocated)
            P.chunk_clock = d_clock;
            P.nchunks = nchunks;
            SPT_HIP(c, hipMemsetAsync(d_clock + nchunks, 0, (size_t)nchunks * sizeof(uint32_t), st));
        }
        SPT_HIP(c, hipMemsetAsync(c->d_queue, 0, 256, st));
        SPT_HIP(c, hipEventRecord(c->ev_start, st));
        SPT_HIP(c, spt_pool_launch(&P, (uint32_t)blocks, pool, st));
        SPT_HIP(c, hipEventRecord(c->ev_mid, st));
        SPT_HIP(c, spt_k_finalize(c->d_cells, static_cast<float*>(d_out_rgb), (uint32_t)npix, scale, (flags & SPT_FLAG_NORMALISE) ? 1 : 0, nb, st));
        SPT_HIP(c, hipEventRecord(c->ev_stop, st));
        if (P.chunk_clock) {                                         // (after ev_stop: not part of the frame's device time, overlaps the caller's next step)
            SPT_HIP(c, spt_pool_chunk_order(P.chunk_clock, nchunks, (uint32_t)ntasks, c->d_chunk_tables, c->d_chunk_tables + 3 * c->chunk_cap, st));
            SPT_HIP(c, hipEventRecord(c->ev_order, st));
            c->order_pending = true;
            c->order_key = key;
            c->order_valid = true;
            c->last_nchunks = nchunks;
        } else {
            c->last_nchunks = 0;
        }
        c->last_pool_key.swap(key);
        c->pending = true;
        c->last_was_pool = true;
        c->last_kernel = 1;
        c->last = spt_stats{};
        c->last.samples = npix * 4ull * samps;
        c->last.grid_blocks = (uint32_t)blocks;
        c->last.block_threads = 256;
        return 0;
    }
    c->last_was_pool = false;
    c->last_kernel = 0;

    // launch geometry: a persistent grid that fills the chip; the task queue makes any size correct
    const int mat_lds = (c->n <= 256) ? 1 : 0;
    const int big_block = (c->variant & 0x200u) ? 512 : 256;   // A/B on the box: 256 is faster once the LDS reads are prefetched
    const size_t lds = spt_k_lds_bytes(P.n_pad, mat_lds, big_block);
    const int threads = spt_k_block_threads_for(mat_lds, big_block);
    uint32_t per_cu = c->blocks_per_cu;
    if (per_cu == 0) {
        const uint32_t by_lds = (uint32_t)((160u * 1024u) / lds);
        per_cu = by_lds < 1 ? 1 : (by_lds > 8 ? 8 : by_lds);
    }
    uint64_t blocks = (uint64_t)c->cu_count * per_cu;
    const uint64_t needed = (ntasks + threads - 1) / threads;
    if (blocks > needed) blocks = needed;
    if (blocks < 1) blocks = 1;

    {
        const size_t need_stack = spt_k_stack_floats((uint32_t)blocks, threads);
        if (need_stack > c->stack_cap) {
            if (c->d_stack) (void)hipFree(c->d_stack);
            c->d_stack = nullptr; c->stack_cap = 0;
            SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_stack), need_stack * sizeof(float)));
            c->stack_cap = need_stack;
        }
        P.stack = c->d_stack;
    }
    SPT_HIP(c, hipMemsetAsync(c->d_queue, 0, 256, st));
    SPT_HIP(c, hipEventRecord(c->ev_start, st));
    SPT_HIP(c, spt_k_launch(&P, (uint32_t)blocks, mat_lds, (c->needs_guard || !(cam_big <= 1e15f)) ? 1 : 0, (c->variant & 0x100u) ? 1 : 0, 1, big_block, st));
    SPT_HIP(c, hipEventRecord(c->ev_mid, st));
    SPT_HIP(c, spt_k_finalize(c->d_cells, static_cast<float*>(d_out_rgb), (uint32_t)npix, scale, (flags & SPT_FLAG_NORMALISE) ? 1 : 0, nb, st));
    SPT_HIP(c, hipEventRecord(c->ev_stop, st));
    c->pending = true;
    c->last = spt_stats{};
    c->last.samples = npix * 4ull * samps;
    c->last.grid_blocks = (uint32_t)blocks;
    c->last.block_threads = (uint32_t)threads;
    return 0;
}

int spt_sync(spt_ctx* c, spt_stats* stats)
{
    if (!c) return 1;
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->pending) {
        SPT_HIP(c, hipEventSynchronize(c->ev_stop));
        float ms = 0.f, fms = 0.f;
        SPT_HIP(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_mid));
        SPT_HIP(c, hipEventElapsedTime(&fms, c->ev_mid, c->ev_stop));
        c->last.finalize_ms = fms;
        unsigned long long ctr[2] = {0, 0};
        SPT_HIP(c, hipMemcpy(ctr, c->d_counters, sizeof ctr, hipMemcpyDeviceToHost));
        c->last.kernel_ms = ms;
        c->last.bounces = ctr[0];
        c->last.max_depth_kills = ctr[1];
        if (c->mesh_scene && (c->last_kernel == 2 || c->last_kernel == 6 || c->last_kernel == 7) && c->last.samples)
            c->mesh_ratio = (float)((double)c->last.bounces / (double)c->last.samples);
        if (c->variant & 0x100u) SPT_HIP(c, hipMemcpy(c->diag, c->d_counters + 2, sizeof c->diag, hipMemcpyDeviceToHost));
        c->pending = false;
        if (c->last_was_pool || c->last_kernel == 4 || c->last_kernel == 5) {
            SPT_HIP(c, hipMemcpy(c->pool_stats, c->d_counters + 2, sizeof c->pool_stats, hipMemcpyDeviceToHost));
            if (c->pool_stats[6] != 0)
                return c->fail("spt_sync: %llu waves hit the kernel watchdog; the image is incomplete", c->pool_stats[6]);
        }
    }
    if (stats) *stats = c->last;
    return 0;
}

int spt_render(spt_ctx* c, const spt_camera* cam, uint32_t w, uint32_t h, uint32_t samps, uint64_t seed,
               uint32_t flags, float* out_rgb, spt_stats* stats)
{
    if (!c) return 1;
    if (!out_rgb) return c->fail("spt_render: out_rgb is NULL");
    const auto t0 = std::chrono::steady_clock::now();
    SPT_HIP(c, hipSetDevice(c->device));
    const size_t nfl = (size_t)w * h * 3;
    if (nfl == 0) return c->fail("spt_render: empty image");
    if (nfl > c->out_cap) {
        if (c->pending) SPT_HIP(c, hipEventSynchronize(c->ev_stop));
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_out = nullptr; c->out_cap = 0;
        SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_out), nfl * sizeof(float)));
        c->out_cap = nfl;
    }
    if (int rc = spt_render_rows_device(c, cam, w, h, 0, h, samps, seed, flags, c->d_out, nullptr)) return rc;
    SPT_HIP(c, hipMemcpyAsync(out_rgb, c->d_out, nfl * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    SPT_HIP(c, hipStreamSynchronize(c->stream));
    if (int rc = spt_sync(c, nullptr)) return rc;
    c->last.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = c->last;
    return 0;
}

// accumBuffer += outImage of the render thread (smallpt.cpp:924-937), device-resident; clear != 0 restarts the
// accumulation (needClearBuffer, :931-933).  Both pointers: n floats on this context's device, 16-byte aligned.
int spt_accumulate_device(spt_ctx* c, void* d_accum, const void* d_frame, uint64_t n, int clear, void* hip_stream)
{
    if (!c) return 1;
    if (!d_accum || !d_frame || !n) return c->fail("spt_accumulate_device: bad argument");
    if ((reinterpret_cast<uintptr_t>(d_accum) | reinterpret_cast<uintptr_t>(d_frame)) & 15u)
        return c->fail("spt_accumulate_device: buffers must be 16-byte aligned");
    SPT_HIP(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    SPT_HIP(c, spt_k_accumulate(static_cast<float*>(d_accum), static_cast<const float*>(d_frame), (size_t)n, clear, st));
    return 0;
}

// ---- render-thread frame loop with the accumulation buffer in HBM (smallpt.cpp:881-883,895-942,955-959) ----
int spt_progressive_end(spt_ctx* c)
{
    if (!c) return 1;
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->stream) SPT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->pending) SPT_HIP(c, hipEventSynchronize(c->ev_stop));
    if (c->acc_recorded) SPT_HIP(c, hipEventSynchronize(c->ev_acc));   // accumulations other lanes still have in flight
    if (c->d_accum) (void)hipFree(c->d_accum);
    if (c->d_frame) (void)hipFree(c->d_frame);
    c->d_accum = c->d_frame = nullptr;
    c->prog_w = c->prog_h = 0;
    c->acc_recorded = false;
    c->frame_in_flight = false;
    c->lanes_attached = 0;
    if (c->attached_to) {                                           // a lane: its owner's live count (lanes end before their owner does)
        if (c->attached_to->lanes_attached) --c->attached_to->lanes_attached;
        c->attached_to = nullptr;
    }
    return 0;
}

int spt_progressive_begin(spt_ctx* c, uint32_t w, uint32_t h)
{
    if (!c) return 1;
    if (w == 0 || h == 0) return c->fail("spt_progressive_begin: empty image");
    if (int rc = spt_progressive_end(c)) return rc;
    const size_t bytes = (size_t)w * h * 3 * sizeof(float);
    SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_accum), bytes));
    SPT_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_frame), bytes));
    SPT_HIP(c, hipMemsetAsync(c->d_accum, 0, bytes, c->stream));       // accumBuffer.resize(w*h, make_float3(0,0,0)), :882
    if (!c->ev_acc) SPT_HIP(c, hipEventCreateWithFlags(&c->ev_acc, hipEventDisableTiming));
    SPT_HIP(c, hipEventRecord(c->ev_acc, c->stream));                  // later accumulations (any lane) run behind the clearing
    c->acc_recorded = true;
    c->prog_w = w; c->prog_h = h;
    return 0;
}

int spt_progressive_attach(spt_ctx* lane, spt_ctx* owner)
{
    if (!lane || !owner) return 1;
    if (lane == owner) return 0;
    if (!owner->d_accum) return lane->fail("spt_progressive_attach: call spt_progressive_begin on the owner first");
    if (lane->device != owner->device) return lane->fail("spt_progressive_attach: lane and owner are on different devices");
    if (int rc = spt_progressive_end(lane)) return rc;
    SPT_HIP(lane, hipSetDevice(lane->device));
    // a stream of another priority than the owner's (created at the default priority 0 by spt_create); further lanes take the
    // remaining levels in turn (gfx950: -1, 0, 1), so that as few frames in flight as possible share a hardware queue
    int lo = 0, hi = 0;
    SPT_HIP(lane, hipDeviceGetStreamPriorityRange(&lo, &hi));          // lo = numerically largest = lowest priority
    std::vector<int> levels;
    for (int p = hi; p <= lo; ++p) if (p != 0) levels.push_back(p);
    for (size_t a = 0, b = levels.size(); a + 1 < b; a += 2, --b) std::swap(levels[a + 1], levels[b - 1]);   // highest, lowest, second highest, ...
    if (levels.empty()) levels.push_back(0);
    const int prio = levels[owner->lanes_attached++ % levels.size()];
    if (lane->stream) { (void)hipStreamSynchronize(lane->stream); (void)hipStreamDestroy(lane->stream); lane->stream = nullptr; }
    SPT_HIP(lane, hipStreamCreateWithPriority(&lane->stream, hipStreamNonBlocking, prio));
    SPT_HIP(lane, hipMalloc(reinterpret_cast<void**>(&lane->d_frame), (size_t)owner->prog_w * owner->prog_h * 3 * sizeof(float)));
    lane->prog_w = owner->prog_w; lane->prog_h = owner->prog_h;
    lane->attached_to = owner;
    return 0;
}

int spt_progressive_frame_async(spt_ctx* c, spt_ctx* owner, const spt_camera* cam, uint32_t samps, uint64_t seed, int clear)
{
    if (!c || !owner) return 1;
    if (!owner->d_accum) return c->fail("spt_progressive_frame_async: call spt_progressive_begin on the owner first");
    if (!c->d_frame || c->prog_w != owner->prog_w || c->prog_h != owner->prog_h)
        return c->fail("spt_progressive_frame_async: call spt_progressive_attach(lane, owner) first");
    if (c->frame_in_flight) return c->fail("spt_progressive_frame_async: the lane's previous frame has not been waited for");
    // a lane renders ITS context's scene into the owner's accumBuffer: the caller keeps the scenes equal; what can be told apart cheaply is
    if (c != owner && (c->mesh_scene != owner->mesh_scene || (!c->mesh_scene && c->n != owner->n) || (c->mesh_scene && (c->ntris != owner->ntris || c->ninst != owner->ninst))))
        return c->fail("spt_progressive_frame_async: the lane's scene differs from the owner's (kind or size); set the owner's scene on every lane");
    // :922 the frame is the UN-NORMALISED sum of Renderer::render, on the lane's stream
    c->frames_in_flight_hint = owner->lanes_attached + 1u;           // the owner and its lanes each keep a frame in flight
    const int rrc = spt_render_rows_device(c, cam, c->prog_w, c->prog_h, 0, c->prog_h, samps, seed, 0u, c->d_frame, nullptr);
    c->frames_in_flight_hint = 1u;
    if (rrc) return rrc;
    // :927-937 accumBuffer (clear ? = : +=) outImage, behind the previous accumulation whichever lane issued it
    if (owner->acc_recorded) SPT_HIP(c, hipStreamWaitEvent(c->stream, owner->ev_acc, 0));
    SPT_HIP(c, spt_k_accumulate(owner->d_accum, c->d_frame, (size_t)c->prog_w * c->prog_h * 3, clear, c->stream));
    SPT_HIP(c, hipEventRecord(owner->ev_acc, c->stream));
    owner->acc_recorded = true;
    c->frame_in_flight = true;
    return 0;
}

int spt_progressive_wait(spt_ctx* c, spt_stats* stats)
{
    if (!c) return 1;
    SPT_HIP(c, hipSetDevice(c->device));
    SPT_HIP(c, hipStreamSynchronize(c->stream));
    c->frame_in_flight = false;
    return spt_sync(c, stats);
}

int spt_progressive_frame(spt_ctx* c, const spt_camera* cam, uint32_t samps, uint64_t seed, int clear, spt_stats* stats)
{
    if (!c) return 1;
    if (!c->d_accum) return c->fail("spt_progressive_frame: call spt_progressive_begin first");
    if (int rc = spt_progressive_frame_async(c, c, cam, samps, seed, clear)) return rc;
    return spt_progressive_wait(c, stats);
}

int spt_progressive_snapshot(spt_ctx* c, float* out_rgb)
{
    if (!c) return 1;
    if (!c->d_accum || !out_rgb) return c->fail("spt_progressive_snapshot: no accumulation buffer or out_rgb is NULL");
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->acc_recorded) SPT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_acc, 0));   // every accumulation issued so far, any lane
    SPT_HIP(c, hipMemcpyAsync(out_rgb, c->d_accum, (size_t)c->prog_w * c->prog_h * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    SPT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// Test hook (spt_internal.h): the chunk order the last pool launch left for the next launch of its view.
int spt_chunk_order_snapshot(spt_ctx* c, uint32_t* order, uint32_t cap, uint32_t* nchunks)
{
    if (!c || !nchunks) return 1;
    *nchunks = 0;
    if (!c->order_valid || !c->last_was_pool || c->last_nchunks == 0) return 0;
    if (!order || cap < c->last_nchunks) return c->fail("spt_chunk_order_snapshot: room for %u words, the order has %u", cap, c->last_nchunks);
    SPT_HIP(c, hipSetDevice(c->device));
    if (c->order_pending) SPT_HIP(c, hipEventSynchronize(c->ev_order));
    SPT_HIP(c, hipMemcpy(order, c->d_chunk_tables, (size_t)c->last_nchunks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *nchunks = c->last_nchunks;
    return 0;
}

// Diagnostic (tuning variant bit 8): per-phase wave-time sums [0..7], iterations, lane counts of the last launch.
int spt_diag(spt_ctx* c, unsigned long long* out24)
{
    if (!c || !out24) return 1;
    if (c->last_was_pool || c->last_kernel == 4 || c->last_kernel == 5) {
        std::memcpy(out24, c->pool_stats, sizeof c->pool_stats);
        return 0;
    }
    std::memcpy(out24, c->diag, sizeof c->diag);
    return 0;
}

int spt_set_watchdog(spt_ctx* c, double seconds)
{
    if (!c) return 1;
    c->watchdog_ticks = seconds > 0 ? (unsigned long long)(seconds * 2.4e9) : 0ull;   // s_memtime ticks are shader cycles (<= 2.4 GHz)
    return 0;
}

int spt_last_kernel(spt_ctx* c) { return c ? c->last_kernel : -1; }

// Numerics self-test: runs device helper `op` (0 sqrt_fix, 2 sqrt_exact, 3 rcp_exact, 10 sqrt_rsq,
// 4 double division by w, 5/6 sin/cos(2*pi*x), 7 rng_draw(bits(x))) over n host floats.
int spt_selftest_math(spt_ctx* c, int op, const float* in, float* out, uint32_t n, uint32_t w)
{
    if (!c) return 1;
    if (!in || !out || !n || !w) return c->fail("spt_selftest_math: bad argument");
    SPT_HIP(c, hipSetDevice(c->device));
    float *d_in = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_in), (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = spt_k_selftest(op, d_in, d_out, n, w, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) return c->fail("spt_selftest_math: %s", hipGetErrorString(e));
    return 0;
}

// Exhaustive device checks of sqrt_rsq (op 0; 1 = negative control) and rcp_exact<false> (op 2; 3 = negative control) for the bit patterns [first, first + count).
int spt_selftest_range(spt_ctx* c, int op, uint32_t first, uint32_t count, uint64_t* mismatches, uint32_t* first_bad)
{
    if (!c) return 1;
    if (!mismatches || !first_bad) return c->fail("spt_selftest_range: NULL argument");
    SPT_HIP(c, hipSetDevice(c->device));
    unsigned long long* d_m = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_m), 16);
    const unsigned long long init[2] = {0ull, 0xFFFFFFFFull};
    if (e == hipSuccess) e = hipMemcpy(d_m, init, 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = spt_k_selftest_range(op, first, count, d_m, reinterpret_cast<uint32_t*>(d_m + 1), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    unsigned long long out[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(out, d_m, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d_m);
    if (e != hipSuccess) return c->fail("spt_selftest_range: %s", hipGetErrorString(e));
    *mismatches = out[0];
    *first_bad = (uint32_t)out[1];
    return 0;
}

// smallpt.cpp:52
int spt_to_int(float x)
{
    const float cl = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);
    return (int)(std::pow((double)cl, 1 / 2.2) * 255 + .5);
}

// flipY (smallpt.cpp:125-134) + writeImage (smallpt.cpp:136-142); unlike the reference the file is closed.
int spt_write_ppm(const char* path, const float* rgb, uint32_t w, uint32_t h)
{
    if (!path || !rgb || !w || !h) return 1;
    FILE* f = std::fopen(path, "w");
    if (!f) return 1;
    std::fprintf(f, "P3\n%u %u\n%d\n", w, h, 255);
    for (uint32_t r = 0; r < h; ++r) {
        const float* row = rgb + (size_t)(h - 1 - r) * w * 3;
        for (uint32_t x = 0; x < w; ++x)
            std::fprintf(f, "%d %d %d ", spt_to_int(row[3 * x]), spt_to_int(row[3 * x + 1]), spt_to_int(row[3 * x + 2]));
    }
    return std::fclose(f) == 0 ? 0 : 1;
}

}  // extern "C"
