/*
 * smallpt_mi355x.h -- C-ABI of the MI355X-native path tracer (libsmallpt_mi355x.so).
 *
 * This is the drop-in boundary for the reference's render hot path.  Each entry point names the
 * reference interface it replaces (paths relative to the reference tree):
 *
 *   reference seam                                             replaced by
 *   ---------------------------------------------------------  ---------------------------------
 *   Sphere spheres[] / Material (smallpt.cpp:31-50,            spt_set_scene()
 *     scene.h:66-92) + Intersector::addTriangleMesh/build
 *     (smallpt.cpp:489-530: "upload the scene to the device")
 *   cam / cx / cy of cpuRender (smallpt.cpp:277-279)            spt_camera_smallpt()
 *   int cpuRender(argc, argv) (smallpt.cpp:269-379): the        spt_render()           (host image)
 *     offline render; and Vector<float3> Renderer::render(..)   spt_render_rows_device() (row band,
 *     (smallpt.cpp:679-680,692-814), sole caller :922             device-resident, async)
 *   Intersector::traceRays + shadePaths per bounce              inside the kernels; the triangle seam itself
 *     (smallpt.cpp:553-587,154-267)                               (addTriangleMesh/build/traceRays, :427-473) is
 *                                                                 spt_set_meshes() / spt_trace_rays()
 *   accumBuffer += outImage under accumBufferMutex and the      spt_progressive_begin / _frame / _snapshot / _end
 *     GL thread's copy of it (smallpt.cpp:881-883,924-940,       (accumulation buffer resident in HBM)
 *     955-959)
 *   "Elapsed time" stderr line (smallpt.cpp:371-373,809-811)    spt_stats
 *   CHK_PRIME / rtpContextGetLastErrorString                    int status + spt_last_error()
 *     (smallpt.cpp:381-393)
 *
 * Conventions kept from the reference: the image is row-major w*h packed float3 (12 B/pixel), row 0
 * is the BOTTOM row (camera cy is +y; flipY happens only before the PPM, smallpt.cpp:125-134,375),
 * spp = 4 * samps_per_cell (2x2 jitter cells, smallpt.cpp:285-286).  With SPT_FLAG_NORMALISE the
 * image is divided by spp like cpuRender (:358-361); without it the un-normalised SUM is returned
 * like Renderer::render (:790,:813) for a caller that accumulates frames (:924-936) and weights
 * at display time (:957-962, glutils.cpp:230-256).
 *
 * All functions return 0 on success, non-zero on error (message via spt_last_error).  No C++
 * types, no exceptions cross this boundary.  A context is bound to one HIP device and is not
 * thread-safe; use one context per device (one process per GPU under torch.distributed, or one
 * host thread per device).
 */
#ifndef SMALLPT_MI355X_H
#define SMALLPT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPT_API_VERSION 1

typedef struct spt_ctx spt_ctx;

/* Refl_t, scene.h:64 */
enum { SPT_DIFF = 0, SPT_SPEC = 1, SPT_REFR = 2 };

/* 48-byte POD; field order = Sphere constructor, scene.h:91 (radius, center, emission, color, refl),
 * regrouped so that {center, radius} is one 16-byte load. */
typedef struct spt_sphere {
    float    center[3];
    float    radius;
    float    emission[3];
    float    color[3];
    int32_t  refl;
    uint32_t pad;
} spt_sphere;

/* Camera: d = cx*ax + cy*ay + dir; ray = (origin + d*push, normalize(d)).
 * sampler = SPT_SAMPLER_SMALLPT: (ax, ay) from the 2x2-cell tent filter of cpuRender (smallpt.cpp:327-332,
 *   evaluated in double like the reference); the smallpt camera of :277-279 has push = 140 (:333).
 * sampler = SPT_SAMPLER_PINHOLE: (ax, ay) = clip-space position of Renderer::render's box-in-cell sample
 *   (smallpt.cpp:745-760) fed to sampleRay (:626-641); cx, cy = columns 0, 1 of Camera::localToWorld,
 *   dir = column 2 * nearPlaneDistance, origin = column 3, push = 0 (:607-641). */
typedef struct spt_camera {
    float origin[3];
    float dir[3];
    float cx[3];
    float cy[3];
    float push;
    uint32_t sampler;
} spt_camera;
enum { SPT_SAMPLER_SMALLPT = 0, SPT_SAMPLER_PINHOLE = 1 };

typedef struct spt_stats {
    uint64_t samples;        /* camera paths traced (= rows*w*spp)                          */
    uint64_t bounces;        /* closest-hit queries executed (= intersectGlobalSpheres calls) */
    uint64_t max_depth_kills;/* paths cut by the SPT_MAX_DEPTH guard                          */
    float    kernel_ms;      /* HIP-event time of the path-tracing megakernel on its stream   */
    float    finalize_ms;    /* HIP-event time of the cell-fold/normalise/store kernel        */
    float    total_ms;       /* host wall time of the call (spt_render only; incl. D2H copy)  */
    uint32_t grid_blocks;    /* launch geometry actually used                                 */
    uint32_t block_threads;
    uint32_t pad;
} spt_stats;

#define SPT_FLAG_NORMALISE 1u  /* divide by spp (cpuRender); otherwise return the raw sum (render()) */
#define SPT_FLAG_ONE_SHOT  2u  /* scheduling only: this launch neither uses nor records a dispatch order (see spt_render_rows_device): a caller
                                * that will not render the view again saves the clock stores, three small kernels and the order tables */
#define SPT_MAX_DEPTH      4096u
#define SPT_MAX_SPHERES    4096u  /* the exhaustive kernels stage the table in LDS: 16 B geometry per sphere */
#define SPT_MAX_SPHERES_ACCEL 1048576u  /* through a structure (spt_set_sphere_accel: the grid up to about 9 000 spheres, the hierarchy beyond) */

/* Creates a context on HIP device `device_id` (its own non-blocking stream, events, scratch). */
int  spt_create(int device_id, spt_ctx** out);
void spt_destroy(spt_ctx* ctx);
const char* spt_last_error(const spt_ctx* ctx);  /* ctx may be NULL: last error of spt_create */
int  spt_api_version(void);
int  spt_device_count(void);

/* Uploads the sphere table (replaces the global spheres[] + materials vector, smallpt.cpp:31-50,288-290).  Up to SPT_MAX_SPHERES
 * in every mode; up to SPT_MAX_SPHERES_ACCEL in the default mode and SPT_ACCEL_BVH (spt_set_sphere_accel), where the table sits
 * behind a structure, provided every radius is >= 2^-30 and every coordinate within 1e15 (else the call fails and the previous
 * scene stays current). */
int  spt_set_scene(spt_ctx* ctx, const spt_sphere* spheres, uint32_t n);

/* ---- triangle meshes: the reference's Intersector seam (smallpt.cpp:427-473 CPUIntersector, :475-603 OptixIntersector) ----
 * TriMesh (scene.h:6-15), Material (scene.h:66-73), Ray (scene.h:58-62), Hit (scene.h:31-43; dist = 1e20 on a miss). */
typedef struct spt_mesh {
    const float*    positions;   /* positionBuffer: nverts x 3 */
    const float*    normals;     /* normalBuffer:   nverts x 3 */
    const uint32_t* indices;     /* indexBuffer:    ntris x 3  */
    uint32_t nverts, ntris;
} spt_mesh;
typedef struct spt_material { float emission[3]; float color[3]; int32_t refl; uint32_t pad; } spt_material;
typedef struct spt_ray { float o[3]; float d[3]; } spt_ray;
typedef struct spt_hit { float dist; uint32_t instId; uint32_t triId; float x[3]; float n[3]; float uv[2]; } spt_hit;

/* Intersector::addTriangleMesh for every mesh + build() (smallpt.cpp:437-447 / :489-530): uploads the instances
 * (materials[i] belongs to mesh i, :170) and makes the mesh scene current: spt_render* then trace it with the reference's
 * triangle arithmetic (triIntersect scene.cpp:52-70, intersect :95-116, makeHit :73-93; hit.n is the interpolated,
 * un-normalised vertex normal).  spt_set_scene switches back to spheres. */
int  spt_set_meshes(spt_ctx* ctx, const spt_mesh* meshes, uint32_t nmesh, const spt_material* materials);

/* How the closest hit of a mesh scene is found.  SPT_ACCEL_BVH and SPT_ACCEL_EXHAUSTIVE return the same Hit for EVERY ray (since round 4),
 * and the default, SPT_ACCEL_AUTO, picks between these two per launch: the hierarchy, unless the scene has fewer than 256 triangles or --
 * renders only -- fewer than 8192 and more than 15 % of its last launch's closest-hit queries were bounce rays (those walk the plane tree
 * below; under that size the exhaustive loop is then the faster of the two).  Results never depend on the choice.
 *   SPT_ACCEL_BVH: the role of the OptiX Prime model/query of the reference's GPU intersector
 *     (smallpt.cpp:475-603, the intersector the reference actually runs, :605): structures built over the triangles when the
 *     meshes are set.  The triangles they reach go through the same triIntersect arithmetic and the same selection (smallest
 *     dist > 0, lowest (instance, triangle) among equal dist) as the exhaustive loop, and they provably reach every triangle whose
 *     report beats or ties the answer (csrc/spt_tribvh.h): a bounding-volume hierarchy whose boxes are inflated per ray and per
 *     node by the error bound of a report (it knows a cone of the normals below each node); and, because triIntersect has no
 *     determinant cut-off (scene.cpp:62) and reports rounding noise when dot(rd, cross(e1, e2)) is zero to rounding -- a "hit" no
 *     bounding volume contains --, a tree over the triangles' PLANES that finds the triangles in whose plane the ray lies, and a
 *     table (a tree beyond 16 384) of the long edges' LINES of thin triangles (the needles makeSphereTriMesh puts at the poles:
 *     their normal is noise for every ray) that finds the needles whose supporting line the ray's line crosses, wherever along
 *     it.  Rounds 2 and 3 documented those rays as exceptions (18 of 668 000 test rays); tests/test_meshes.py now requires 0
 *     differences on 700 000 random and adversarial rays, the CPU harness tests/sanitize/tribvh_main.cpp runs the same walk
 *     functions against the exhaustive loop.  A render launch lists the triangles in whose plane the camera's origin lies once
 *     (the lines of all its rays of depth 0 pass through that point) and those rays test the list instead of walking the plane
 *     tree.  Cost, shipped scene (8192 triangles), 1280 x 720 x 4 spp: 1.4 ms per pinhole frame, 1.5 ms with the smallpt camera,
 *     against 25-29 ms through the exhaustive loop; spt_trace_rays_device 0.41 Grays/s against 0.125.  A ray that starts hundreds of scene sizes away degrades
 *     towards the exhaustive loop's cost (the error bound grows with the distance), never in result.
 *   SPT_ACCEL_EXHAUSTIVE: every triangle of every instance is tested, as CPUIntersector::intersect does (smallpt.cpp:443-458 over
 *     scene.cpp:95-116): the parity anchor.
 *   SPT_ACCEL_BVH_FAST (opt-in): the bounding-volume hierarchy alone, as in rounds 2-3 -- 1.05 ms for the pinhole frame above.  It
 *     returns the exhaustive Hit whenever the winning triangle's padded box is crossed within the current nearest distance; a ray
 *     lying (to ~1e-7 rad) in a triangle's plane, or crossing a needle's supporting line, may lose the noise "hit" the reference's
 *     arithmetic reports there.  Rendered images have never met the condition (tests compare them), constructed rays do.
 * Applies to spt_trace_rays and to spt_render* / spt_progressive_* of a mesh scene; may be changed at any time. */
#define SPT_ACCEL_EXHAUSTIVE 0
#define SPT_ACCEL_BVH        1
#define SPT_ACCEL_BVH_FAST   3   /* mesh scenes only: the spatial hierarchy alone (rounds 2-3), see above */
#define SPT_ACCEL_AUTO       4   /* mesh scenes only, the default: SPT_ACCEL_BVH or SPT_ACCEL_EXHAUSTIVE, whichever is expected to be faster */
int  spt_set_mesh_accel(spt_ctx* ctx, int accel);
/* How the closest hit of a SPHERE table larger than the 24 the material-sorted kernel unrolls is found (smallpt.cpp:54-70 loops
 * over all of them).  Every mode returns the exhaustive loop's hit for every ray -- same intersectAnalytic arithmetic
 * (scene.cpp:129-140) on the spheres it tests, same selection (smallest t > eps, lowest index among equal t) -- and the
 * structures are exhaustive-equivalent BY CONSTRUCTION: intersectAnalytic divides by nothing, so the error of a reported hit is
 * bounded (101 u (|c - o|^2 + r^2) + | |d|^2 - 1 | t^2 in |p - c|^2 - r^2) and no sphere is skipped unless it misses the ray by
 * more than that bound (csrc/spt_grid.h).
 *   SPT_ACCEL_GRID (default): a uniform grid held in LDS; spheres more than 16 x the median radius (walls, lights) are tested
 *     for every ray, rays outside the error bound's precondition take the exhaustive loop.  Scenes that do not qualify (<= 24
 *     spheres, degenerate radii / coordinates, tables beyond the LDS: about 9 000 spheres) go through the hierarchy below from
 *     1024 spheres on and through the exhaustive kernels otherwise.
 *   SPT_ACCEL_BVH: a bounding-volume hierarchy with per-ray inflated boxes (round 2).
 *   SPT_ACCEL_EXHAUSTIVE: every sphere for every ray. */
#define SPT_ACCEL_GRID       2
int  spt_set_sphere_accel(spt_ctx* ctx, int accel);
/* Vector<Hit> Intersector::traceRays(const PathContrib*, size_t) (smallpt.cpp:460-470, :553-587): closest hit of n rays
 * against the current mesh scene; host buffers in and out like the reference's RTP_BUFFER_TYPE_HOST queries (:571-575). */
int  spt_trace_rays(spt_ctx* ctx, const spt_ray* rays, uint64_t n, spt_hit* hits);
/* The same query on DEVICE buffers of this context's device (n spt_ray in, n spt_hit out; what OptiX Prime's RTP_BUFFER_TYPE_CUDA_LINEAR
 * buffers are to the reference's intersector, smallpt.cpp:571-575): enqueued on `hip_stream` (NULL = the context's stream), returns
 * without waiting.  No bytes cross the host link. */
int  spt_trace_rays_device(spt_ctx* ctx, const void* d_rays, uint64_t n, void* d_hits, void* hip_stream);
/* Host-only helper: makeSphereTriMesh(origin, radius, subdivLongitude) (scene.cpp:3-48): fills (L+1)(2L+1) positions and
 * normals and 4L^2 triangles (L = subdiv_longitude, default 32 at scene.h:17); returns the triangle count. */
uint32_t spt_make_sphere_trimesh(const float origin[3], float radius, uint32_t subdiv_longitude,
                                 float* positions, float* normals, uint32_t* indices);

/* Host-only helper: the camera constants of cpuRender for a w x h image (smallpt.cpp:277-279). */
int  spt_camera_smallpt(uint32_t w, uint32_t h, spt_camera* out);

/* Host-only helper: the pinhole Camera of the interactive driver, Camera{vx, vy, vz, org, nearPlaneDistance}
 * (smallpt.cpp:607-624; main() uses vx=(1,0,0), vz=(0,0,-1), vy=normalize(cross(vx,vz)), org=(0,-1,0), near=1,
 * :885-899).  Selects SPT_SAMPLER_PINHOLE. */
int  spt_camera_pinhole(const float vx[3], const float vy[3], const float vz[3], const float org[3],
                        float near_plane_distance, spt_camera* out);

/* Renders the full w x h image and copies it to out_rgb (host, w*h*3 floats).  Blocking. */
int  spt_render(spt_ctx* ctx, const spt_camera* cam, uint32_t w, uint32_t h,
                uint32_t samps_per_cell, uint64_t seed, uint32_t flags,
                float* out_rgb, spt_stats* stats);

/* Renders rows [row_begin, row_begin+row_count) of the w x h image into d_out_rgb, a DEVICE pointer
 * to row_count*w*3 floats on this context's device.  The launch is enqueued on `hip_stream`
 * (a hipStream_t cast to void*; NULL = the context's own stream) and returns without waiting for it
 * (a context keeps one launch in flight: a call made while the previous launch is still running first waits for it).
 * Pixel/sample RNG keys use the GLOBAL pixel index, so any row partition over any number of GPUs
 * yields the same image.  Call spt_sync() before reading stats.
 * Scheduling only (never the result): for tables of <= 24 spheres and >= 16 samples per cell a context remembers how long each group
 * of sample blocks took in its last launch, and a launch of the same scene, camera, image, band, sample count AND SEED starts the
 * expensive ones first: re-rendering a view is ~4 % shorter at 1024 spp than rendering it the first time (79.4 -> 76.3 ms on config 2).
 * A launch records those times only when it repeats its predecessor (recording costs 0.6 ms of such a launch), so the gain starts with the
 * third identical launch.
 * Another seed of the view runs in the static order like a first launch -- measured, the previous seed's order makes it 1 % SLOWER
 * (profiles/r04_cost_order_seeds.txt; round 3 claimed the gain for any seed without having stepped it).  SPT_FLAG_ONE_SHOT opts a launch out. */
int  spt_render_rows_device(spt_ctx* ctx, const spt_camera* cam, uint32_t w, uint32_t h,
                            uint32_t row_begin, uint32_t row_count,
                            uint32_t samps_per_cell, uint64_t seed, uint32_t flags,
                            void* d_out_rgb, void* hip_stream);

/* Progressive accumulation of the viewer's render thread (smallpt.cpp:924-937), device-resident:
 * d_accum[i] = clear ? d_frame[i] : d_accum[i] + d_frame[i] for n floats (both 16-byte aligned, on this device);
 * enqueued on hip_stream (NULL = the context's stream).  Display weight = 1/(frames*spp) (smallpt.cpp:957). */
int  spt_accumulate_device(spt_ctx* ctx, void* d_accum, const void* d_frame, uint64_t n, int clear, void* hip_stream);

/* The same for a rank of a multi-GPU render whose rows are dealt out round-robin in blocks of `block_rows` rows (a power of
 * two): block t of the image (rows [t*B, (t+1)*B)) belongs to rank t % world.  Contiguous bands of a Cornell-like image
 * differ by up to 1.34x in cost (floor and spheres below, ceiling above); interleaved blocks balance the ranks.  d_out_rgb
 * receives this rank's spt_interleaved_row_count() rows packed in ascending row order. */
uint32_t spt_interleaved_row_count(uint32_t h, uint32_t block_rows, uint32_t world, uint32_t rank);
int  spt_render_interleaved_device(spt_ctx* ctx, const spt_camera* cam, uint32_t w, uint32_t h, uint32_t block_rows,
                                   uint32_t world, uint32_t rank, uint32_t samps_per_cell, uint64_t seed, uint32_t flags,
                                   void* d_out_rgb, void* hip_stream);

/* The render thread's frame loop (smallpt.cpp:895-942) with accumBuffer (:881-883) resident in HBM behind the boundary:
 *   spt_progressive_begin    allocates the w*h*3 accumulation buffer and a frame buffer on the context's device;
 *   spt_progressive_frame    = `outImage = renderer.render(camera, ..., sampleCountPerJitterCell, threadCount, seed)` (:922,
 *                            un-normalised sum) followed by `accumBuffer (clear ? = : +=) outImage` (:927-937); blocking;
 *   spt_progressive_snapshot = `image = accumBuffer` under the mutex (:955-959): copies the accumulation buffer to host
 *                            memory in the layout drawWeightedRGBImage(const float*, w, h, weight[3]) takes (glutils.h:153,
 *                            glutils.cpp:230-256: GL_RGB / GL_FLOAT rows, bottom row first); the caller supplies the weight
 *                            1/(sampleCount*sampleCountPerPixel) of :957;
 *   spt_progressive_end      frees the two buffers. */
int  spt_progressive_begin(spt_ctx* ctx, uint32_t w, uint32_t h);
int  spt_progressive_frame(spt_ctx* ctx, const spt_camera* cam, uint32_t samps_per_cell, uint64_t seed, int clear, spt_stats* stats);
int  spt_progressive_snapshot(spt_ctx* ctx, float* out_rgb);
int  spt_progressive_end(spt_ctx* ctx);
/* The same loop with SEVERAL FRAMES IN FLIGHT.  The reference overlaps its render thread with the GL thread (smallpt.cpp:895-962);
 * on the GPU the end of a 4-spp frame is a handful of long specular chains that leave most of the chip idle, so a host that
 * issues frame k+1 before frame k has drained keeps it busy.  A context renders one frame at a time (its scratch buffers belong
 * to the frame), hence one context per frame in flight:
 *   spt_progressive_attach(lane, owner)   `lane` (another context on the same device, with the same scene) gets its own frame
 *                            buffer of the owner's size and a stream whose priority differs from the owner's (equal-priority
 *                            streams of a process share a hardware queue and would serialise the frames);
 *   spt_progressive_frame_async(lane, owner, cam, samps, seed, clear)   enqueues render + accumulation on the lane's stream and
 *                            returns without waiting.  `lane` may be the owner itself.  The accumulations into the owner's
 *                            accumBuffer run in the order of the calls (chained by events), so accumBuffer is bit-identical
 *                            to the blocking loop's; the lane's previous frame must have been waited for;
 *   spt_progressive_wait(lane, stats)     waits for the lane's frame in flight (render and accumulation);
 *   spt_progressive_snapshot(owner, ...)  waits for every accumulation issued so far, then copies.
 * One host thread drives all lanes of an owner (contexts are not thread-safe). */
int  spt_progressive_attach(spt_ctx* lane, spt_ctx* owner);
int  spt_progressive_frame_async(spt_ctx* lane, spt_ctx* owner, const spt_camera* cam, uint32_t samps_per_cell, uint64_t seed, int clear);
int  spt_progressive_wait(spt_ctx* lane, spt_stats* stats);

/* Waits for the last launch of this context and fills stats (may be NULL). */
int  spt_sync(spt_ctx* ctx, spt_stats* stats);

/* Image output helpers kept from the reference: toInt (smallpt.cpp:52), flipY (:125-134) and the
 * ASCII P3 writer (:136-142).  rgb is w*h*3 floats, row 0 = bottom; the file gets the flipped image. */
int  spt_to_int(float x);
int  spt_write_ppm(const char* path, const float* rgb, uint32_t w, uint32_t h);

#ifdef __cplusplus
}
#endif
#endif /* SMALLPT_MI355X_H */
