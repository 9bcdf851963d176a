/*
 * smallpt_mi355x_multi.h -- multi-GPU front of the C-ABI (libsmallpt_mi355x_multi.so, links RCCL).
 *
 * The reference renders on exactly one device (rtpContextSetCudaDeviceNumbers(context, 1, &device) with device 0,
 * smallpt.cpp:480-481) behind `Vector<float3> Renderer::render(...)` (smallpt.cpp:679-680,692-814; sole caller :922).
 * This entry point keeps that call shape -- one call, one framebuffer -- and spreads it over the GPUs of one node:
 *
 *   * the image rows are dealt out to the devices round-robin in blocks of 16 rows (the reference's own unit of
 *     parallelism is the row, smallpt.cpp:317,736; contiguous bands of a Cornell-like image differ by up to 1.34x in cost,
 *     so SPT_MULTI_CONTIGUOUS -- band g = rows [g*h/G ...) -- is only an option); the RNG is keyed by the GLOBAL pixel
 *     index, so the assembled image is bit-identical for every device count and either partition;
 *   * one host thread + one spt_ctx + one HIP stream per device; every device renders its band with
 *     spt_render_rows_device (include/smallpt_mi355x.h);
 *   * the bands are assembled on the root device (device_ids[0]) by ONE exchange step: every other rank ncclSend()s
 *     its packed rows, the root ncclRecv()s them (all receives fused in one ncclGroupStart/End -- point-to-point over xGMI,
 *     7 links into the root in parallel, no ring, no reduction) and scatters the row blocks into the framebuffer with one
 *     strided device copy per rank; with contiguous bands the receives land straight in the framebuffer's row slices and
 *     the root's own band is rendered in place.  With one device no RCCL communicator is created at all
 *     (unless SPT_MULTI_SELF_EXCHANGE is passed, which routes the root's band through a grouped self send/recv:
 *     a rehearsal of the RCCL path for boxes with a single GPU).
 *
 * Failure: every device finishes (or fails) its rows before any device enqueues its part of the exchange, so a render error on
 * one device (HIP error, kernel watchdog) makes spt_multi_render return non-zero with that device's message -- no peer is left
 * waiting in ncclRecv -- and the object stays usable.
 *
 * All functions return 0 on success; errors via spt_multi_last_error.  Not thread-safe; one call at a time.
 */
#ifndef SMALLPT_MI355X_MULTI_H
#define SMALLPT_MI355X_MULTI_H

#include "smallpt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spt_multi spt_multi;

#define SPT_MULTI_SELF_EXCHANGE 1u  /* create flag: with one device, still send the band through RCCL (self send/recv) */
#define SPT_MULTI_COPY_EXCHANGE 4u  /* create flag: assemble with hipMemcpyPeerAsync pulls by the root instead of RCCL; with this
                                       transport several ranks may share a device (device ids may repeat) */
#define SPT_MULTI_CONTIGUOUS    2u  /* create flag: contiguous row bands instead of round-robin blocks of 16 rows */

typedef struct spt_multi_stats {
    uint64_t samples;          /* whole image */
    uint64_t bounces;
    uint64_t max_depth_kills;
    float    render_ms;        /* slowest device: HIP-event time of its megakernel + store kernel          */
    float    gather_ms;        /* root: HIP-event time of the RCCL exchange on its stream (0 with one device) */
    float    total_ms;         /* host wall time of the call                                               */
    uint32_t ndev;
    uint32_t pad;
} spt_multi_stats;

/* device_ids[0] is the root (the framebuffer is assembled there).  ndev >= 1; ids must be distinct (RCCL transport). */
int  spt_multi_create(const int* device_ids, int ndev, uint32_t flags, spt_multi** out);
void spt_multi_destroy(spt_multi* m);
const char* spt_multi_last_error(const spt_multi* m);   /* m may be NULL: last error of spt_multi_create */
int  spt_multi_device_count(const spt_multi* m);

/* Uploads the sphere table to every device (spt_set_scene). */
int  spt_multi_set_scene(spt_multi* m, const spt_sphere* spheres, uint32_t n);
/* The triangle seam on every device: Intersector::addTriangleMesh + build (smallpt.cpp:437-447, spt_set_meshes) -- the scene the
 * reference's live path renders (smallpt.cpp:818-842, 922) -- and the closest-hit modes (spt_set_mesh_accel, spt_set_sphere_accel). */
int  spt_multi_set_meshes(spt_multi* m, const spt_mesh* meshes, uint32_t nmesh, const spt_material* materials);
int  spt_multi_set_mesh_accel(spt_multi* m, int accel);
int  spt_multi_set_sphere_accel(spt_multi* m, int accel);

/* Row band of rank `rank` of `world` for an image of height h: rows split as evenly as possible, the first h % world
 * ranks get one more row; bands are in rank order = row order. */
void spt_multi_row_band(uint32_t h, uint32_t world, uint32_t rank, uint32_t* row_begin, uint32_t* row_count);

/* Renders the w x h image on all devices and assembles it on the root device.  out_rgb: host buffer of w*h*3 floats
 * (may be NULL: the framebuffer then stays on the root device, see spt_multi_framebuffer).  Same conventions as
 * spt_render (row 0 = bottom, SPT_FLAG_NORMALISE). */
int  spt_multi_render(spt_multi* m, const spt_camera* cam, uint32_t w, uint32_t h,
                      uint32_t samps_per_cell, uint64_t seed, uint32_t flags,
                      float* out_rgb, spt_multi_stats* stats);

/* Device pointer (root device) of the framebuffer assembled by the last spt_multi_render: w*h*3 floats. */
void* spt_multi_framebuffer(spt_multi* m);

/* The viewer's render loop (smallpt.cpp:895-942) over all devices, the counterpart of spt_progressive_* (include/smallpt_mi355x.h):
 *   _begin     allocates accumBuffer (w*h*3 floats, zeroed) on the root device;
 *   _frame     = `outImage = renderer.render(camera, ..., seed)` (:922) on all devices, assembled on the root, followed by
 *                accumBuffer = outImage (clear != 0: the frame after a request, :924-930) or accumBuffer += outImage (:932-937) there;
 *                frames are raw sums (no division by spp), as Renderer::render returns them;
 *   _snapshot  = `image = accumBuffer` under the mutex (:955-959): copies accumBuffer to w*h*3 host floats;
 *   _end       frees accumBuffer.
 * One frame at a time (a frame in flight on every device already fills the node). */
int  spt_multi_progressive_begin(spt_multi* m, uint32_t w, uint32_t h);
int  spt_multi_progressive_frame(spt_multi* m, const spt_camera* cam, uint32_t samps_per_cell, uint64_t seed, int clear, spt_multi_stats* stats);
int  spt_multi_progressive_snapshot(spt_multi* m, float* out_rgb);
int  spt_multi_progressive_end(spt_multi* m);

#ifdef __cplusplus
}
#endif
#endif /* SMALLPT_MI355X_MULTI_H */
