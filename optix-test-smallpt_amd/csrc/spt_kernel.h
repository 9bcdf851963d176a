// spt_kernel.h -- launch parameters shared by the kernel TU (hipcc) and the C-ABI TU.
#pragma once
#include <hip/hip_runtime.h>
#include "spt_tribvh.h"
#include <stdint.h>

#define SPT_K_MAX_DEPTH 4096u

namespace spt {

struct KParams {
    // camera (smallpt.cpp:277-279,333)
    float cam_o[3], cam_d[3], cam_cx[3], cam_cy[3];
    float cam_push;
    uint32_t sampler;        // 0 smallpt tent filter (cpuRender), 1 box-in-cell + pinhole (Renderer::render)
    float inv_wf, inv_hf;    // 1.f/w, 1.f/h (pixelSize, smallpt.cpp:746)
    // image / band
    uint32_t w, h, row_begin, row_count;
    // local row ry of this launch is image row row_begin + (ry >> rb_log2) * rb_stride + (ry & rb_mask): a contiguous band
    // is (0, 1, 0); rows dealt out round-robin in blocks of B = 2^rb_log2 rows to `world` ranks are (log2 B, world * B, B - 1)
    uint32_t rb_log2, rb_stride, rb_mask;
    double inv_w, inv_h;     // RN(1/w), RN(1/h) for the exact double division of smallpt.cpp:331-332
    uint32_t samps;          // samples per jitter cell (spp = 4*samps)
    uint32_t ntasks;         // 4 * row_count * w * nb: one task = one block of a jitter cell's samples (D9)
    uint32_t nb_log2, sb;    // nb = 1 << nb_log2 blocks per cell, sb samples per block (last block may be shorter)
    uint32_t park_threshold; // glass-shading pass runs when this many lanes of a wave wait for it
    // RNG seed hashes (D7), computed on the host once per render
    uint32_t s0, s1;
    // scene tables (device memory)
    uint32_t n, n_pad;
    const float4* geom;      // n x {cx, cy, cz, r*r}
    const float4* mat;       // n x 3: {e.xyz, refl bits}, {color.xyz, pmax}, {color*(1/pmax), 0}
    // outputs
    float4* cells;           // ntasks block sums
    uint32_t* queue;         // task queue head (zeroed before launch)
    unsigned long long* counters;  // [0] bounces, [1] depth-cap kills, [2..16] DIAG phase times / lane counts
                                   // pool kernel: [2..4] batches per class (GEN, DIFF, REFR), [5..7] lanes per class, [8] watchdog hits
    float* stack;                  // pending transmitted children of the glass split: 64-byte records in global memory
                                   // (pool kernel: waves x slots x 3; megakernel / mesh kernel: threads x 3)
    // pool kernel (spt_pool.hip) only
    uint2* slot_state;             // waves x pool slots x {task id, next sample}
    unsigned long long watchdog_ticks;  // pool and grid kernels: s_memtime ticks after which a wave gives up (0 = never)
    // pool kernel: cost-ordered dispatch (spt_pool.hip "chunk order").  The task queue hands out chunks of 64 consecutive task ids;
    // chunk_clock[c] / chunk_clock[nchunks + c] receive the time chunk c was fetched / its last task was completed (s_memtime >> 6,
    // the clock of the one wave that works on it), and chunk_order -- when not null -- is the permutation a previous launch of the
    // same view derived from those: the queue's k-th chunk is chunk_order[k], most expensive first.
    const uint32_t* chunk_order;
    uint32_t* chunk_clock;
    uint32_t nchunks;
};

// Triangle-mesh scene (spt_mesh.hip): the reference's TriMesh instances flattened into device tables
struct MParams {
    const float4* tris;            // ntris x 3: {v0.xyz, n.x} {v1-v0, n.y} {v2-v0, n.z}, n = cross(v1-v0, v2-v0) (scene.cpp:56-60)
    const uint4* tri_index;        // ntris x {i1, i2, i3 (global vertex ids), instance}
    const float4* verts;           // nverts x 2: {position, 0} {normal, 0}
    const uint32_t* inst_first_tri;// first global triangle of every instance
    const float4* mats;            // 3 rows per instance, as for spheres
    uint32_t ntris, ninst;
    // optional hierarchy (SPT_ACCEL_BVH, spt_bvh.h); null = the exhaustive loop
    const float4* bvh_nodes;       // 4 per node: both children's boxes + their references
    const float4* bvh_tris;        // the records of `tris` in leaf order
    const uint32_t* bvh_index;     // global triangle index of every leaf-order triangle
    const float4* bvh_cones;       // 3 per node of the hierarchy: the children's normal cones (spt_tribvh.h (1)); null = SPT_ACCEL_BVH_FAST
    const float4* plane_nodes;     // cone tree over the regular triangles' planes (spt_tribvh.h (2)); null = none
    const float4* line_nodes;      // cone tree over the thin triangles' long edges (spt_tribvh.h (3)); null = none or the table below
    const float4* flat_lines;      // table form of the thin triangles: groups {p, count} {eh, tol} ... (spt_tribvh.h); null = none or the tree above
    const uint32_t* flat_line_index;
    uint32_t nline_slots;
    // rays of depth 0 of a render launch (their lines all pass through the camera's origin) skip the plane tree and test this list instead:
    // the regular triangles in whose plane that origin lies (spt_bvh.h camera_planes; empty, as a rule).  cam_cull = 0: spt_trace_rays
    const uint32_t* cam_planes;
    uint32_t ncam, cam_cull;
    // hierarchy kernel: 1 = the lanes of a wave are dealt an 8 x 8 tile of pixels (spt_deal.h deal_task_tiles: coherent walks, 2.5 x on diffuse
    // scenes); 0 = a pixel's tasks and its neighbours' go to different waves (deal_task), chosen when a material is SPEC or REFR: a tile of
    // pixels looking into a mirror or glass ball would put 64 chains of thousands of bounces into one wave
    uint32_t strips;
    // sphere tables through the same kernel (spt_set_sphere_accel): bvh_tris holds one {centre, r*r} per sphere in leaf order,
    // `always` the spheres kept out of the tree; hits take centre / material from KParams::geom / mat
    const uint32_t* always;
    uint32_t nalways, sphere_mode;
};

// Grid-pool kernel (spt_gpool.hip): the launch parameters besides KParams / GridParams
struct QParams {
    float4* slots;                 // waves x S x 6 float4: the path state of every slot (global memory, cache resident)
    uint32_t S;                    // path slots per wave (multiple of 16, <= 256)
    uint32_t R;                    // begun walks a wave can hold in LDS (56 bytes each; multiple of 4)
    uint32_t drain;                // walker lanes that must be finished / empty before the wave stops walking to exchange them
    uint32_t min_batch;            // smallest shading / generation batch worth running while the walkers starve
    uint32_t walk_iters;           // walk iterations between the issue of a batch's loads and the batch's code
};

}  // namespace spt

namespace spt { struct GridParams; }
extern "C" size_t spt_gpool_lds_bytes(const spt::GridParams* G, uint32_t waves, uint32_t S, uint32_t R);
extern "C" size_t spt_gpool_slot_floats(uint32_t blocks, uint32_t waves, uint32_t S);
extern "C" size_t spt_gpool_stack_floats(uint32_t blocks, uint32_t waves, uint32_t S);
extern "C" hipError_t spt_gpool_launch(const spt::KParams* K, const spt::GridParams* G, const uint32_t* d_cells, const uint16_t* d_refs,
                                       const uint32_t* d_always, const spt::QParams* Q, uint32_t blocks, uint32_t threads, int stats, hipStream_t stream);
extern "C" size_t spt_grid_lds_bytes(const spt::GridParams* G);
extern "C" int spt_grid_block_threads(void);
extern "C" size_t spt_grid_stack_floats(uint32_t blocks, uint32_t threads);
extern "C" hipError_t spt_grid_launch(const spt::KParams* K, const spt::GridParams* G, const uint32_t* d_cells, const uint16_t* d_refs,
                                      const uint32_t* d_always, uint32_t blocks, uint32_t threads, uint32_t leave_q, int stats, int where, hipStream_t stream);
extern "C" size_t spt_grid_lds_bytes_tables(const spt::GridParams* G);
extern "C" size_t spt_mesh_lds_bytes(int bvh);
extern "C" size_t spt_mesh_stack_floats(uint32_t blocks);
extern "C" hipError_t spt_mesh_launch(const spt::KParams* K, const spt::MParams* M, uint32_t blocks, hipStream_t stream);
extern "C" hipError_t spt_mesh_trace_rays(const spt::MParams* M, const float* d_rays, uint64_t nrays, float* d_hits, hipStream_t stream);
extern "C" size_t spt_k_lds_bytes(uint32_t n_pad, int mat_lds, int big_block);
extern "C" size_t spt_k_stack_floats(uint32_t blocks, int block_threads);
extern "C" hipError_t spt_k_launch(const spt::KParams* P, uint32_t blocks, int mat_lds, int guard, int diag, int bign, int big_block, hipStream_t stream);
extern "C" hipError_t spt_pool_chunk_order(const uint32_t* chunk_clock, uint32_t nchunks, uint32_t ntasks, uint32_t* chunk_order, uint32_t* work512, hipStream_t stream);
extern "C" hipError_t spt_k_finalize(const float4* cells, float* out, uint32_t npix, float scale, int normalise, uint32_t nb, hipStream_t stream);
extern "C" int spt_k_block_threads(void);
extern "C" int spt_k_block_threads_for(int mat_lds, int big_block);
extern "C" hipError_t spt_k_selftest(int op, const float* d_in, float* d_out, uint32_t n, uint32_t w, hipStream_t stream);
extern "C" size_t spt_pool_lds_bytes(uint32_t n, int pool);
extern "C" size_t spt_pool_stack_floats(uint32_t blocks, int pool);
extern "C" size_t spt_pool_state_bytes(uint32_t blocks, int pool);
extern "C" int spt_pool_max_spheres(void);
extern "C" int spt_pool_default_slots(void);
extern "C" int spt_pool_has_size(int pool);
extern "C" hipError_t spt_pool_launch(const spt::KParams* K, uint32_t blocks, int pool, hipStream_t stream);
extern "C" hipError_t spt_k_selftest_range(int op, uint32_t first, uint32_t count, unsigned long long* d_mismatches, uint32_t* d_first_bad, hipStream_t stream);
extern "C" hipError_t spt_k_accumulate(float* accum, const float* frame, size_t n, int clear, hipStream_t stream);
