/*
 * spt_internal.h -- test / tuning hooks of libsmallpt_mi355x.so.  NOT part of the drop-in boundary
 * (include/smallpt_mi355x.h): nothing here replaces a reference interface.  Used by tests/, tools/ and
 * bench.py's A/B switches only; results never depend on any of these knobs.
 */
#ifndef SPT_INTERNAL_H
#define SPT_INTERNAL_H

#include "../../include/smallpt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knobs (0 = default).  blocks_per_cu caps the persistent grid; variant bits 0..7 = number of
 * waiting lanes that triggers a wave's glass-shading pass (0 = default 8), bit 8 = instrumented kernel
 * build (see spt_diag), bit 9 = 512-thread workgroups for tables above 256 spheres, bit 10 = force the megakernel where the pool kernel
 * would run (and the grid kernel), bits 12:11 = pool slots per wave (0: 160, 3: 128; 1: 96 and 2: 192 in -DSPT_POOL_SIZES builds), bits 23:16 = grid kernel:
 * 1 + q, a wave leaves its walk phase when 16 x walking lanes < q x waiting lanes (0 = default q = 16), bits 31:24 = grid cells per sphere
 * (read by spt_set_scene; 0 = default 4).  Results never depend on these. */
int  spt_set_tuning(spt_ctx* ctx, uint32_t blocks_per_cu, uint32_t variant);
/* Large sphere tables through the uniform grid: lane_owned = 1 keeps the kernel whose lanes own their path (spt_grid.hip) where the
 * default -- wave-private path pools with walker lanes, spt_gpool.hip -- would run; slots / ready / drain / min_batch / walk_iters set
 * the pool geometry (0 = default 192 slots per wave, up to 96 begun walks per wave in LDS, an exchange per 24 finished walker lanes,
 * batches of >= 32 while the walkers starve, 4 walk iterations behind a batch's loads).  Results never depend on these. */
int  spt_set_grid_pools(spt_ctx* ctx, int lane_owned, uint32_t slots, uint32_t ready, uint32_t drain, uint32_t min_batch, uint32_t walk_iters);
/* Pool kernel: bit 13 = hand the task chunks out in their static order (no cost-ordered dispatch, spt_kernel.h KParams::chunk_order; the
 * grid kernel reads bits 15:13 as its workgroup size). */
/* Pool kernel, cost-ordered dispatch: copies the chunk order that the last pool launch left for the next launch of the same view
 * (a permutation of 0 .. nchunks - 1, most expensive chunk first) to `order` (room for `cap` words).  *nchunks = 0 when that
 * launch recorded none (a few samples per cell, tuning bit 13, another kernel).  Synchronises with the device. */
int  spt_chunk_order_snapshot(spt_ctx* ctx, uint32_t* order, uint32_t cap, uint32_t* nchunks);
/* Diagnostics of the last launch when variant bit 8 selected the instrumented kernel build:
 * out24[0..7] = wave-time (shader clocks) per phase, [8] iterations, [9..14] lane/run counters (24 words are written). */
int  spt_diag(spt_ctx* ctx, unsigned long long* out24);

/* Pool kernel only: a wave gives up `seconds` after its start (0 = never; default).  A launch in which that happened
 * makes spt_sync fail instead of returning an incomplete image.  Tests set a few seconds so that a scheduling bug
 * cannot hang the GPU box. */
int  spt_set_watchdog(spt_ctx* ctx, double seconds);
/* Which kernel ran the last launch: 1 = material-sorted pool kernel (spt_pool.hip), 0 = megakernel (spt_kernel.hip),
 * 2 = mesh kernel (spt_mesh.hip, triangles, exhaustive loop; 6 = through the exact hierarchy, 7 = through the plain one), 3 = mesh kernel over a sphere hierarchy (SPT_ACCEL_BVH), 4 = grid kernel with lane-owned
 * paths (spt_grid.hip), 5 = grid kernel with wave-private path pools (spt_gpool.hip).
 * After a grid launch spt_diag returns out24[0..1] = cell steps / sphere tests of the walks, [2..3] = wave iterations of either kind,
 * [4] = rays that took the exhaustive loop, [5] = rounds, [7] = shaded hits.
 * After a pool launch spt_diag returns out24[0..2] = batches per class (GEN, DIFF, REFR), [3..5] = lanes per class. */
int  spt_last_kernel(spt_ctx* ctx);

/* Numerics self-test of the kernel's exact-math helpers (host arrays in/out, n elements):
 * op 0 sqrt_fix, 2 sqrt_exact, 3 rcp_exact, 10 sqrt_rsq, 4 (float)((double)x / w) by the FMA sequence,
 * 5/6 sin/cos(2*pi*x) (D17), 7 rng_draw keyed by bits(x), 8/9 sin/cos from the raw draw bits carried in x.
 * Used by tests/test_gpu_math.py. */
int  spt_selftest_math(spt_ctx* ctx, int op, const float* in, float* out, uint32_t n, uint32_t w);

/* Exhaustive device checks of the two helpers whose exactness rests on the hardware's v_rsq_f32 / v_rcp_f32 tables
 * (csrc/spt_device.h), over every binary32 bit pattern in [first, first + count): number of mismatches and the smallest
 * offending pattern (0xFFFFFFFF if none).
 *   op 0  sqrt_rsq (the kernels' square root) against the CPU-proven sqrt_fix
 *   op 1  its uncorrected first estimate against the same: must mismatch (proves the comparison can fail)
 *   op 2  rcp_exact<false> (the kernels' reciprocal) against the compiler's IEEE division 1.0f / x
 *   op 3  bare v_rcp_f32 against the same: negative control
 * op 10 of spt_selftest_math evaluates sqrt_rsq elementwise. */
int  spt_selftest_range(spt_ctx* ctx, int op, uint32_t first, uint32_t count, uint64_t* mismatches, uint32_t* first_bad);

/* Host-only self-test of the SPT_ACCEL_BVH builder (csrc/spt_bvh.cpp; no device call, runs without a GPU): builds the
 * structures of csrc/spt_tribvh.h over the meshes' triangles and checks that every REGULAR triangle sits in exactly one leaf of the
 * spatial hierarchy and of the plane tree, every THIN one in the line table / tree, that every ancestor's box, normal cone, sigma / tau /
 * te (planes) or lam (lines) covers it, and that no reference lies deeper than the 32-entry traversal stack allows.
 * out4 = {nodes, leaves, depth of the spatial hierarchy, regular triangles (thin ones -- the pole needles of makeSphereTriMesh -- and
 * triangles with an edge of length zero are the rest)};
 * returns 0 = valid, 2 = invalid (reason in `why`), 1 = builder error. */
int  spt_selftest_bvh(const spt_mesh* meshes, uint32_t nmesh, uint32_t* out4, char* why, uint32_t why_len);
/* The same for the sphere hierarchy of spt_set_sphere_accel; out4 = {nodes, leaves, depth, always-tested spheres}. */
int  spt_selftest_sphere_bvh(const spt_sphere* spheres, uint32_t n, uint32_t* out4, char* why, uint32_t why_len);
/* Host-only self-test of the SPT_ACCEL_GRID builder (csrc/spt_grid.cpp): builds the uniform grid over the table at
 * `cells_per_sphere` (0 = the default resolution) and checks that every sphere is listed in every cell its error-bound cube
 * meets, that references are ascending and in range and that the ray test admits every origin inside the box.
 * out8 = {dim x, dim y, dim z, references, always-tested spheres, table bytes, usable, most references in one cell}; 0 = valid, 2 = not usable / invalid, 1 = builder error. */
int  spt_selftest_sphere_grid(const spt_sphere* spheres, uint32_t n, uint32_t cells_per_sphere, uint32_t* out8, char* why, uint32_t why_len);

/* libsmallpt_mi355x_multi.so: kernel watchdog (spt_set_watchdog) of ONE rank's context, so that a test can make exactly one
 * device's render fail and check that spt_multi_render returns its error instead of hanging in the exchange. */
struct spt_multi;
int  spt_multi_set_rank_watchdog(struct spt_multi* m, uint32_t rank, double seconds);
/* ... and a failure INSIDE the exchange: `rank` fails its part of the next RCCL exchange after every rank's rows are complete.  The failing
 * rank aborts every communicator (ncclCommAbort) so that no peer stays blocked in a send / receive; spt_multi_render returns its error and
 * the next call builds new communicators. */
int  spt_multi_inject_exchange_failure(struct spt_multi* m, uint32_t rank);

#ifdef __cplusplus
}
#endif
#endif /* SPT_INTERNAL_H */
