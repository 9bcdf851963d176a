// spt_kernel.hip -- persistent-wavefront path-tracing megakernel for gfx950 (MI355X).
//
// Replaces the reference's per-bounce host loop { Intersector::traceRays -> shadePaths -> compact }
// (smallpt.cpp:349-356 / :779-807) by one launch in which every lane owns a path from camera ray to
// termination:
//   * work unit ("task") = one D9 block of consecutive samples of one jitter cell of one pixel
//     (((pixel*4 + sy*2+sx) << nb_log2) | block, smallpt.cpp:299-309); a lane runs the block's samples in order
//     and writes ONE 16-byte block sum.  Lanes take
//     tasks from their wave's private chunk of 64 task ids; a wave touches the global queue word once
//     per chunk, so there is neither a per-tile tail nor contention on the queue.
//   * the recursive radiance() / the wavefront path buffers become an iterative loop with path
//     regeneration (camera rays are generated in batches into a 2-entry register queue); the glass
//     split (smallpt.cpp:248-254) uses a <=3-entry per-lane stack in LDS.
//   * the sphere table is staged in LDS once per workgroup ({center, r*r} 16 B per sphere) and read
//     with wave-uniform (broadcast) ds_read_b128 in the closest-hit loop (smallpt.cpp:54-70).
//   * RNG is counter-based (D7), so the image does not depend on grid size, scheduling or GPU count.
//   * every float operation is one IEEE binary32 operation (-ffp-contract=off); the short sqrt /
//     reciprocal sequences of spt_device.h are proven correctly rounded (tools/verify_exact_math.c).
// A second small kernel folds the block sums of a pixel's four cells in fixed order, normalises and writes the
// packed float3 rows with coalesced 16-byte stores (D9).  This kernel serves tables of more than 24 spheres, scenes
// that need the range-guarded sqrt and colours outside [0,1]; everything else runs spt_pool.hip.
#include "spt_device.h"
#include "spt_kernel.h"

namespace spt {

constexpr int kBlock = 256;
constexpr float kInf = 1e20f;   // maths.h:16
constexpr uint32_t kEpsKeyBias = 0x38D1B717u + 1u;            // bits(1e-4f) + 1
constexpr uint32_t kInfKey = 0x60AD78ECu - kEpsKeyBias;       // key of 1e20f (maths.h:16)

// Per-thread stack of pending transmitted children of the glass split (smallpt.cpp:252), <= 3 per lane, in global
// memory: one 64-byte line per record {o, depth | branch << 16} {d, -} {w, -} {pad} (pushes and pops are rare; keeping
// the stack out of LDS lets a CU hold more workgroups, which is what the large-table scenes need).
constexpr int kStackEntries = 3;
constexpr int kChunk = 64;         // task ids fetched from the global queue per atomic

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// One pre-generated camera ray (path regeneration queue entry): un-normalised direction, 1/|dd|, RNG keys.
struct CamRay { f3 dd; float inv; uint32_t k0, k1; };

// Per-lane path state.
struct Path {
    f3 o, d, w;
    uint32_t depth, branch, rbase;   // rbase = k0 + ctr(branch, depth, 0) * golden (D7)
};

// shadePaths body up to the material switch (smallpt.cpp:170-198): hit point, normal, emission, Russian
// roulette.  Returns false if the path dies in the roulette.
template <bool GUARD>
__device__ __forceinline__ bool shade_common(const Path& p, float t, const float4 gh, const float4 me, const float4 mc,
                                             const float4* mats, uint32_t inst, uint32_t k1, f3& acc,
                                             f3& hx, f3& n, f3& nl, f3& f)
{
    hx = p.o + p.d * t;                                                    // scene.cpp:137
    n = normalize<GUARD>(mk(hx.x - gh.x, hx.y - gh.y, hx.z - gh.z));       // scene.cpp:124
    nl = dot(n, p.d) < 0 ? n : neg(n);                                     // :174 (D2)
    f = mk(mc.x, mc.y, mc.z);                                              // :175
    acc = acc + p.w * mk(me.x, me.y, me.z);                                // :179 (D4)
    if (p.depth > 5) {                                                     // :188 (D5)
        if (rng_draw(p.rbase, k1) < mc.w) {
            const float4 mf = mats[3 * inst + 2];                          // color * (1/pmax)
            f = mk(mf.x, mf.y, mf.z);                                      // :192
        } else {
            return false;                                                  // :196
        }
    }
    return true;
}

// extend() smallpt.cpp:120-123 + D18 depth cap + zero-weight cut.  Returns whether the path continues.
__device__ __forceinline__ bool extend(Path& p, f3 no, f3 nd, f3 nf, uint32_t& nkill)
{
    p.w = p.w * nf;
    p.o = no; p.d = nd;
    ++p.depth;
    p.rbase += 4u * kGolden;
    if (p.depth >= SPT_K_MAX_DEPTH) { ++nkill; return false; }
    return !(p.w.x == 0.f && p.w.y == 0.f && p.w.z == 0.f);
}

// DIAG builds add s_memtime stamps around the phases and report per-phase wave-time sums (diagnostic only;
// selected by spt_set_tuning variant bit 8, never timed as the product kernel).
#define SPT_STAMP(slot)                                                        \
    if (DIAG) {                                                                \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();         \
        tsum[slot] += now__ - tlast;                                           \
        tlast = now__;                                                         \
    }

template <bool MAT_LDS, bool GUARD, bool DIAG, bool BIGN, int BLOCK>
__global__ __launch_bounds__(BLOCK) void megakernel(const KParams P)
{
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long iters = 0, lanes_d1 = 0, lanes_d2 = 0, lanes_d3 = 0, runs_d3 = 0, runs_c1 = 0, lanes_c1 = 0;
    extern __shared__ float4 lds[];
    float4* s_geom = lds;                                  // n entries {c.xyz, r*r}
    float4* s_mat = lds + P.n_pad;                         // 3*n entries when MAT_LDS

    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < P.n; i += BLOCK) {
        s_geom[i] = P.geom[i];
        if (MAT_LDS) {
            s_mat[3 * i + 0] = P.mat[3 * i + 0];
            s_mat[3 * i + 1] = P.mat[3 * i + 1];
            s_mat[3 * i + 2] = P.mat[3 * i + 2];
        }
    }
    __syncthreads();
    const float4* mats = MAT_LDS ? s_mat : P.mat;

    const f3 cam_o = mk(P.cam_o[0], P.cam_o[1], P.cam_o[2]);
    const f3 cam_d = mk(P.cam_d[0], P.cam_d[1], P.cam_d[2]);
    const f3 cam_cx = mk(P.cam_cx[0], P.cam_cx[1], P.cam_cx[2]);
    const f3 cam_cy = mk(P.cam_cy[0], P.cam_cy[1], P.cam_cy[2]);

    // per-lane persistent state
    bool alive = false;          // a path is in flight (possibly parked)
    bool parked = false;         // hit a REFR sphere; waits for the wave's next glass-shading pass
    bool task_valid = false, queue_empty = false;
    uint32_t task = 0, sp = 0;
    uint32_t s_gen = 0, s_end = 0;   // next sample of the task to generate a camera ray for / end of the task's sample block (D9)
    uint32_t rcount = 0;         // camera rays ready in the two-entry register queue (ra = oldest)
    CamRay ra{mk(0, 0, 0), 0.f, 0u, 0u}, rb{mk(0, 0, 0), 0.f, 0u, 0u};
    uint32_t px = 0, py = 0, cell = 0, p0 = 0, p1 = 0, k0 = 0, k1 = 0;
    Path p{mk(0, 0, 0), mk(0, 0, 1), mk(0, 0, 0), 0u, 0u, 0u};
    f3 acc = mk(0, 0, 0);
    float hit_t = 0.f;           // closest hit of the current bounce
    uint32_t hit_inst = 0;
    f3 pk_hx = mk(0, 0, 0), pk_n = mk(0, 0, 0), pk_nl = mk(0, 0, 0), pk_f = mk(0, 0, 0);   // parked glass hit
    uint32_t nbounce = 0, nkill = 0;
    uint32_t chunk_next = 0, chunk_end = 0;      // wave-uniform: this wave's private range of task ids

    float4* const gstack = reinterpret_cast<float4*>(P.stack) + ((size_t)blockIdx.x * BLOCK + tid) * (kStackEntries * 4);
    auto stack_rec = [&](uint32_t e) -> float4* { return gstack + e * 4u; };

    for (;;) {
        SPT_STAMP(7)
        // ---- phase A: resume a pending transmitted child (smallpt.cpp:252) ----
        if (!alive && sp > 0) {
            --sp;
            const float4* rec = stack_rec(sp);
            const float4 s0 = rec[0], s1 = rec[1], s2 = rec[2];
            p.o = mk(s0.x, s0.y, s0.z); p.d = mk(s1.x, s1.y, s1.z); p.w = mk(s2.x, s2.y, s2.z);
            const uint32_t db = __float_as_uint(s0.w);
            p.depth = db & 0xFFFFu; p.branch = db >> 16;
            p.rbase = rng_base(k0, p.branch, p.depth);
            alive = true;
        }
        SPT_STAMP(0)
        // ---- phase B: task completion + wave-aggregated fetch from the global queue ----
        const bool need_task = !alive && rcount == 0 && s_gen == s_end && !queue_empty;   // sp == 0 here
        const unsigned long long need_mask = __ballot(need_task);
        if (need_mask != 0ull) {
            if (need_task && task_valid) P.cells[task] = make_float4(acc.x, acc.y, acc.z, 0.0f);
            // Tasks are handed out from a wave-private chunk of kChunk consecutive task ids; only the chunk
            // refill touches the global queue word (one device-scope atomic per kChunk tasks per wave: a
            // per-task atomic saturates the single queue word at ~90 dequeues/us, MI355X_MICROARCH.md "dequeue").
            const uint32_t cnt = (uint32_t)__popcll(need_mask);
            const uint32_t rank = (uint32_t)__popcll(need_mask & ((1ull << lane_id()) - 1ull));
            const uint32_t avail = chunk_end - chunk_next;            // wave-uniform
            uint32_t base_old = chunk_next, base_new = 0;
            if (cnt > avail) {
                const int leader = __ffsll((long long)need_mask) - 1;
                uint32_t nb = 0;
                if ((int)lane_id() == leader) nb = atomicAdd(P.queue, (uint32_t)kChunk);
                base_new = __builtin_amdgcn_readfirstlane(__shfl(nb, leader));
                chunk_next = base_new + (cnt - avail);
                chunk_end = base_new + (uint32_t)kChunk;
            } else {
                chunk_next += cnt;
            }
            const uint32_t base = rank < avail ? base_old : base_new - avail;
            if (need_task) {
                task = deal_task(base + rank, P.ntasks);             // (spt_device.h: a pixel's blocks go to different waves)
                task_valid = task < P.ntasks;
                if (task_valid) {
                    // task = ((pixel * 4 + cell) << nb_log2) | block: one block of a jitter cell's samples (D9)
                    const uint32_t cellid = task >> P.nb_log2;
                    const uint32_t blk = task & ((1u << P.nb_log2) - 1u);
                    const uint32_t pix_local = cellid >> 2;
                    cell = cellid & 3u;
                    const uint32_t ry = pix_local / P.w;
                    px = pix_local - ry * P.w;
                    py = P.row_begin + (ry >> P.rb_log2) * P.rb_stride + (ry & P.rb_mask);      // contiguous band or interleaved row blocks
                    const uint32_t pixel_idx = py * P.w + px;          // GLOBAL index (smallpt.cpp:298)
                    p0 = mix32(pixel_idx + P.s0);
                    p1 = mix32(pixel_idx ^ P.s1);
                    s_gen = blk * P.sb;
                    s_end = s_gen + P.sb < P.samps ? s_gen + P.sb : P.samps;
                    acc = mk(0, 0, 0);
                } else {
                    queue_empty = true;
                }
            }
        }
        SPT_STAMP(1)
        // ---- phase C1: batched path regeneration (smallpt.cpp:325-340).  Runs only when some lane is out
        // of camera rays; then EVERY lane with a free queue slot generates one, so the ~130-instruction
        // generator executes with most lanes active instead of once per terminated path. ----
        const bool starved = !alive && rcount == 0 && task_valid && s_gen < s_end;
        if (__ballot(starved) != 0ull) {
            if (DIAG) { ++runs_c1; lanes_c1 += __popcll(__ballot(task_valid && s_gen < s_end && rcount < 2u)); }
            if (task_valid && s_gen < s_end && rcount < 2u) {
                if (nbounce > 0x40000000u) {           // keep the 32-bit per-lane counter from wrapping at extreme spp
                    atomicAdd(&P.counters[0], (unsigned long long)nbounce);
                    nbounce = 0;
                }
                const uint32_t index_in_pixel = cell * P.samps + s_gen;      // smallpt.cpp:306
                CamRay e;
                e.k0 = mix32(p0 ^ (index_in_pixel * kGolden));
                e.k1 = mix32(p1 + index_in_pixel * 0x85EBCA6Bu);
                const float u1 = rng_draw(e.k0 + ((1u << 28) | 0u) * kGolden, e.k1);
                const float u2 = rng_draw(e.k0 + ((1u << 28) | 1u) * kGolden, e.k1);
                const uint32_t sx = cell & 1u, sy = cell >> 1;
                float ax, ay;
                if (P.sampler == 0u) {
                    // tent filter :327-330; r in {0} U [2^-23, 2): the un-guarded sqrt_rsq is exact here
                    const float r1 = 2 * u1;
                    const float q1 = sqrt_rsq(r1 < 1 ? r1 : 2 - r1);
                    const float dx = r1 < 1 ? q1 - 1 : 1 - q1;
                    const float r2 = 2 * u2;
                    const float q2 = sqrt_rsq(r2 < 1 ? r2 : 2 - r2);
                    const float dy = r2 < 1 ? q2 - 1 : 1 - q2;
                    // :331-332 in double as in the reference.  a / w is evaluated as q0 = a*y, q = fma(fma(-q0,w,a), y, q0)
                    // with y = RN(1/w): the correctly rounded quotient (Markstein; checked in tools/verify_exact_math.c).
                    const double tx = ((double)sx + .5 + (double)dx) / 2.0 + (double)px;
                    const double ty = ((double)sy + .5 + (double)dy) / 2.0 + (double)py;
                    const double qx0 = tx * P.inv_w, qy0 = ty * P.inv_h;
                    const double qx = __builtin_fma(__builtin_fma(-qx0, (double)P.w, tx), P.inv_w, qx0);
                    const double qy = __builtin_fma(__builtin_fma(-qy0, (double)P.h, ty), P.inv_h, qy0);
                    ax = (float)(qx - .5); ay = (float)(qy - .5);
                } else {
                    // Renderer::render sampling, smallpt.cpp:745-760 + sampleRay :626-633 (all binary32):
                    // jittered = ((groupColumn, groupRow) + rand) * 0.5; box filter 0.5 * (2 r - 1); raster position
                    // (col + 0.5f) + that; * pixelSize; clip = 2 n - 1.
                    const float jx = ((float)sx + u1) * 0.5f, jy = ((float)sy + u2) * 0.5f;      // :750
                    const float fx = 0.5f * (2 * jx - 1), fy = 0.5f * (2 * jy - 1);              // :753-758
                    const float nx = (((float)px + 0.5f) + fx) * P.inv_wf;                       // :628-631
                    const float ny = (((float)py + 0.5f) + fy) * P.inv_hf;
                    ax = 2.f * nx - 1.f; ay = 2.f * ny - 1.f;                                    // :633
                }
                e.dd = cam_cx * ax + cam_cy * ay + cam_d;
                e.inv = rcp_exact(sqrt_exact(dot(e.dd, e.dd)));
                if (rcount == 0) ra = e; else rb = e;
                ++rcount;
                ++s_gen;
            }
        }
        SPT_STAMP(2)
        // ---- phase C2: start the next camera path from the queue (registers only) ----
        if (!alive && rcount > 0) {
            k0 = ra.k0; k1 = ra.k1;
            p.o = cam_o + ra.dd * P.cam_push;                                           // :333
            p.d = ra.dd * ra.inv;                                                       // normalize(d)
            p.w = mk(1, 1, 1); p.depth = 0; p.branch = 0; p.rbase = k0;                  // :338-339
            ra = rb;
            --rcount;
            alive = true;
        }
        SPT_STAMP(3)
        if (__ballot(alive) == 0ull) break;   // no lane has a path, a stack entry, a camera ray, a sample or a task left

        // ---- phase D1: closest hit, smallpt.cpp:54-70 over scene.cpp:129-140 (D1, D16).  Branch-free per
        // sphere: det < 0 gives sqrt = NaN and every comparison below is false, like the early return. ----
        bool shade = false;
        // A wave that has only a few rays left to answer (the end of a launch; one roulette-immune path -- colour (1,1,1) mirror or glass --
        // bouncing in a closed ball up to the depth cap) answers them ONE BY ONE WITH ALL ITS LANES: lane l tests spheres l, l + 64, ...
        // in ascending order with strict '<', then the wave takes the lexicographic minimum of (key, index): the key the sequential loop
        // ends with and the lowest index among the spheres that give it (:61).  n / 64 tests per lane and ray instead of n in one lane.
        uint32_t coop_key = kInfKey, coop_inst = 0;
        bool coop = false;
        if (BIGN) {
            unsigned long long todo = __ballot(alive && !parked);
            coop = todo != 0ull && (uint32_t)__popcll(todo) <= 4u && P.n >= 64u;
            if (coop) {
                while (todo != 0ull) {
                    const int rl = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const f3 ro = mk(__shfl(p.o.x, rl), __shfl(p.o.y, rl), __shfl(p.o.z, rl)), rd = mk(__shfl(p.d.x, rl), __shfl(p.d.y, rl), __shfl(p.d.z, rl));
                    uint32_t wk = kInfKey, wi = 0;
                    for (uint32_t i = threadIdx.x & 63u; i < P.n; i += 64u) {
                        const float4 g = s_geom[i];
                        const f3 op = mk(g.x - ro.x, g.y - ro.y, g.z - ro.z);               // :132
                        const float b = dot(op, rd);                                         // :133
                        const float det = b * b - dot(op, op) + g.w;                         // :133
                        const float sd = GUARD ? sqrt_exact(det) : sqrt_rsq(det);            // :134
                        const uint32_t key1 = __float_as_uint(b - sd) - kEpsKeyBias;         // :135
                        const uint32_t key2 = __float_as_uint(b + sd) - kEpsKeyBias;
                        const uint32_t key = key1 < key2 ? key1 : key2;
                        if (key < wk) { wk = key; wi = i; }
                    }
#pragma unroll 1
                    for (int off = 32; off > 0; off >>= 1) {
                        const uint32_t k2 = (uint32_t)__shfl_xor((int)wk, off), i2 = (uint32_t)__shfl_xor((int)wi, off);
                        const bool better = (k2 < wk) | ((k2 == wk) & (i2 < wi));
                        wk = better ? k2 : wk; wi = better ? i2 : wi;
                    }
                    if ((int)(threadIdx.x & 63u) == rl) { coop_key = wk; coop_inst = wk == kInfKey ? 0u : wi; }
                }
            }
        }
        if (alive && !parked) {
            ++nbounce;
            // Selection on integer keys: for positive floats the bit pattern orders like the value, so with
            // key(t) = bits(t) - (bits(eps) + 1) (mod 2^32) the test "t > eps && t < nearest" (scene.cpp:135-136,
            // smallpt.cpp:61) is ONE unsigned compare key < nearest_key: t <= eps, negative t and NaN (det < 0) all
            // wrap to keys above bits(1e20).  "t1 > eps ? t1 : t2" is min(key1, key2) because t2 >= t1.
            uint32_t near_key = kInfKey;
            uint32_t inst = 0;
            // one sphere test; branch-free (det < 0 -> NaN keys that never win)
            auto test_sphere = [&](const float4 g, uint32_t i, float b, float det) {
                const float sd = GUARD ? sqrt_exact(det) : sqrt_rsq(det);          // :134
                const uint32_t key1 = __float_as_uint(b - sd) - kEpsKeyBias;           // :135
                const uint32_t key2 = __float_as_uint(b + sd) - kEpsKeyBias;
                const uint32_t key = key1 < key2 ? key1 : key2;
                if (key < near_key) { near_key = key; inst = i; }                      // :135-136, smallpt.cpp:61
            };
            auto b_det = [&](const float4 g, float& b, float& det) {
                const f3 op = mk(g.x - p.o.x, g.y - p.o.y, g.z - p.o.z);               // :132
                b = dot(op, p.d);                                                      // :133
                det = b * b - dot(op, op) + g.w;                                       // :133 (g.w = r*r)
            };
            if (coop) {
                near_key = coop_key; inst = coop_inst;
            } else if (BIGN) {
                // Large tables: most spheres are missed by every ray of the wave (det < 0 in all lanes,
                // scene.cpp:134).  Groups of four spheres share one LDS wait and one wave-uniform test; the
                // sqrt/selection part runs only for groups in which some lane has det >= 0.  Exact: a skipped
                // sphere would have produced NaN keys in every lane.
                // Two register sets of four spheres are processed alternately; each is refilled (LDS, wave-uniform
                // broadcast reads) right after use, so its latency hides behind the other set's arithmetic.
                auto group = [&](uint32_t i, const float4 g0, const float4 g1, const float4 g2, const float4 g3) {
                    float b0, b1, b2, b3, d0, d1, d2, d3;
                    b_det(g0, b0, d0); b_det(g1, b1, d1); b_det(g2, b2, d2); b_det(g3, b3, d3);
                    if (__ballot(d0 >= 0.0f || d1 >= 0.0f || d2 >= 0.0f || d3 >= 0.0f) != 0ull) {
                        if (__ballot(d0 >= 0.0f) != 0ull) test_sphere(g0, i, b0, d0);
                        if (__ballot(d1 >= 0.0f) != 0ull) test_sphere(g1, i + 1, b1, d1);
                        if (__ballot(d2 >= 0.0f) != 0ull) test_sphere(g2, i + 2, b2, d2);
                        if (__ballot(d3 >= 0.0f) != 0ull) test_sphere(g3, i + 3, b3, d3);
                    }
                };
                uint32_t i = 0;
                const uint32_t n8 = P.n & ~7u;
                if (n8) {
                    // Hand-issued LDS reads (hipcc splits prefetched float4 records into 32 ds_read_b32 with an
                    // address move each).  LDS operations complete in order, so "lgkmcnt(4)" = everything except the
                    // four youngest reads has landed: the set being consumed is ready while the other set's refill
                    // is still in flight.  No other LDS/SMEM operation is issued inside this loop.
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f a0, a1, a2, a3, c0, c1, c2, c3;
                    const uint32_t geom_base = (uint32_t)(uintptr_t)s_geom;      // LDS byte address of the table
#define SPT_LDS4(r0, r1, r2, r3, byteaddr)                                                                         \
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\t"                           \
                                 "ds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48"                      \
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(byteaddr) : "memory")
#define SPT_WAIT4(r0, r1, r2, r3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory")
#define SPT_F4(v) make_float4((v).x, (v).y, (v).z, (v).w)
                    __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): nothing older in flight
                    SPT_LDS4(a0, a1, a2, a3, geom_base);
                    SPT_LDS4(c0, c1, c2, c3, geom_base + 64u);
                    for (; i < n8; i += 8) {
                        const uint32_t j = i + 8 < n8 ? i + 8 : i;      // next pair of groups (clamped: re-reads the last)
                        SPT_WAIT4(a0, a1, a2, a3);
                        __builtin_amdgcn_sched_barrier(0);
                        group(i, SPT_F4(a0), SPT_F4(a1), SPT_F4(a2), SPT_F4(a3));
                        SPT_LDS4(a0, a1, a2, a3, geom_base + j * 16u);
                        SPT_WAIT4(c0, c1, c2, c3);
                        __builtin_amdgcn_sched_barrier(0);
                        group(i + 4, SPT_F4(c0), SPT_F4(c1), SPT_F4(c2), SPT_F4(c3));
                        SPT_LDS4(c0, c1, c2, c3, geom_base + j * 16u + 64u);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : : "memory");
#undef SPT_LDS4
#undef SPT_WAIT4
#undef SPT_F4
                }
                for (; i < P.n; ++i) {
                    const float4 g = s_geom[i];
                    float b, det;
                    b_det(g, b, det);
                    test_sphere(g, i, b, det);
                }
            } else {
                // Small tables (n <= 24): fully unrolled with wave-uniform early exit, so that the LDS offsets and
                // the sphere indices are immediates (no per-sphere SGPR->VGPR moves, which issue at half rate).
#define SPT_SPH(i)                                                   \
                if (P.n <= (i)) goto spheres_done;                                  \
                {                                                                   \
                    const float4 g = s_geom[(i)];                                   \
                    float b, det;                                                   \
                    b_det(g, b, det);                                               \
                    test_sphere(g, (i), b, det);                                    \
                }
                SPT_SPH(0) SPT_SPH(1) SPT_SPH(2) SPT_SPH(3) SPT_SPH(4) SPT_SPH(5) SPT_SPH(6) SPT_SPH(7)
                SPT_SPH(8) SPT_SPH(9) SPT_SPH(10) SPT_SPH(11) SPT_SPH(12) SPT_SPH(13) SPT_SPH(14) SPT_SPH(15)
                SPT_SPH(16) SPT_SPH(17) SPT_SPH(18) SPT_SPH(19) SPT_SPH(20) SPT_SPH(21) SPT_SPH(22) SPT_SPH(23)
#undef SPT_SPH
            spheres_done:;
            }
            const float nearest = near_key == kInfKey ? kInf : __uint_as_float(near_key + kEpsKeyBias);
            hit_t = nearest; hit_inst = inst;
            if (nearest == kInf) alive = false;                                        // :168 miss (D13)
            else shade = true;
        }
        SPT_STAMP(4)
        if (DIAG) { ++iters; lanes_d1 += __popcll(__ballot(shade || (!alive && hit_t == kInf))); lanes_d2 += __popcll(__ballot(shade)); }
        // ---- phase D2: shadePaths (smallpt.cpp:170-223): common part for every hit, then DIFF / SPEC;
        // REFR hits are parked after the common part (hit point, normal, emission, roulette) ----
        if (shade) {
            const float4 me = mats[3 * hit_inst + 0];         // emission.xyz, refl
            const float4 gh = s_geom[hit_inst];
            const float4 mc = mats[3 * hit_inst + 1];         // color.xyz, pmax
            const int refl = __float_as_int(me.w) & 3;          // bit 2 = "emissive" flag for the pool kernel
            f3 hx, n, nl, f;
            bool cont = shade_common<GUARD>(p, hit_t, gh, me, mc, mats, hit_inst, k1, acc, hx, n, nl, f);
            if (cont) {
                if (refl == 2) {
                    parked = true;
                    pk_hx = hx; pk_n = n; pk_nl = nl; pk_f = f;
                } else {
                    const f3 no = hx + nl * 0.02f;                                     // :172 (D3)
                    f3 nd;
                    if (refl == 0) {                                                   // DIFF :208-215
                        const uint32_t u1bits = rng_draw_bits(p.rbase + kGolden, k1);
                        const float r2 = rng_draw(p.rbase + 2u * kGolden, k1);
                        const float r2s = sqrt_rsq(r2);                               // r2 in {0} U [2^-24, 1)
                        float sn, cs;
                        sincos2pi_bits(u1bits, sn, cs);                                 // D17, from the raw bits of u1
                        const f3 ww = nl;
                        // u = normalize(cross(|w.x| > .1 ? (0,1,0) : (1,0,0), w)), :211.  (double)fabs(w.x) > .1 <=>
                        // fabsf(w.x) >= 0.1f.  cross((0,1,0),w) = (w.z, 0, -w.x); cross((1,0,0),w) = (0, -w.z, w.y):
                        // the products with the axis' zeros only contribute signed zeros, which never reach a
                        // non-zero value or a comparison downstream.
                        const bool ay = __builtin_fabsf(ww.x) >= 0.1f;
                        const f3 ur = mk(ay ? ww.z : 0.f, ay ? 0.f : -ww.z, ay ? -ww.x : ww.y);
                        const float s2 = ay ? ww.x : ww.y;
                        const float qu = ww.z * ww.z + s2 * s2;       // dot(ur, ur) with the zero term dropped
                        const f3 uu = ur * rcp_exact<false>(sqrt_rsq<true, true>(qu));   // qu in [0.01, 1]
                        const f3 vv = cross(ww, uu);
                        nd = normalize<false>(uu * cs * r2s + vv * sn * r2s + ww * sqrt_rsq<true, true>(1 - r2)); // :212
                    } else {
                        nd = p.d - n * 2.0f * dot(n, p.d);                             // SPEC :218-223
                    }
                    cont = extend(p, no, nd, f, nkill);
                }
            }
            alive = cont;
        }
        SPT_STAMP(5)
        // ---- phase D3: glass (REFR, smallpt.cpp:225-263).  Executed for all parked lanes at once when enough
        // of them wait (or nothing else is left to do), so the block does not run for one or two lanes on
        // every iteration. ----
        {
            const unsigned long long pm = __ballot(parked);
            const unsigned long long others = __ballot(alive && !parked);
            if (pm != 0ull && ((uint32_t)__popcll(pm) >= P.park_threshold || others == 0ull)) {
                if (DIAG) { ++runs_d3; lanes_d3 += __popcll(pm); }
                if (parked) {
                    parked = false;
                    const f3 hx = pk_hx, n = pk_n, nl = pk_nl, f = pk_f;
                    const f3 off = nl * 0.02f;                                         // :172 (D3)
                    f3 no = hx + off, nf = f;
                    f3 nd = p.d - n * 2.0f * dot(n, p.d);                              // :218 reflRay
                    const bool into = dot(n, nl) > 0;                                  // :225
                    const float nnt = into ? 1.0f / 1.5f : 1.5f / 1.0f;                // :228
                    const float ddn = dot(p.d, nl);                                    // :229
                    const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn);               // :230
                    if (!(cos2t < 0)) {                                                // else TIR :232-236
                        const f3 tdir = normalize<true>(p.d * nnt - n * ((into ? 1.0f : -1.0f) * (ddn * nnt + sqrt_exact(cos2t)))); // :238
                        const float R0 = (0.5f * 0.5f) / (2.5f * 2.5f);                // :240-242
                        const float c = 1 - (into ? -ddn : dot(tdir, n));              // :243
                        const float c2 = c * c;                                        // :244
                        const float Re = R0 + (1 - R0) * c2 * c2 * c;                  // :245
                        const float Tr = 1 - Re;                                       // :246
                        const f3 xin = hx - off;                                       // D3
                        if (p.depth <= 2) {                                            // :248 split (D6)
                            // transmitted child -> LDS stack; reflected child continues (:251-252)
                            const f3 tw = p.w * (f * Tr);
                            if (!(tw.x == 0.f && tw.y == 0.f && tw.z == 0.f)) {
                                float4* rec = stack_rec(sp);
                                rec[0] = make_float4(xin.x, xin.y, xin.z, __uint_as_float((p.depth + 1u) | ((p.branch | (1u << p.depth)) << 16)));
                                rec[1] = make_float4(tdir.x, tdir.y, tdir.z, 0.f);
                                rec[2] = make_float4(tw.x, tw.y, tw.z, 0.f);
                                rec[3] = make_float4(0.f, 0.f, 0.f, 0.f);      // completes the 64-byte line
                                ++sp;
                            }
                            nf = f * Re;
                        } else {
                            const float Pr = 0.25f + 0.5f * Re;                        // :256
                            const bool pick_refl = rng_draw(p.rbase + kGolden, k1) < Pr;   // :257
                            // :259 f*Re/P  or  :263 f*Tr/(1-P): operator/(float3,float) multiplies by 1.0f/x
                            const float inv = rcp_exact(pick_refl ? Pr : 1.f - Pr);
                            nf = f * (pick_refl ? Re : Tr) * inv;
                            if (!pick_refl) { no = xin; nd = tdir; }
                        }
                    }
                    alive = extend(p, no, nd, nf, nkill);
                }
            }
        }
        SPT_STAMP(6)
    }

    if (DIAG && lane_id() == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&P.counters[2 + i], tsum[i]);
        atomicAdd(&P.counters[10], iters); atomicAdd(&P.counters[11], lanes_d1); atomicAdd(&P.counters[12], lanes_d2);
        atomicAdd(&P.counters[13], lanes_d3); atomicAdd(&P.counters[14], runs_d3); atomicAdd(&P.counters[15], runs_c1);
        atomicAdd(&P.counters[16], lanes_c1);
    }
    // stats: wave reduction then one atomic per wave
    unsigned long long nb = nbounce, nk = nkill;
    for (int off = 32; off > 0; off >>= 1) { nb += __shfl_down(nb, off); nk += __shfl_down(nk, off); }
    if (lane_id() == 0) {
        atomicAdd(&P.counters[0], nb);
        if (nk) atomicAdd(&P.counters[1], nk);
    }
}

// D9: cell = ((B0 + B1) + B2) + ... over its nb block sums, pixel = ((c0 + c1) + c2) + c3, optional * (1/spp)
// (smallpt.cpp:358-361).  A workgroup finishes 64 pixels = 256 jitter cells: their 256 * nb block sums (contiguous in
// HBM) are loaded with fully coalesced 16-byte loads into LDS, one padded row of nb + 1 float4 per cell (the odd row
// pitch makes the per-cell ds_read_b128 conflict-free); then one lane per cell adds its row in block order, the four
// cell sums of a pixel meet in the first lane of the quad, and the 64 packed float3 (768 B) are transposed through LDS
// and written as 48 coalesced 16-byte stores (ALIGNED16 build; the scalar build serves output pointers that are not
// 16-byte aligned).  HBM-bound: 64 * nb + 12 bytes per pixel.
template <bool ALIGNED16>
__global__ __launch_bounds__(kBlock) void finalize(const float4* __restrict__ cells, float* __restrict__ out,
                                                   uint32_t npix, float scale, int normalise, uint32_t nb, uint32_t nb_log2)
{
    constexpr uint32_t kPix = kBlock / 4;                       // pixels per workgroup
    extern __shared__ float4 s_rows[];                          // 256 x (nb + 1) float4, then the 768-byte output tile
    float4* const s_px = s_rows + kBlock * (nb + 1u);
    const uint32_t base = blockIdx.x * kPix;
    const size_t cell0 = (size_t)blockIdx.x * kBlock;           // first cell of the workgroup
    const size_t ncell = (size_t)npix * 4u;
    for (uint32_t j = 0; j < nb; ++j) {
        const uint32_t e = j * kBlock + threadIdx.x;            // element of the workgroup's contiguous 256 * nb float4
        const uint32_t c = e >> nb_log2, k = e & (nb - 1u);
        if (cell0 + c < ncell) s_rows[c * (nb + 1u) + k] = cells[cell0 * nb + e];
    }
    __syncthreads();
    const uint32_t p = base + (threadIdx.x >> 2);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < npix) {
        const float4* row = s_rows + threadIdx.x * (nb + 1u);
        a = row[0];
        for (uint32_t k = 1; k < nb; ++k) {
            const float4 v = row[k];
            a.x += v.x; a.y += v.y; a.z += v.z;
        }
    }
    // lanes 4q .. 4q+3 hold c0 .. c3 of one pixel: ((c0 + c1) + c2) + c3 in lane 4q
    const float x1 = __shfl_down(a.x, 1), y1 = __shfl_down(a.y, 1), z1 = __shfl_down(a.z, 1);
    const float x2 = __shfl_down(a.x, 2), y2 = __shfl_down(a.y, 2), z2 = __shfl_down(a.z, 2);
    const float x3 = __shfl_down(a.x, 3), y3 = __shfl_down(a.y, 3), z3 = __shfl_down(a.z, 3);
    float x = ((a.x + x1) + x2) + x3, y = ((a.y + y1) + y2) + y3, z = ((a.z + z1) + z2) + z3;
    if (normalise) { x *= scale; y *= scale; z *= scale; }
    const bool lead = (threadIdx.x & 3u) == 0u;
    const uint32_t q = threadIdx.x >> 2;                        // pixel within the workgroup
    const bool full = base + kPix <= npix;                      // workgroup-uniform
    if (ALIGNED16 && full) {
        float* s = reinterpret_cast<float*>(s_px);
        if (lead) { s[3 * q + 0] = x; s[3 * q + 1] = y; s[3 * q + 2] = z; }
        __syncthreads();
        if (threadIdx.x < (kPix * 3) / 4)
            reinterpret_cast<float4*>(out + (size_t)base * 3)[threadIdx.x] = s_px[threadIdx.x];
    } else if (lead && p < npix) {
        out[3 * (size_t)p + 0] = x; out[3 * (size_t)p + 1] = y; out[3 * (size_t)p + 2] = z;
    }
}

// Progressive accumulation of the viewer's render thread (smallpt.cpp:924-937): accum = clear ? frame : accum + frame,
// float4-vectorised grid-stride loop; n4 float4 elements plus a scalar tail.
__global__ __launch_bounds__(kBlock) void accumulate(float* __restrict__ accum, const float* __restrict__ frame, size_t n, int clear)
{
    const size_t n4 = n / 4;
    float4* a4 = reinterpret_cast<float4*>(accum);
    const float4* f4 = reinterpret_cast<const float4*>(frame);
    for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        const float4 f = f4[i];
        if (clear) { a4[i] = f; continue; }
        float4 a = a4[i];
        a.x += f.x; a.y += f.y; a.z += f.z; a.w += f.w;
        a4[i] = a;
    }
    for (size_t i = n4 * 4 + blockIdx.x * (size_t)kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        accum[i] = clear ? frame[i] : accum[i] + frame[i];
}

// Applies one of the exact-math device helpers elementwise (numerics self-test, tests/test_gpu_math.py).
__global__ void selftest_math(int op, const float* __restrict__ in, float* __restrict__ out, uint32_t n, double inv_w, uint32_t w)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    float y;
    switch (op) {
    case 0: y = sqrt_fix(x); break;
    case 2: y = sqrt_exact(x); break;
    case 3: y = rcp_exact(x); break;
    case 4: {   // double division a / w by the Markstein sequence, a = (double)x
        const double a = (double)x;
        const double q0 = a * inv_w;
        y = (float)__builtin_fma(__builtin_fma(-q0, (double)w, a), inv_w, q0);
        break;
    }
    case 5: { float sn, cs; sincos2pi(x, sn, cs); y = sn; break; }
    case 6: { float sn, cs; sincos2pi(x, sn, cs); y = cs; break; }
    case 10: y = sqrt_rsq(x); break;
    case 8: { float sn, cs; sincos2pi_bits(__float_as_uint(x), sn, cs); y = sn; break; }   // x carries the raw bits
    case 9: { float sn, cs; sincos2pi_bits(__float_as_uint(x), sn, cs); y = cs; break; }
    default: y = rng_draw(__float_as_uint(x), 0x9ABCDEF0u); break;
    }
    out[i] = y;
}

// Exhaustive checks of the helpers whose exactness rests on the hardware's v_rsq_f32 / v_rcp_f32 tables, over every binary32
// bit pattern in [first, first + count): counts the mismatches and keeps the smallest offending pattern.
// OP 0: sqrt_rsq against the CPU-proven sqrt_fix; 1: its uncorrected estimate (negative control);
// OP 2: rcp_exact<false> against the compiler's IEEE division; 3: bare v_rcp_f32 (negative control).
// Grid-stride; one launch covers 2^-96 .. FLT_MAX in a few milliseconds.
template <int OP>
__global__ __launch_bounds__(256) void selftest_range(uint32_t first, uint32_t count, unsigned long long* mismatches, uint32_t* first_bad)
{
    unsigned long long bad = 0;
    uint32_t worst = 0xFFFFFFFFu;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const uint32_t bits = first + (uint32_t)i;
        const float x = __uint_as_float(bits);
        float got, want;
        if (OP <= 1) { got = sqrt_rsq<OP == 0>(x); want = sqrt_fix(x); }
        else { got = OP == 2 ? rcp_exact<false>(x) : __builtin_amdgcn_rcpf(x); want = 1.0f / x; }   // the compiler's IEEE division
        if (__float_as_uint(got) != __float_as_uint(want)) { ++bad; worst = bits < worst ? bits : worst; }
    }
    if (bad) { atomicAdd(mismatches, bad); atomicMin(first_bad, worst); }
}

}  // namespace spt

extern "C" hipError_t spt_k_selftest_range(int op, uint32_t first, uint32_t count, unsigned long long* d_mismatches, uint32_t* d_first_bad, hipStream_t stream)
{
    switch (op) {
    case 0: hipLaunchKernelGGL(spt::selftest_range<0>, dim3(256 * 32), dim3(256), 0, stream, first, count, d_mismatches, d_first_bad); break;
    case 1: hipLaunchKernelGGL(spt::selftest_range<1>, dim3(256 * 32), dim3(256), 0, stream, first, count, d_mismatches, d_first_bad); break;
    case 2: hipLaunchKernelGGL(spt::selftest_range<2>, dim3(256 * 32), dim3(256), 0, stream, first, count, d_mismatches, d_first_bad); break;
    case 3: hipLaunchKernelGGL(spt::selftest_range<3>, dim3(256 * 32), dim3(256), 0, stream, first, count, d_mismatches, d_first_bad); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" hipError_t spt_k_accumulate(float* accum, const float* frame, size_t n, int clear, hipStream_t stream)
{
    size_t blocks = (n / 4 + spt::kBlock - 1) / spt::kBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(spt::accumulate, dim3((unsigned)blocks), dim3(spt::kBlock), 0, stream, accum, frame, n, clear);
    return hipGetLastError();
}

extern "C" hipError_t spt_k_selftest(int op, const float* d_in, float* d_out, uint32_t n, uint32_t w, hipStream_t stream)
{
    hipLaunchKernelGGL(spt::selftest_math, dim3((n + 255) / 256), dim3(256), 0, stream, op, d_in, d_out, n, 1.0 / (double)w, w);
    return hipGetLastError();
}

// ---- launch wrappers used by spt_api.cpp ----

template <bool M, bool G, bool D, bool B, int BLOCK>
static hipError_t launch_variant(const spt::KParams* P, uint32_t blocks, size_t lds, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spt::megakernel<M, G, D, B, BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((spt::megakernel<M, G, D, B, BLOCK>), dim3(blocks), dim3(BLOCK), lds, stream, *P);
    return hipGetLastError();
}

// Build variants: mat_lds (materials staged in LDS, n <= 256), guard (range-guarded sqrt in the hot loop, only
// for degenerate scenes), diag (instrumented), bign (grouped wave-uniform det < 0 skip, large tables).
// Large tables (materials in HBM) run `big_block` = 256 or 512 threads per workgroup (one LDS copy of the
// geometry per 4 or 8 waves); everything else runs 256.
extern "C" int spt_k_block_threads_for(int mat_lds, int big_block) { return mat_lds ? spt::kBlock : (big_block == 512 ? 512 : 256); }

extern "C" size_t spt_k_lds_bytes(uint32_t n_pad, int mat_lds, int big_block)
{
    const size_t block = (size_t)spt_k_block_threads_for(mat_lds, big_block);
    (void)block;
    return (size_t)n_pad * 16u * (mat_lds ? 4u : 1u);
}

extern "C" size_t spt_k_stack_floats(uint32_t blocks, int block_threads) { return (size_t)blocks * (size_t)block_threads * spt::kStackEntries * 16u; }

extern "C" hipError_t spt_k_launch(const spt::KParams* P, uint32_t blocks, int mat_lds, int guard, int diag, int bign, int big_block, hipStream_t stream)
{
    const size_t lds = spt_k_lds_bytes(P->n_pad, mat_lds, big_block);
    const bool b512 = !mat_lds && big_block == 512;
    constexpr int B0 = spt::kBlock;
    if (diag) {
        if (mat_lds) return launch_variant<true, false, true, false, B0>(P, blocks, lds, stream);
        return b512 ? launch_variant<false, false, true, true, 512>(P, blocks, lds, stream) : launch_variant<false, false, true, true, 256>(P, blocks, lds, stream);
    }
    (void)bign;   // product launches always take the grouped closest-hit loop: the unrolled small-table form (n <= 24) spilled 26 scalar
                  // registers and only served scenes the pool kernel refuses (range-guarded square root, colours outside [0,1])
    if (mat_lds) return guard ? launch_variant<true, true, false, true, B0>(P, blocks, lds, stream) : launch_variant<true, false, false, true, B0>(P, blocks, lds, stream);
    if (b512) return guard ? launch_variant<false, true, false, true, 512>(P, blocks, lds, stream) : launch_variant<false, false, false, true, 512>(P, blocks, lds, stream);
    return guard ? launch_variant<false, true, false, true, 256>(P, blocks, lds, stream) : launch_variant<false, false, false, true, 256>(P, blocks, lds, stream);
}

extern "C" hipError_t spt_k_finalize(const float4* cells, float* out, uint32_t npix, float scale, int normalise, uint32_t nb, hipStream_t stream)
{
    const uint32_t per_block = spt::kBlock / 4;                 // 64 pixels * 12 B = 768 B per workgroup keeps every block base 16-byte aligned
    const uint32_t blocks = (npix + per_block - 1) / per_block;
    uint32_t nb_log2 = 0;
    while ((1u << nb_log2) < nb) ++nb_log2;
    const size_t lds = ((size_t)spt::kBlock * (nb + 1u) + (per_block * 3) / 4) * sizeof(float4);
    if ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)
        hipLaunchKernelGGL(spt::finalize<true>, dim3(blocks), dim3(spt::kBlock), lds, stream, cells, out, npix, scale, normalise, nb, nb_log2);
    else
        hipLaunchKernelGGL(spt::finalize<false>, dim3(blocks), dim3(spt::kBlock), lds, stream, cells, out, npix, scale, normalise, nb, nb_log2);
    return hipGetLastError();
}

extern "C" int spt_k_block_threads(void) { return spt::kBlock; }
