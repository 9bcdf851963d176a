// spt_kernel.hip -- persistent-wavefront path-tracing megakernel for gfx950 (MI355X).
//
// Replaces the reference's per-bounce host loop { Intersector::traceRays -> shadePaths -> compact }
// (smallpt.cpp:349-356 / :779-807) by one launch in which every lane owns a path from camera ray to
// termination:
//   * work unit ("task") = one jitter cell of one pixel (pixel*4 + sy*2+sx, smallpt.cpp:299-309);
//     a lane runs the task's `samps` samples in order and writes ONE 16-byte cell sum.  Lanes pull
//     tasks from a global atomic queue (wave-aggregated fetch), so there is no per-tile tail.
//   * the recursive radiance() / the wavefront path buffers become an iterative loop with path
//     regeneration; the glass split (smallpt.cpp:248-254) uses a <=3-entry per-lane stack in LDS.
//   * the sphere table is staged in LDS once per workgroup ({center, r*r} 16 B per sphere) and read
//     with wave-uniform (broadcast) ds_read_b128 in the closest-hit loop (smallpt.cpp:54-70).
//   * RNG is counter-based (D7), so the image does not depend on grid size, scheduling or GPU count.
// A second tiny kernel folds the four cell sums of a pixel in fixed order and normalises (D9).
#include "spt_device.h"
#include "spt_kernel.h"

namespace spt {

constexpr int kBlock = 256;
constexpr float kInf = 1e20f;   // maths.h:16
constexpr float kEps = 1e-4f;   // scene.cpp:133

// Per-thread LDS scratch, laid out [slot][field][thread] so that every access is conflict-free:
//   stack: pending transmitted children of the glass split (smallpt.cpp:252), <= 3 per lane
//   ring : pre-generated camera rays of the lane's current task (path regeneration queue)
constexpr int kStackFields = 10;   // o.xyz d.xyz w.xyz (depth | branch << 16)
constexpr int kStackEntries = 3;
constexpr int kRingFields = 6;     // dd.xyz, 1/|dd|, k0, k1
constexpr int kRing = SPT_RING;    // ring slots per lane

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <bool MAT_LDS>
__global__ __launch_bounds__(kBlock) void megakernel(const KParams P)
{
    extern __shared__ float4 lds[];
    float4* s_geom = lds;                                  // n entries {c.xyz, r*r}
    float4* s_mat = lds + P.n_pad;                         // 3*n entries when MAT_LDS
    float* s_stack = reinterpret_cast<float*>(lds + P.n_pad + (MAT_LDS ? 3 * P.n_pad : 0));
    float* s_ring = s_stack + kStackEntries * kStackFields * kBlock;

    const int tid = threadIdx.x;
    for (uint32_t i = tid; i < P.n; i += kBlock) {
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
    bool alive = false;          // a path is in flight
    bool task_valid = false, queue_empty = false;
    uint32_t task = 0, sp = 0;
    uint32_t s_gen = P.samps;    // next sample of the task to generate a camera ray for
    uint32_t rcount = 0, rhead = 0;   // camera-ray ring: entries ready, index of the oldest
    uint32_t px = 0, py = 0, cell = 0, p0 = 0, p1 = 0, k0 = 0, k1 = 0, rbase = 0;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 1), w = mk(0, 0, 0), acc = mk(0, 0, 0);
    uint32_t depth = 0, branch = 0;
    uint32_t nbounce = 0, nkill = 0;

    auto stack_at = [&](uint32_t e, int f) -> float& { return s_stack[(e * kStackFields + f) * kBlock + tid]; };
    auto ring_at = [&](uint32_t e, int f) -> float& { return s_ring[(e * kRingFields + f) * kBlock + tid]; };

    for (;;) {
        // ---- phase A: resume a pending transmitted child (smallpt.cpp:252) ----
        if (!alive && sp > 0) {
            --sp;
            o = mk(stack_at(sp, 0), stack_at(sp, 1), stack_at(sp, 2));
            d = mk(stack_at(sp, 3), stack_at(sp, 4), stack_at(sp, 5));
            w = mk(stack_at(sp, 6), stack_at(sp, 7), stack_at(sp, 8));
            const uint32_t db = __float_as_uint(stack_at(sp, 9));
            depth = db & 0xFFFFu; branch = db >> 16;
            rbase = rng_base(k0, branch, depth);
            alive = true;
        }
        // ---- phase B: task completion + wave-aggregated fetch from the global queue ----
        const bool idle = !alive && rcount == 0;                 // (sp == 0 here: phase A would have popped)
        const bool need_task = idle && s_gen == P.samps && !queue_empty;
        const unsigned long long need_mask = __ballot(need_task);
        if (need_mask != 0ull) {
            if (need_task && task_valid) P.cells[task] = make_float4(acc.x, acc.y, acc.z, 0.0f);
            const uint32_t cnt = (uint32_t)__popcll(need_mask);
            const int leader = __ffsll((long long)need_mask) - 1;
            uint32_t base = 0;
            if ((int)lane_id() == leader) base = atomicAdd(P.queue, cnt);
            base = __builtin_amdgcn_readfirstlane(__shfl(base, leader));
            const uint32_t rank = (uint32_t)__popcll(need_mask & ((1ull << lane_id()) - 1ull));
            if (need_task) {
                task = base + rank;
                task_valid = task < P.ntasks;
                if (task_valid) {
                    const uint32_t pix_local = task >> 2;
                    cell = task & 3u;
                    const uint32_t ry = pix_local / P.w;
                    px = pix_local - ry * P.w;
                    py = P.row_begin + ry;
                    const uint32_t pixel_idx = py * P.w + px;          // GLOBAL index (smallpt.cpp:298)
                    p0 = mix32(pixel_idx + P.s0);
                    p1 = mix32(pixel_idx ^ P.s1);
                    s_gen = 0;
                    acc = mk(0, 0, 0);
                } else {
                    queue_empty = true;
                }
            }
        }
        // ---- phase C1: batched path regeneration (smallpt.cpp:325-340).  Runs only when some lane is out
        // of camera rays; then EVERY lane with a free ring slot generates one, so the ~150-instruction
        // generator executes with most lanes active instead of once per terminated path. ----
        const bool starved = !alive && rcount == 0 && task_valid && s_gen < P.samps;
        if (__ballot(starved) != 0ull) {
            if (task_valid && s_gen < P.samps && rcount < (uint32_t)kRing) {
                const uint32_t index_in_pixel = cell * P.samps + s_gen;      // smallpt.cpp:306
                const uint32_t gk0 = mix32(p0 ^ (index_in_pixel * kGolden));
                const uint32_t gk1 = mix32(p1 + index_in_pixel * 0x85EBCA6Bu);
                const float u1 = rng_draw(gk0 + ((1u << 28) | 0u) * kGolden, gk1);
                const float u2 = rng_draw(gk0 + ((1u << 28) | 1u) * kGolden, gk1);
                // tent filter :327-330; r in {0} U [2^-23, 2): the un-guarded sqrt fix-up is exact here
                const float r1 = 2 * u1;
                const float a1 = r1 < 1 ? r1 : 2 - r1;
                const float q1 = sqrt_fix(a1);
                const float dx = r1 < 1 ? q1 - 1 : 1 - q1;
                const float r2 = 2 * u2;
                const float a2 = r2 < 1 ? r2 : 2 - r2;
                const float q2 = sqrt_fix(a2);
                const float dy = r2 < 1 ? q2 - 1 : 1 - q2;
                const uint32_t sx = cell & 1u, sy = cell >> 1;
                // :331-332 in double as in the reference.  a / w is evaluated as q0 = a*y, q = fma(fma(-q0,w,a), y, q0)
                // with y = RN(1/w): the correctly rounded quotient (Markstein; checked in tools/verify_exact_math.c).
                const double tx = ((double)sx + .5 + (double)dx) / 2.0 + (double)px;
                const double ty = ((double)sy + .5 + (double)dy) / 2.0 + (double)py;
                const double qx0 = tx * P.inv_w, qy0 = ty * P.inv_h;
                const double qx = __builtin_fma(__builtin_fma(-qx0, (double)P.w, tx), P.inv_w, qx0);
                const double qy = __builtin_fma(__builtin_fma(-qy0, (double)P.h, ty), P.inv_h, qy0);
                const float ax = (float)(qx - .5), ay = (float)(qy - .5);
                const f3 dd = cam_cx * ax + cam_cy * ay + cam_d;
                const float inv = 1.0f / sqrt_exact(dot(dd, dd));
                uint32_t slot = rhead + rcount;
                if (slot >= (uint32_t)kRing) slot -= (uint32_t)kRing;
                ring_at(slot, 0) = dd.x; ring_at(slot, 1) = dd.y; ring_at(slot, 2) = dd.z; ring_at(slot, 3) = inv;
                ring_at(slot, 4) = __uint_as_float(gk0); ring_at(slot, 5) = __uint_as_float(gk1);
                ++rcount;
                ++s_gen;
            }
        }
        // ---- phase C2: start the next camera path from the ring (cheap: 6 LDS reads + 9 VALU) ----
        if (!alive && rcount > 0) {
            const f3 dd = mk(ring_at(rhead, 0), ring_at(rhead, 1), ring_at(rhead, 2));
            const float inv = ring_at(rhead, 3);
            k0 = __float_as_uint(ring_at(rhead, 4));
            rbase = k0;                                          // k0 + ctr(branch 0, depth 0) * golden
            k1 = __float_as_uint(ring_at(rhead, 5));
            o = cam_o + dd * P.cam_push;                                                // :333
            d = dd * inv;                                                               // normalize(d)
            w = mk(1, 1, 1); depth = 0; branch = 0;                                      // :338-339
            ++rhead; if (rhead >= (uint32_t)kRing) rhead = 0;
            --rcount;
            alive = true;
        }
        if (__ballot(alive) == 0ull) break;   // no lane has a path, a stack entry, a camera ray, a sample or a task left

        // ---- phase D: one bounce = intersectGlobalSpheres + shadePaths body ----
        if (alive) {
            ++nbounce;
            // closest hit, smallpt.cpp:54-70 over scene.cpp:129-140 (D1, D16).  Branch-free per sphere:
            // det < 0 gives sqrt = NaN and every comparison below is false, exactly like the early return.
            float nearest = kInf;
            uint32_t inst = 0;
            float4 g = s_geom[0];
            for (uint32_t i = 0; i < P.n; ++i) {
                const float4 gn = s_geom[i + 1 < P.n ? i + 1 : i];   // prefetch next sphere (wave-uniform LDS broadcast)
                const f3 op = mk(g.x - o.x, g.y - o.y, g.z - o.z);                    // :132
                const float b = dot(op, d);                                            // :133
                const float det = b * b - dot(op, op) + g.w;                           // :133 (g.w = r*r)
                const float sd = sqrt_exact(det);                                      // :134
                const float t1 = b - sd, t2 = b + sd;                                  // :135
                const float t = t1 > kEps ? t1 : t2;
                if (t > kEps && t < nearest) { nearest = t; inst = i; }                // :135-136, smallpt.cpp:61
                g = gn;
            }
            if (nearest == kInf) {
                alive = false;                                                         // :168 miss
            } else {
                const float4 gh = s_geom[inst];
                const float4 me = mats[3 * inst + 0];         // emission.xyz, refl
                const float4 mc = mats[3 * inst + 1];         // color.xyz, pmax
                const f3 hx = o + d * nearest;                                         // scene.cpp:137
                const f3 n = normalize(mk(hx.x - gh.x, hx.y - gh.y, hx.z - gh.z));     // scene.cpp:124
                const f3 nl = dot(n, d) < 0 ? n : neg(n);                              // :174 (D2)
                f3 f = mk(mc.x, mc.y, mc.z);                                           // :175
                acc = acc + w * mk(me.x, me.y, me.z);                                  // :179 (D4)
                const int refl = __float_as_int(me.w);
                bool cont = true;
                if (depth > 5) {                                                       // :188 (D5)
                    if (rng_draw(rbase, k1) < mc.w) {
                        const float4 mf = mats[3 * inst + 2]; // color * (1/pmax)
                        f = mk(mf.x, mf.y, mf.z);                                      // :192
                    } else {
                        cont = false;                                                  // :196
                    }
                }
                if (cont) {
                    const f3 off = nl * 0.02f;                                         // :172 (D3)
                    f3 no = hx + off, nd, nf = f;
                    if (refl == 0) {                                                   // DIFF :208-215
                        const float u1 = rng_draw(rbase + kGolden, k1);
                        const float r2 = rng_draw(rbase + 2u * kGolden, k1);
                        const float r2s = sqrt_fix(r2);                               // r2 in {0} U [2^-24, 1)
                        float sn, cs;
                        sincos2pi(u1, sn, cs);                                          // D17
                        const f3 ww = nl;
                        // (double)fabs(w.x) > .1  <=>  fabsf(w.x) >= 0.1f  (0.1f is the least float above 0.1)
                        const f3 uu = normalize(cross(__builtin_fabsf(ww.x) >= 0.1f ? mk(0, 1, 0) : mk(1, 0, 0), ww));
                        const f3 vv = cross(ww, uu);
                        nd = normalize(uu * cs * r2s + vv * sn * r2s + ww * sqrt_fix(1 - r2)); // :212
                    } else {
                        const f3 rd = d - n * 2.0f * dot(n, d);                        // :218
                        nd = rd;
                        if (refl == 2) {                                               // REFR :225-263
                            const bool into = dot(n, nl) > 0;                          // :225
                            const float nnt = into ? 1.0f / 1.5f : 1.5f / 1.0f;        // :228
                            const float ddn = dot(d, nl);                              // :229
                            const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn);       // :230
                            if (!(cos2t < 0)) {                                        // else TIR :232-236
                                const f3 tdir = normalize(d * nnt - n * ((into ? 1.0f : -1.0f) * (ddn * nnt + sqrt_exact(cos2t)))); // :238
                                const float R0 = (0.5f * 0.5f) / (2.5f * 2.5f);        // :240-242
                                const float c = 1 - (into ? -ddn : dot(tdir, n));      // :243
                                const float c2 = c * c;                                // :244
                                const float Re = R0 + (1 - R0) * c2 * c2 * c;          // :245
                                const float Tr = 1 - Re;                               // :246
                                const f3 xin = hx - off;                               // D3
                                if (depth <= 2) {                                      // :248 split (D6)
                                    // transmitted child -> LDS stack; reflected child continues (:251-252)
                                    const f3 tw = w * (f * Tr);
                                    const bool keep = !(tw.x == 0.f && tw.y == 0.f && tw.z == 0.f);
                                    if (keep) {
                                        stack_at(sp, 0) = xin.x; stack_at(sp, 1) = xin.y; stack_at(sp, 2) = xin.z;
                                        stack_at(sp, 3) = tdir.x; stack_at(sp, 4) = tdir.y; stack_at(sp, 5) = tdir.z;
                                        stack_at(sp, 6) = tw.x; stack_at(sp, 7) = tw.y; stack_at(sp, 8) = tw.z;
                                        stack_at(sp, 9) = __uint_as_float((depth + 1u) | ((branch | (1u << depth)) << 16));
                                        ++sp;
                                    }
                                    nf = f * Re;
                                } else {
                                    const float Pr = 0.25f + 0.5f * Re;                // :256
                                    if (rng_draw(rbase + kGolden, k1) < Pr) {
                                        nf = f * Re * (1.0f / Pr);                     // :259
                                    } else {
                                        nf = f * Tr * (1.0f / (1.f - Pr));             // :263
                                        no = xin; nd = tdir;
                                    }
                                }
                            }
                        }
                    }
                    // extend(), smallpt.cpp:120-123, + D18 depth cap + zero-weight cut
                    w = w * nf;
                    o = no; d = nd;
                    ++depth;
                    rbase += 4u * kGolden;
                    if (depth >= SPT_K_MAX_DEPTH) { cont = false; ++nkill; }
                    else if (w.x == 0.f && w.y == 0.f && w.z == 0.f) cont = false;
                }
                alive = cont;
            }
        }
    }

    // stats: wave reduction then one atomic per wave
    unsigned long long nb = nbounce, nk = nkill;
    for (int off = 32; off > 0; off >>= 1) { nb += __shfl_down(nb, off); nk += __shfl_down(nk, off); }
    if (lane_id() == 0) {
        atomicAdd(&P.counters[0], nb);
        if (nk) atomicAdd(&P.counters[1], nk);
    }
}

// D9: pixel = ((c0 + c1) + c2) + c3, optional * (1/spp) (smallpt.cpp:358-361).  One lane per pixel,
// 64 B contiguous cell reads per lane, 12 B per pixel written; w*rows*12 bytes of HBM stores.
__global__ __launch_bounds__(kBlock) void finalize(const float4* __restrict__ cells, float* __restrict__ out,
                                                   uint32_t npix, float scale, int normalise)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= npix) return;
    const float4 c0 = cells[4 * p + 0], c1 = cells[4 * p + 1], c2 = cells[4 * p + 2], c3 = cells[4 * p + 3];
    float x = ((c0.x + c1.x) + c2.x) + c3.x;
    float y = ((c0.y + c1.y) + c2.y) + c3.y;
    float z = ((c0.z + c1.z) + c2.z) + c3.z;
    if (normalise) { x *= scale; y *= scale; z *= scale; }
    out[3 * p + 0] = x; out[3 * p + 1] = y; out[3 * p + 2] = z;
}

}  // namespace spt

// ---- launch wrappers used by spt_api.cpp ----
extern "C" size_t spt_k_lds_bytes(uint32_t n_pad, int mat_lds)
{
    return (size_t)n_pad * 16u * (mat_lds ? 4u : 1u) +
           (size_t)(spt::kStackEntries * spt::kStackFields + spt::kRing * spt::kRingFields) * spt::kBlock * 4u;
}

extern "C" hipError_t spt_k_launch(const spt::KParams* P, uint32_t blocks, int mat_lds, hipStream_t stream)
{
    const size_t lds = spt_k_lds_bytes(P->n_pad, mat_lds);
    if (mat_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spt::megakernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(spt::megakernel<true>, dim3(blocks), dim3(spt::kBlock), lds, stream, *P);
    } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spt::megakernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(spt::megakernel<false>, dim3(blocks), dim3(spt::kBlock), lds, stream, *P);
    }
    return hipGetLastError();
}

extern "C" hipError_t spt_k_finalize(const float4* cells, float* out, uint32_t npix, float scale, int normalise, hipStream_t stream)
{
    const uint32_t blocks = (npix + spt::kBlock - 1) / spt::kBlock;
    hipLaunchKernelGGL(spt::finalize, dim3(blocks), dim3(spt::kBlock), 0, stream, cells, out, npix, scale, normalise);
    return hipGetLastError();
}

extern "C" int spt_k_block_threads(void) { return spt::kBlock; }
