// spt_pool.hip -- material-sorted persistent path-tracing kernel for gfx950 (MI355X), small sphere tables.
//
// Same algorithm and arithmetic as spt_kernel.hip (the per-bounce host loop { Intersector::traceRays ->
// shadePaths -> compact } of smallpt.cpp:349-356 / :779-807 collapsed into one launch), different scheduling:
//
//   * every WAVE owns a private pool of P path slots in LDS (P = 144 by default).  A slot is one task
//     (= one D9 block of consecutive samples of one jitter cell of one pixel, smallpt.cpp:299-309) with at most
//     one path in flight, so emission events of a block are accumulated in exactly the order of D9
//     (sample-ascending, DFS pre-order).
//   * slots wait in one of three wave-private LIFO lists by the NEXT thing their path needs:
//       GEN   start the next camera sample / pop a pending transmitted child / fetch a new task
//       DIFF  shade a DIFF or SPEC hit   (smallpt.cpp:208-223)
//       REFR  shade a glass hit          (smallpt.cpp:225-263)
//     Each iteration the wave pulls up to 64 slots of ONE class, runs that class's code with every lane
//     active, then -- in the same lanes -- the closest-hit query (smallpt.cpp:54-70) and the class-independent
//     part of shadePaths (emission, Russian roulette, weight update, :170-198), and pushes each slot onto
//     the list of its next class.  The megakernel of spt_kernel.hip runs DIFF shading at ~69 % and glass
//     shading at ~18 % lane utilisation because a lane owns its path; here a batch is ~96 % full
//     (three lists hold 144 entries; tools/pool_sim.py models the policy, the kernel counts its batches) at
//     the price of one LDS round trip of the 48-byte path state per bounce.  No cross-wave communication, no
//     barriers after scene staging.
//   * LDS per slot (70 bytes): path state between the phases {hit point, rbase} {direction, depth|branch|inst|refl|
//     stack count} {weight, k1} as three float4, the block sum of the task as three floats, {task id, next sample},
//     one byte in the GEN list and one in the array shared by the DIFF list (growing up) and the REFR list (growing
//     down).  144 slots per wave = 40 KB per 256-thread workgroup: four workgroups (16 waves) per CU.  The <= 3
//     pending transmitted children of the glass split (smallpt.cpp:252) are 48-byte records in global memory.
//   * A batch of <= 4 rays (the end of a launch, whose duration the last mirror <-> glass chains set) runs the closest hit
//     lane-parallel: lane 16 r + i tests sphere i for ray r, the row minimum comes from DPP rotations.
//   * RNG (D7), summation order (D9), sin/cos (D17), depth cap (D18), zero-weight cut (D19) and every
//     arithmetic expression are those of spt_kernel.hip / the oracle: results are bit-identical.
#include "spt_device.h"
#include "spt_kernel.h"

namespace spt {

constexpr uint32_t kEpsBias = 0x38D1B717u + 1u;                  // bits(1e-4f) + 1
constexpr uint32_t kInfKeyP = 0x60AD78ECu - kEpsBias;            // key of 1e20f
constexpr int kPoolBlock = 256;
constexpr uint32_t kNoTask = 0xFFFFFFFFu;
constexpr int kStackWords = 16;                                  // one pending child = one 64-byte line: {o, depth|branch<<16} {d, k0} {w, k1} {pad}
constexpr int kMaxUnroll = 24;                                   // spheres handled by the unrolled closest-hit code
#ifndef SPT_POOL_NARROW
#define SPT_POOL_NARROW 1                                        // lane-parallel closest hit for batches of <= 4 rays
#endif
#ifndef SPT_POOL_TASK_LDS
#define SPT_POOL_TASK_LDS 1                                      // {task id, next sample} of a slot in LDS (else in global memory)
#endif
constexpr bool kTaskLds = SPT_POOL_TASK_LDS != 0;
constexpr int kSlotBytes = kTaskLds ? 70 : 62;                   // LDS per pool slot (layout in poolkernel)

__device__ __forceinline__ uint32_t lane_id_p() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t rank_in(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t umin2(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// packed word of a slot: [11:0] depth, [14:12] branch bits (D7), [15] weight-may-be-non-finite flag,
// [27:16] sphere index, [29:28] Refl_t, [31:30] pending transmitted children of the slot's current sample
__device__ __forceinline__ uint32_t pack_path(uint32_t depth, uint32_t branchf, uint32_t inst, uint32_t refl, uint32_t sp)
{
    return depth | (branchf << 12) | (inst << 16) | (refl << 28) | (sp << 30);
}

enum { C_GEN = 0, C_DIFF = 1, C_REFR = 2 };

template <int P, int NG>
__global__ __launch_bounds__(kPoolBlock) void poolkernel(const KParams K)
{
    static_assert(P % 16 == 0 && P <= 256, "pool size");
    extern __shared__ float4 lds[];
    constexpr int kWaveF4 = (kSlotBytes * P) / 16;               // float4 per wave region
    static_assert((kSlotBytes * P) % 16 == 0, "wave region must be float4-aligned");
    const uint32_t lane = lane_id_p();
    const uint32_t wave = threadIdx.x >> 6;
    float4* const A0 = lds + wave * kWaveF4;                     // {hx.xyz, rbase}
    float4* const A1 = A0 + P;                                   // {d.xyz, packed}
    float4* const A2 = A1 + P;                                   // {w.xyz, k1}
    float* const ACX = reinterpret_cast<float*>(A2 + P);         // block sum of the slot's task (D9), SoA
    float* const ACY = ACX + P;
    float* const ACZ = ACY + P;
    uint2* const TS = reinterpret_cast<uint2*>(ACZ + P);         // {task id, next sample} (LDS build)
    uint8_t* const LG = reinterpret_cast<uint8_t*>(ACZ + P) + (kTaskLds ? 8 * P : 0);   // GEN list
    // LG[P .. 2P-1]: array shared by the DIFF list (from index 0 up) and the REFR list (from P-1 down)
    float4* const s_geom = lds + (kPoolBlock / 64) * kWaveF4;    // 3 NG x {c.xyz, r*r}
    float4* const s_mat = s_geom + 3 * NG;                       // 3 x (3 NG) material rows

    for (uint32_t i = threadIdx.x; i < 3u * NG; i += kPoolBlock) {
        const bool real = i < K.n;
        s_geom[i] = real ? K.geom[i] : make_float4(0.f, 0.f, 0.f, -__builtin_inff());   // padding: never hit
        s_mat[3 * i + 0] = real ? K.mat[3 * i + 0] : make_float4(0.f, 0.f, 0.f, 0.f);
        s_mat[3 * i + 1] = real ? K.mat[3 * i + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        s_mat[3 * i + 2] = real ? K.mat[3 * i + 2] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // every slot starts on the GEN list as a finished, task-less slot
    const uint32_t wave_gid = blockIdx.x * (kPoolBlock / 64) + wave;
    uint2* const gtask = kTaskLds ? TS : K.slot_state + (size_t)wave_gid * P;    // {task id, next sample} per slot
    for (uint32_t s = lane; s < (uint32_t)P; s += 64) {
        gtask[s] = make_uint2(kNoTask, 0u);
        A1[s].w = __uint_as_float(0u);                           // stack count 0
        LG[s] = (uint8_t)s;
    }
    __syncthreads();

    // pending transmitted children: [slot][entry] records of one 64-byte line each (48 bytes used, the line is written whole)
    float4* const gstack = reinterpret_cast<float4*>(K.stack) + (size_t)wave_gid * (3 * 4 * P);
    auto stack_rec = [&](uint32_t e, uint32_t slot) -> float4* { return gstack + (slot * 3u + e) * 4u; };

    const f3 cam_o = mk(K.cam_o[0], K.cam_o[1], K.cam_o[2]);
    const f3 cam_d = mk(K.cam_d[0], K.cam_d[1], K.cam_d[2]);
    const f3 cam_cx = mk(K.cam_cx[0], K.cam_cx[1], K.cam_cx[2]);
    const f3 cam_cy = mk(K.cam_cy[0], K.cam_cy[1], K.cam_cy[2]);

    // wave-uniform state
    uint32_t nG = (uint32_t)P, nD = 0u, nR = 0u;                 // list lengths
    uint32_t chunk_next = 0, chunk_end = 0;                      // this wave's private range of task ids
    bool queue_empty = false;
    unsigned long long nbounce = 0;                              // closest-hit queries of this wave
    uint32_t nkill = 0;                                          // per lane
    uint32_t nrec = 0;                                           // per lane: pending-child records written
    uint32_t itG = 0, itD = 0, itR = 0;                          // batches per class (utilisation report)
    uint32_t lnG = 0, lnD = 0, lnR = 0;                          // lanes per class (32 bits per wave: scalar registers are scarce in this loop)
    uint32_t itTail = 0, itFull = 0;                             // batches after the task queue ran dry / full batches
    uint32_t lnTail = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    uint32_t it_total = 0;
    bool timed_out = false;
    uint32_t dry_lo = 0, dry_hi = 0;                             // when this wave found the task queue empty: written once, read once,
                                                                 // so parked in vector registers rather than in scarce scalar ones
    // The tables of the cost-ordered dispatch are parked there too (5 registers of 30 spare ones).  In scalar registers they cost the
    // loop 20 spills; re-read from the kernel-argument segment where they are used, the compiler also re-reads other arguments
    // there, and a scalar load that may still be in flight when the closest-hit code is reached (they return out of order) turns
    // its counted waits on the sphere records -- lgkmcnt(5), (4), ... -- into one wait for all of them: 2.8 % of the launch.
    uint32_t ord_lo, ord_hi, clk_lo, clk_hi, nch_v;
    asm volatile("v_mov_b32 %0, %5\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %7\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %9"
                 : "=v"(ord_lo), "=v"(ord_hi), "=v"(clk_lo), "=v"(clk_hi), "=v"(nch_v)
                 : "s"((uint32_t)(uintptr_t)K.chunk_order), "s"((uint32_t)((uintptr_t)K.chunk_order >> 32)),
                   "s"((uint32_t)(uintptr_t)K.chunk_clock), "s"((uint32_t)((uintptr_t)K.chunk_clock >> 32)), "s"(K.nchunks));

#ifdef SPT_POOL_PHASES
    unsigned long long ph[5] = {0, 0, 0, 0, 0};                  // select+pop, class code, closest hit, post, push (lone-path latency study)
#define PH_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - ph_t; ph_t = t_; }
#else
#define PH_STAMP(i)
#endif
    for (;;) {
#ifdef SPT_POOL_PHASES
        unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif
        // ---- choose the class of this batch: a full batch of the rarest class first, else the longest list ----
        uint32_t c, b, lbase;
        if (nR >= 64u) c = C_REFR;
        else if (nG >= 64u) c = C_GEN;
        else if (nD >= 64u) c = C_DIFF;
        else {
            c = C_DIFF;
            uint32_t m = nD;
            if (nG > m) { c = C_GEN; m = nG; }
            if (nR > m) { c = C_REFR; m = nR; }
            if (m == 0u) break;                                   // every list empty: all tasks of this wave are done
        }
        if (c == C_GEN) { b = nG < 64u ? nG : 64u; nG -= b; lbase = nG; ++itG; lnG += b; }
        else if (c == C_DIFF) { b = nD < 64u ? nD : 64u; nD -= b; lbase = (uint32_t)P + nD; ++itD; lnD += b; }
        else { b = nR < 64u ? nR : 64u; lbase = 2u * (uint32_t)P - nR; nR -= b; ++itR; lnR += b; }   // REFR entries: LDR[P-nR .. P-1]
        if (queue_empty) { ++itTail; lnTail += b; }
        if (b == 64u) ++itFull;
        if ((++it_total & 255u) == 0u && K.watchdog_ticks != 0ull &&
            __builtin_amdgcn_s_memtime() - t_start > K.watchdog_ticks) { timed_out = true; break; }

        const bool valid = lane < b;
        const uint32_t slot = valid ? (uint32_t)LG[lbase + lane] : 0u;    // LDR follows LG: index P.. = LDR[0..]

        // per-lane path registers handed from the class code to the closest-hit query
        f3 o = mk(0, 0, 0), d = mk(0, 0, 1), w = mk(0, 0, 0);
        uint32_t depth = 0, branchf = 0, rbase = 0, k1 = 0, sp = 0;
        bool has_ray = false;
        bool retired = false;                                    // GEN only: no task left for this slot

        PH_STAMP(0)
        if (c == C_GEN) {
            // ================= GEN: continue the slot's task (smallpt.cpp:304-340, :252 pop) =================
            const uint2 ts = gtask[slot];
            uint32_t task = ts.x, snext = ts.y;
            sp = __float_as_uint(A1[slot].w) >> 30;
            // task = ((pixel * 4 + cell) << nb_log2) | block (D9); a task-less slot has snext == send == 0
            uint32_t send = 0;
            if (task != kNoTask) {
                const uint32_t sbeg = (task & ((1u << K.nb_log2) - 1u)) * K.sb;
                send = sbeg + K.sb < K.samps ? sbeg + K.sb : K.samps;
            }
            bool gen = false;
            bool need_task = false;
            if (valid) {
                if (sp > 0u) {                                   // pending transmitted child (reflected subtree is done)
                    --sp;
                    const float4* rec = stack_rec(sp, slot);
                    const float4 s0 = rec[0], s1 = rec[1], s2 = rec[2];
                    o = mk(s0.x, s0.y, s0.z); d = mk(s1.x, s1.y, s1.z); w = mk(s2.x, s2.y, s2.z);
                    const uint32_t db = __float_as_uint(s0.w);
                    depth = db & 0xFFFu; branchf = db >> 16;
                    const uint32_t k0 = __float_as_uint(s1.w);
                    k1 = __float_as_uint(s2.w);
                    rbase = rng_base(k0, branchf & 7u, depth);
                    has_ray = true;
                } else if (snext == send) {
                    need_task = true;                            // sample block finished (or the slot never had a task)
                } else {
                    gen = true;
                }
            }
            const unsigned long long need_mask = __ballot(need_task);
            if (need_mask != 0ull) {
                // cost-ordered dispatch (see chunk_order_kernel): completion times go to the chunk's clock word, fetch times below
                // (pointers rebuilt from integers are generic to the compiler: said to be global, or it emits FLAT operations, which count
                // on lgkmcnt as well and take the counted waits of the closest-hit code with them)
                typedef __attribute__((address_space(1))) uint32_t* GWords;
                uint32_t c_lo = clk_lo, c_hi = clk_hi, o_lo = ord_lo, o_hi = ord_hi;
                asm volatile("" : "+v"(c_lo), "+v"(c_hi), "+v"(o_lo), "+v"(o_hi));      // (keeps the null tests here instead of in four scalar registers across the loop)
                const GWords clk = (GWords)(((unsigned long long)c_hi << 32) | c_lo);
                uint32_t now32;
                {
                    const unsigned long long t = __builtin_amdgcn_s_memtime();
                    asm volatile("v_mov_b32 %0, %1" : "=v"(now32) : "s"((uint32_t)(t >> 6)));    // consumed here: no scalar-memory result is left in flight
                }
                if (need_task && task != kNoTask) {
                    K.cells[task] = make_float4(ACX[slot], ACY[slot], ACZ[slot], 0.0f);
                    if (clk) (void)__hip_atomic_fetch_max(clk + nch_v + (task >> 6), now32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // wave-private chunks of task ids; only the refill touches the global queue word
                const uint32_t cntn = (uint32_t)__popcll(need_mask);
                const uint32_t rk = rank_in(need_mask);
                const uint32_t avail = chunk_end - chunk_next;
                const uint32_t base_old = chunk_next;
                uint32_t base_new = 0;
                if (cntn > avail) {
                    if (!queue_empty) {
                        const int leader = __ffsll((long long)need_mask) - 1;
                        uint32_t nb = 0;
                        if ((int)lane == leader) nb = atomicAdd(K.queue, 64u);
                        base_new = uni(__shfl(nb, leader));
                        if (base_new >= K.ntasks) {
                            queue_empty = true;
                            const unsigned long long t = __builtin_amdgcn_s_memtime();
                            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(dry_lo), "=v"(dry_hi) : "s"((uint32_t)t), "s"((uint32_t)(t >> 32)));
                        } else {
                            const GWords ord = (GWords)(((unsigned long long)o_hi << 32) | o_lo);
                            if (ord) base_new = uni(ord[base_new >> 6]) << 6;        // the queue's k-th chunk of 64 tasks (declared wave-uniform: the list arithmetic stays scalar)
                            if (clk && (int)lane == leader) clk[base_new >> 6] = now32;
                        }
                    } else {
                        base_new = K.ntasks;                     // nothing left: ids >= ntasks mean "no task"
                    }
                    chunk_next = base_new + (cntn - avail);
                    chunk_end = base_new + 64u;
                    if (queue_empty) { chunk_next = chunk_end = 0; }
                } else {
                    chunk_next += cntn;
                }
                if (need_task) {
                    const uint32_t nt = rk < avail ? base_old + rk : base_new + (rk - avail);
                    if (nt < K.ntasks) {
                        task = nt; gen = true;
                        snext = (task & ((1u << K.nb_log2) - 1u)) * K.sb;
                        ACX[slot] = 0.f; ACY[slot] = 0.f; ACZ[slot] = 0.f;
                    } else {
                        retired = true;                          // the slot is pushed onto no list
                    }
                }
            }
            if (gen) {
                // ---- camera ray of sample `snext` of the cell (smallpt.cpp:325-340 / :745-760) ----
                const uint32_t cellid = task >> K.nb_log2;
                const uint32_t pix_local = cellid >> 2, cell = cellid & 3u;
                const uint32_t ry = pix_local / K.w;
                const uint32_t px = pix_local - ry * K.w;
                const uint32_t py = K.row_begin + (ry >> K.rb_log2) * K.rb_stride + (ry & K.rb_mask);   // band or interleaved row blocks
                const uint32_t pixel_idx = py * K.w + px;                    // GLOBAL index (:298)
                const uint32_t p0 = mix32(pixel_idx + K.s0);
                const uint32_t p1 = mix32(pixel_idx ^ K.s1);
                const uint32_t index_in_pixel = cell * K.samps + snext;      // :306
                const uint32_t k0 = mix32(p0 ^ (index_in_pixel * kGolden));
                k1 = mix32(p1 + index_in_pixel * 0x85EBCA6Bu);
                const float u1 = rng_draw(k0 + ((1u << 28) | 0u) * kGolden, k1);
                const float u2 = rng_draw(k0 + ((1u << 28) | 1u) * kGolden, k1);
                const uint32_t sx = cell & 1u, sy = cell >> 1;
                float ax, ay;
                if (K.sampler == 0u) {
                    const float r1 = 2 * u1;                                  // tent filter :327-330
                    const float q1 = sqrt_rsq(r1 < 1 ? r1 : 2 - r1);
                    const float dx = r1 < 1 ? q1 - 1 : 1 - q1;
                    const float r2 = 2 * u2;
                    const float q2 = sqrt_rsq(r2 < 1 ? r2 : 2 - r2);
                    const float dy = r2 < 1 ? q2 - 1 : 1 - q2;
                    // :331-332 in double like the reference; a / w as the exact Markstein sequence (tools/verify_exact_math.c)
                    const double tx = ((double)sx + .5 + (double)dx) / 2.0 + (double)px;
                    const double ty = ((double)sy + .5 + (double)dy) / 2.0 + (double)py;
                    const double qx0 = tx * K.inv_w, qy0 = ty * K.inv_h;
                    const double qx = __builtin_fma(__builtin_fma(-qx0, (double)K.w, tx), K.inv_w, qx0);
                    const double qy = __builtin_fma(__builtin_fma(-qy0, (double)K.h, ty), K.inv_h, qy0);
                    ax = (float)(qx - .5); ay = (float)(qy - .5);
                } else {
                    const float jx = ((float)sx + u1) * 0.5f, jy = ((float)sy + u2) * 0.5f;      // :750
                    const float fx = 0.5f * (2 * jx - 1), fy = 0.5f * (2 * jy - 1);              // :753-758
                    const float nx = (((float)px + 0.5f) + fx) * K.inv_wf;                       // :628-631
                    const float ny = (((float)py + 0.5f) + fy) * K.inv_hf;
                    ax = 2.f * nx - 1.f; ay = 2.f * ny - 1.f;                                    // :633
                }
                const f3 dd = cam_cx * ax + cam_cy * ay + cam_d;
                const float inv = rcp_exact(sqrt_exact(dot(dd, dd)));
                o = cam_o + dd * K.cam_push;                                                    // :333
                d = dd * inv;                                                                   // normalize(d)
                w = mk(1, 1, 1); depth = 0; branchf = 0; rbase = k0;                             // :338-339
                gtask[slot] = make_uint2(task, snext + 1u);
                has_ray = true;
            }
        } else {
            // ================= DIFF / REFR: shade the waiting hit (smallpt.cpp:170-263) =================
            const float4 a0 = A0[slot], a1 = A1[slot], a2 = A2[slot];
            const f3 hx = mk(a0.x, a0.y, a0.z);
            rbase = __float_as_uint(a0.w);
            const f3 din = mk(a1.x, a1.y, a1.z);
            const uint32_t pk = valid ? __float_as_uint(a1.w) : 0u;       // idle lanes: sphere 0, never stored
            w = mk(a2.x, a2.y, a2.z);
            k1 = __float_as_uint(a2.w);
            depth = pk & 0xFFFu; branchf = (pk >> 12) & 0xFu; sp = pk >> 30;
            const uint32_t inst = (pk >> 16) & 0xFFFu;
            const float4 gh = s_geom[inst];
            const f3 n = normalize<false>(mk(hx.x - gh.x, hx.y - gh.y, hx.z - gh.z));       // scene.cpp:124
            const f3 nl = dot(n, din) < 0 ? n : neg(n);                                     // :174 (D2)
            if (c == C_DIFF) {
                const bool is_diff = valid && ((pk >> 28) & 3u) == 0u;   // idle lanes (pk = 0) must not drag a mirror-only batch through the diffuse code
                o = hx + nl * 0.02f;                                                        // :172 (D3)
                if (is_diff) {                                                              // DIFF :208-215
                    const uint32_t u1bits = rng_draw_bits(rbase + kGolden, k1);
                    const float r2 = rng_draw(rbase + 2u * kGolden, k1);
                    const float r2s = sqrt_rsq(r2);
                    float sn, cs;
                    sincos2pi_bits(u1bits, sn, cs);                                          // D17
                    const f3 ww = nl;
                    const bool ay = __builtin_fabsf(ww.x) >= 0.1f;                          // (double)fabs(w.x) > .1, :211
                    const f3 ur = mk(ay ? ww.z : 0.f, ay ? 0.f : -ww.z, ay ? -ww.x : ww.y);
                    const float s2 = ay ? ww.x : ww.y;
                    const float qu = ww.z * ww.z + s2 * s2;
                    const f3 uu = ur * rcp_exact<false>(sqrt_rsq<true, true>(qu));
                    const f3 vv = cross(ww, uu);
                    d = normalize<false>(uu * cs * r2s + vv * sn * r2s + ww * sqrt_rsq<true, true>(1 - r2));   // :212
                } else {
                    d = din - n * 2.0f * dot(n, din);                                       // SPEC :218-223
                }
                // the weight was multiplied, the depth cap and the zero-weight cut applied before the slot was queued
                ++depth;
                rbase += 4u * kGolden;
                has_ray = valid;
            } else {
                // ---- glass, smallpt.cpp:225-263 ----
                const float4 mfc = s_mat[3 * inst + (depth > 5u ? 2 : 1)];                  // f after the roulette (:192)
                const f3 f = mk(mfc.x, mfc.y, mfc.z);
                const f3 off = nl * 0.02f;                                                  // :172 (D3)
                f3 no = hx + off, nf = f;
                f3 nd = din - n * 2.0f * dot(n, din);                                       // :218 reflRay
                const bool into = dot(n, nl) > 0;                                           // :225
                const float nnt = into ? 1.0f / 1.5f : 1.5f / 1.0f;                         // :228
                const float ddn = dot(din, nl);                                             // :229
                const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn);                        // :230
                if (valid && !(cos2t < 0)) {                                                // else TIR :232-236
                    const f3 tdir = normalize<true>(din * nnt - n * ((into ? 1.0f : -1.0f) * (ddn * nnt + sqrt_exact(cos2t)))); // :238
                    const float R0 = (0.5f * 0.5f) / (2.5f * 2.5f);                         // :240-242
                    const float cc = 1 - (into ? -ddn : dot(tdir, n));                      // :243
                    const float c2 = cc * cc;                                               // :244
                    const float Re = R0 + (1 - R0) * c2 * c2 * cc;                          // :245
                    const float Tr = 1 - Re;                                                // :246
                    const f3 xin = hx - off;                                                // D3
                    if (depth <= 2u) {                                                      // :248 split (D6)
                        const f3 tw = w * (f * Tr);
                        if (!(tw.x == 0.f && tw.y == 0.f && tw.z == 0.f)) {
                            const uint32_t br = branchf & 7u;
                            const bool nonfin = !(__builtin_fabsf(tw.x) < __builtin_inff() && __builtin_fabsf(tw.y) < __builtin_inff() && __builtin_fabsf(tw.z) < __builtin_inff());
                            float4* rec = stack_rec(sp, slot);
                            rec[0] = make_float4(xin.x, xin.y, xin.z, __uint_as_float((depth + 1u) | ((br | (1u << depth) | ((branchf & 8u) | (nonfin ? 8u : 0u))) << 16)));
                            rec[1] = make_float4(tdir.x, tdir.y, tdir.z, __uint_as_float(rbase - ((br << 29) | (depth << 2)) * kGolden));   // k0
                            rec[2] = make_float4(tw.x, tw.y, tw.z, __uint_as_float(k1));
                            rec[3] = make_float4(0.f, 0.f, 0.f, 0.f);              // completes the line: no partial-line write
                            ++sp;
                            ++nrec;
                        }
                        nf = f * Re;
                    } else {
                        const float Pr = 0.25f + 0.5f * Re;                                 // :256
                        const bool pick_refl = rng_draw(rbase + kGolden, k1) < Pr;          // :257
                        const float inv = rcp_exact(pick_refl ? Pr : 1.f - Pr);             // :259 / :263
                        nf = f * (pick_refl ? Re : Tr) * inv;
                        if (!pick_refl) { no = xin; nd = tdir; }
                    }
                }
                // extend() smallpt.cpp:120-123 + D18 + D19
                w = w * nf;
                o = no; d = nd;
                ++depth;
                rbase += 4u * kGolden;
                if (valid) {
                    if (depth >= SPT_K_MAX_DEPTH) ++nkill;
                    else if (!(w.x == 0.f && w.y == 0.f && w.z == 0.f)) has_ray = true;
                    if (!(__builtin_fabsf(w.x) < __builtin_inff() && __builtin_fabsf(w.y) < __builtin_inff() && __builtin_fabsf(w.z) < __builtin_inff())) branchf |= 8u;
                }
            }
        }

        // ================= closest hit, smallpt.cpp:54-70 over scene.cpp:129-140 (D1, D16) =================
        // Selection on integer keys key(t) = bits(t) - (bits(eps) + 1): "t > eps && t < nearest" is one unsigned
        // compare; det < 0 gives NaN keys that never win (see spt_kernel.hip phase D1).  The table is padded by the host
        // to 3 * NG spheres with never-hit entries (r*r = -inf: det = -inf, NaN keys), so the loop is fully unrolled
        // without bounds tests; ascending index with strict '<' = lowest index wins ties.
        PH_STAMP(1)
        nbounce += (unsigned long long)__popcll(__ballot(has_ray));
        uint32_t next = C_GEN;                                   // slots without a continuing path go back to GEN
        const bool queued = valid && !retired;
        uint32_t near_key = kInfKeyP, inst = 0;
        if (SPT_POOL_NARROW && 3 * NG <= 16 && b <= 4u) {
            // ---- narrow closest hit: a batch of <= 4 rays (the end of a launch: its duration is set by the last, longest
            // paths) spreads the sphere tests over the wave -- lane 16 r + i tests sphere i for ray r -- instead of
            // running all of them in <= 4 lanes: ~60 instead of ~340 instructions on the critical path.  Same keys, same
            // winner: unsigned minimum over the row, lowest sphere index among equal keys.  All 64 lanes take part. ----
            const uint32_t r = lane >> 4, si = lane & 15u;
            const f3 ro = mk(__shfl(o.x, r), __shfl(o.y, r), __shfl(o.z, r));
            const f3 rd = mk(__shfl(d.x, r), __shfl(d.y, r), __shfl(d.z, r));
            uint32_t key = 0xFFFFFFFFu;
            if (si < 3u * NG) {
                const float4 g = s_geom[si];
                const f3 op = mk(g.x - ro.x, g.y - ro.y, g.z - ro.z);                       // :132
                const float bb = dot(op, rd);                                               // :133
                const float det = bb * bb - dot(op, op) + g.w;                              // :133
                const float sd = sqrt_rsq(det);                                         // :134
                const uint32_t key1 = __float_as_uint(bb - sd) - kEpsBias;                  // :135
                const uint32_t key2 = __float_as_uint(bb + sd) - kEpsBias;
                key = key1 < key2 ? key1 : key2;
            }
            // minimum over each row of 16 lanes by DPP row rotations (no LDS round trips on the critical path)
            uint32_t m = key;
            m = umin2(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x121, 0xF, 0xF, false));   // row_ror:1
            m = umin2(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x122, 0xF, 0xF, false));   // row_ror:2
            m = umin2(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x124, 0xF, 0xF, false));   // row_ror:4
            m = umin2(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x128, 0xF, 0xF, false));   // row_ror:8
            const unsigned long long eq = __ballot(key == m);                               // row r: bits 16 r .. 16 r + 15
            const uint32_t row = (uint32_t)(eq >> ((lane & 3u) * 16u)) & 0xFFFFu;           // lane r < 4 looks at its ray's row
            const uint32_t m1 = (uint32_t)__builtin_amdgcn_readlane((int)m, 16), m2 = (uint32_t)__builtin_amdgcn_readlane((int)m, 32),
                           m3 = (uint32_t)__builtin_amdgcn_readlane((int)m, 48);
            const uint32_t mrow = lane == 1u ? m1 : (lane == 2u ? m2 : (lane == 3u ? m3 : m));   // lane 0 sits in row 0
            if (has_ray) {                                                                  // ray r lives in lane r
                near_key = mrow < kInfKeyP ? mrow : kInfKeyP;
                inst = (uint32_t)__ffs((int)row) - 1u;                                      // :61 strict <: lowest index wins ties
            }
        } else if (has_ray) {
            // nk[i + 1] = min(nk[i], key1, key2) of sphere i: one v_min3_u32 per sphere; the index of the winner is
            // recovered afterwards as the last i at which the running minimum changed = the lowest index among equal
            // nearest distances (:61 strict <).
            uint32_t nk[3 * NG + 1];
            nk[0] = kInfKeyP;
            // the wave-uniform (broadcast) LDS reads of up to nine spheres are issued together ahead of their arithmetic
#pragma unroll
            for (int base = 0; base < 3 * NG; base += 9) {
                float4 g[9];
#pragma unroll
                for (int j = 0; j < 9; ++j)
                    if (base + j < 3 * NG) g[j] = s_geom[base + j];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    if (base + j < 3 * NG) {
                        const f3 op = mk(g[j].x - o.x, g[j].y - o.y, g[j].z - o.z);         // :132
                        const float bb = dot(op, d);                                        // :133
                        const float det = bb * bb - dot(op, op) + g[j].w;                   // :133 (g.w = r*r)
                        const float sd = sqrt_rsq(det);                                 // :134
                        const uint32_t key1 = __float_as_uint(bb - sd) - kEpsBias;          // :135
                        const uint32_t key2 = __float_as_uint(bb + sd) - kEpsBias;
                        nk[base + j + 1] = umin3(nk[base + j], key1, key2);
                    }
                }
            }
            near_key = nk[3 * NG];
#pragma unroll
            for (int i = 1; i < 3 * NG; ++i)
                if (nk[i + 1] != nk[i]) inst = (uint32_t)i;
        }
        PH_STAMP(2)
        if (has_ray) {
            // ---- class-independent part of shadePaths (smallpt.cpp:168-198) ----
            if (near_key != kInfKeyP) {                                                     // else :168 miss (D13)
                const float t = __uint_as_float(near_key + kEpsBias);
                const float4 me = s_mat[3 * inst + 0];                                      // emission.xyz, refl | emissive << 2
                const float4 mc = s_mat[3 * inst + 1];                                      // color.xyz, pmax
                const float4 mf = s_mat[3 * inst + 2];                                      // color * (1/pmax), :192: read with the others (one LDS round trip)
                const uint32_t rb = __float_as_uint(me.w);
                const uint32_t refl = rb & 3u;
                if ((rb & 4u) != 0u || (branchf & 8u) != 0u) {                              // :179 (D4); + w*0 is skipped, exact for finite w
                    ACX[slot] = ACX[slot] + w.x * me.x; ACY[slot] = ACY[slot] + w.y * me.y; ACZ[slot] = ACZ[slot] + w.z * me.z;
                }
                f3 f = mk(mc.x, mc.y, mc.z);                                                // :175
                bool cont = true;
                if (depth > 5u) {                                                           // :188 (D5)
                    if (rng_draw(rbase, k1) < mc.w) {
                        f = mk(mf.x, mf.y, mf.z);
                    } else {
                        cont = false;                                                       // :196
                    }
                }
                if (cont) {
                    if (refl != 2u) {
                        // extend() of the DIFF / SPEC child (:214,:221): weight, D18 depth cap, D19 zero-weight cut
                        w = w * f;
                        if (depth + 1u >= SPT_K_MAX_DEPTH) { ++nkill; cont = false; }
                        else if (w.x == 0.f && w.y == 0.f && w.z == 0.f) cont = false;
                    }
                    if (cont) {
                        const f3 hx = o + d * t;                                            // scene.cpp:137
                        A0[slot] = make_float4(hx.x, hx.y, hx.z, __uint_as_float(rbase));
                        A1[slot] = make_float4(d.x, d.y, d.z, __uint_as_float(pack_path(depth, branchf, inst, refl, sp)));
                        A2[slot] = make_float4(w.x, w.y, w.z, __uint_as_float(k1));
                        next = refl == 2u ? C_REFR : C_DIFF;
                    }
                }
            }
        }
        PH_STAMP(3)
        // ================= push every slot onto the list of its next class =================
        {
            const bool to_gen = queued && next == C_GEN;
            const unsigned long long mg = __ballot(to_gen);
            const unsigned long long md = __ballot(queued && next == C_DIFF);
            const unsigned long long mr = __ballot(queued && next == C_REFR);
            uint32_t pos = nG + rank_in(mg);
            if (next == C_DIFF) pos = (uint32_t)P + nD + rank_in(md);
            if (next == C_REFR) pos = 2u * (uint32_t)P - 1u - nR - rank_in(mr);
            if (queued) LG[pos] = (uint8_t)slot;
            // a slot whose path ended keeps the count of its sample's pending transmitted children for the next GEN visit
            if (to_gen) A1[slot].w = __uint_as_float(sp << 30);
            nG += (uint32_t)__popcll(mg);
            nD += (uint32_t)__popcll(md);
            nR += (uint32_t)__popcll(mr);
        }
        PH_STAMP(4)
    }

    // stats: one atomic per wave
    unsigned long long nk = nkill, nr = nrec;
    for (int off = 32; off > 0; off >>= 1) { nk += __shfl_down(nk, off); nr += __shfl_down(nr, off); }
    if (lane == 0) {
        atomicAdd(&K.counters[0], nbounce);
        if (nk) atomicAdd(&K.counters[1], nk);
        atomicAdd(&K.counters[2], (unsigned long long)itG); atomicAdd(&K.counters[3], (unsigned long long)itD);
        atomicAdd(&K.counters[4], (unsigned long long)itR);
        atomicAdd(&K.counters[5], (unsigned long long)lnG); atomicAdd(&K.counters[6], (unsigned long long)lnD); atomicAdd(&K.counters[7], (unsigned long long)lnR);
        if (timed_out) atomicAdd(&K.counters[8], 1ull);
        atomicAdd(&K.counters[9], (unsigned long long)itTail); atomicAdd(&K.counters[10], (unsigned long long)lnTail);
        atomicAdd(&K.counters[11], (unsigned long long)itFull);
        // launch timeline in s_memtime ticks, wave-local differences only (the counter is not synchronised across XCDs):
        // longest wave, longest and summed time a wave kept running after it found the task queue empty
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        atomicMax(&K.counters[12], t_end - t_start);
        const unsigned long long t_dry = ((unsigned long long)uni(dry_hi) << 32) | uni(dry_lo);
        if (t_dry) { atomicMax(&K.counters[13], t_end - t_dry); atomicAdd(&K.counters[14], t_end - t_dry); }
        atomicAdd(&K.counters[15], t_end - t_start);
#ifdef SPT_POOL_PHASES
        for (int i = 0; i < 5; ++i) atomicAdd(&K.counters[17 + i], ph[i]);
        atomicAdd(&K.counters[22], (unsigned long long)it_total);
#endif
        atomicAdd(&K.counters[16], nr);                           // pending-child records written (64 B out, 64 B back in each)
    }
}

}  // namespace spt

// LDS per 256-thread workgroup: four wave pools + the padded scene table (16 B geometry + 48 B material per sphere)
namespace spt {
// Chunk order for the next launch of the same view: chunks by the time their wave spent on them (fetch to last completion), longest
// first -- 256 logarithmic buckets (8 per octave), one workgroup.  A launch ends on the blocks that were started last, and the cost of
// a block varies by two orders of magnitude with the pixel (every sample of a pixel that looks at the mirror ball's contact point
// runs ~1000 bounces: such a 32-sample block keeps one slot busy for ~20 ms of an 80 ms launch); started first, they end with the rest.
// The order inside a bucket is whatever the atomics give: dispatch order never changes a result.
__device__ __forceinline__ uint32_t cost_bucket(uint32_t cost)
{
    const uint32_t lg = 31u - (uint32_t)__builtin_clz(cost | 1u);
    const uint32_t frac = lg >= 3u ? (cost >> (lg - 3u)) & 7u : 0u;
    return 255u - (lg * 8u + frac);
}

// n chunks, the first `sorted` of them ordered (the last chunk of a task count that is no multiple of 64 stays last: a slot that is
// handed an id beyond the last task retires for the rest of the launch).  Three small launches over slices of kOrderSlice chunks --
// bucket counts, their prefix sums, scatter; `work` = 256 global counters + 256 offsets -- 15 us for the
// 393 216 chunks of config 2 (one workgroup doing all of it with LDS atomics took 350 us, behind every launch).
constexpr uint32_t kOrderSlice = 4096;

__global__ __launch_bounds__(1024) void chunk_count_kernel(const uint32_t* __restrict__ clock, uint32_t n, uint32_t sorted, uint32_t* __restrict__ work)
{
    __shared__ uint32_t hist[256];
    if (threadIdx.x < 256u) hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t first = blockIdx.x * kOrderSlice, last = first + kOrderSlice < sorted ? first + kOrderSlice : sorted;
    for (uint32_t i = first + threadIdx.x; i < last; i += 1024u) atomicAdd(&hist[cost_bucket(clock[n + i] - clock[i])], 1u);
    __syncthreads();
    if (threadIdx.x < 256u && hist[threadIdx.x]) atomicAdd(&work[threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void chunk_scan_kernel(uint32_t* __restrict__ work)
{
    __shared__ uint32_t cnt[256];
    cnt[threadIdx.x] = work[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t run = 0u;
        for (int b = 0; b < 256; ++b) { const uint32_t c = cnt[b]; cnt[b] = run; run += c; }
    }
    __syncthreads();
    work[256 + threadIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(1024) void chunk_scatter_kernel(const uint32_t* __restrict__ clock, uint32_t n, uint32_t sorted, uint32_t* __restrict__ work,
                                                             uint32_t* __restrict__ order)
{
    __shared__ uint32_t hist[256];
    if (threadIdx.x < 256u) hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t first = blockIdx.x * kOrderSlice, last = first + kOrderSlice < sorted ? first + kOrderSlice : sorted;
    for (uint32_t i = first + threadIdx.x; i < last; i += 1024u) atomicAdd(&hist[cost_bucket(clock[n + i] - clock[i])], 1u);
    __syncthreads();
    if (threadIdx.x < 256u) { const uint32_t c = hist[threadIdx.x]; hist[threadIdx.x] = c ? atomicAdd(&work[256 + threadIdx.x], c) : 0u; }   // this slice's range in every bucket
    __syncthreads();
    for (uint32_t i = first + threadIdx.x; i < last; i += 1024u) order[atomicAdd(&hist[cost_bucket(clock[n + i] - clock[i])], 1u)] = i;
    if (blockIdx.x == 0u) for (uint32_t i = sorted + threadIdx.x; i < n; i += 1024u) order[i] = i;
}
}  // namespace spt

extern "C" hipError_t spt_pool_chunk_order(const uint32_t* chunk_clock, uint32_t nchunks, uint32_t ntasks, uint32_t* chunk_order, uint32_t* work512, hipStream_t stream)
{
    const uint32_t sorted = (ntasks & 63u) != 0u && nchunks > 0u ? nchunks - 1u : nchunks;
    const uint32_t slices = sorted ? (sorted + spt::kOrderSlice - 1u) / spt::kOrderSlice : 1u;
    hipError_t e = hipMemsetAsync(work512, 0, 256 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(spt::chunk_count_kernel, dim3(slices), dim3(1024), 0, stream, chunk_clock, nchunks, sorted, work512);
    hipLaunchKernelGGL(spt::chunk_scan_kernel, dim3(1), dim3(256), 0, stream, work512);
    hipLaunchKernelGGL(spt::chunk_scatter_kernel, dim3(slices), dim3(1024), 0, stream, chunk_clock, nchunks, sorted, work512, chunk_order);
    return hipGetLastError();
}

extern "C" size_t spt_pool_lds_bytes(uint32_t n, int pool)
{
    const uint32_t ng = n == 0 ? 1u : (n + 2u) / 3u;
    return (size_t)(spt::kPoolBlock / 64) * (size_t)spt::kSlotBytes * (size_t)pool + (size_t)ng * 3u * 64u;
}

extern "C" size_t spt_pool_stack_floats(uint32_t blocks, int pool)
{
    return (size_t)blocks * (spt::kPoolBlock / 64) * 3u * spt::kStackWords * (size_t)pool;
}

extern "C" size_t spt_pool_state_bytes(uint32_t blocks, int pool) { return (size_t)blocks * (spt::kPoolBlock / 64) * (size_t)pool * sizeof(uint2); }

extern "C" int spt_pool_max_spheres(void) { return spt::kMaxUnroll; }
// 160 slots per wave with the 62-byte slot, 144 with the 70-byte slot: four workgroups (16 waves) per CU either way
extern "C" int spt_pool_default_slots(void) { return spt::kTaskLds ? 144 : 160; }

template <int P, int NG>
static hipError_t launch_pool(const spt::KParams* K, uint32_t blocks, size_t lds, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spt::poolkernel<P, NG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((spt::poolkernel<P, NG>), dim3(blocks), dim3(spt::kPoolBlock), lds, stream, *K);
    return hipGetLastError();
}

template <int P>
static hipError_t launch_pool_ng(const spt::KParams* K, uint32_t blocks, size_t lds, hipStream_t stream)
{
    switch (K->n == 0 ? 1u : (K->n + 2u) / 3u) {
    case 1: return launch_pool<P, 1>(K, blocks, lds, stream);
    case 2: return launch_pool<P, 2>(K, blocks, lds, stream);
    case 3: return launch_pool<P, 3>(K, blocks, lds, stream);
    case 4: return launch_pool<P, 4>(K, blocks, lds, stream);
    case 5: return launch_pool<P, 5>(K, blocks, lds, stream);
    case 6: return launch_pool<P, 6>(K, blocks, lds, stream);
    case 7: return launch_pool<P, 7>(K, blocks, lds, stream);
    case 8: return launch_pool<P, 8>(K, blocks, lds, stream);
    default: return hipErrorInvalidValue;
    }
}

// pool sizes this build carries (spt_set_tuning refuses the others up front)
extern "C" int spt_pool_has_size(int pool)
{
#ifdef SPT_POOL_SIZES
    if (pool == 96 || pool == 192) return 1;
#endif
    return pool == 128 || pool == 144 || pool == 160;
}

extern "C" hipError_t spt_pool_launch(const spt::KParams* K, uint32_t blocks, int pool, hipStream_t stream)
{
    const size_t lds = spt_pool_lds_bytes(K->n, pool);
    if (pool == 128) return launch_pool_ng<128>(K, blocks, lds, stream);
    if (pool == 160) return launch_pool_ng<160>(K, blocks, lds, stream);
    if (pool == 144) return launch_pool_ng<144>(K, blocks, lds, stream);
#ifdef SPT_POOL_SIZES
    if (pool == 96) return launch_pool_ng<96>(K, blocks, lds, stream);
    if (pool == 192) return launch_pool_ng<192>(K, blocks, lds, stream);
#endif
    return hipErrorInvalidValue;
}
