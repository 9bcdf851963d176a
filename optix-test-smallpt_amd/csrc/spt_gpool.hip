// spt_gpool.hip -- persistent path-tracing kernel for large sphere tables on gfx950 (MI355X), round 4: the uniform grid of spt_grid.h
// (exhaustive-equivalent closest hit of intersectGlobalSpheres, smallpt.cpp:54-70 over scene.cpp:129-140) driven by wave-private
// PATH POOLS instead of lanes that own their path (spt_grid.hip, which stays for tables that leave no LDS for the pools).
//
// Why: in spt_grid.hip a lane carries one path through regenerate -> begin -> walk -> shade, so at any moment about half the lanes
// of a wave walk while the others wait to be shaded, and the walkers split between the TEST and the STEP body: 45 % lane
// utilisation (profiles/r03_config5_pmc_summary.txt).  Here the three kinds of work run on different sets of lanes:
//
//   * every WAVE owns S path slots.  A slot is one task (one D9 block of a jitter cell, smallpt.cpp:299-309) with at most one path
//     in flight, so emission events are accumulated in the order of D9.  A slot's state -- origin, direction, weight, RNG keys, depth,
//     task, block sum: 96 bytes in a 128-byte line -- lives in GLOBAL memory (wave-private lines, L2 / Infinity-Cache resident); LDS holds the grid
//     tables (one copy per CU, as in spt_grid.hip), per wave a stack of R begun walks (READY, 56 bytes each) and byte lists of slot ids.
//   * the 64 lanes of the wave are WALKERS: a lane holds one walk in registers (ray, exit parameters, cell, nearest key) and runs
//     the fused body { leave the cell if all its spheres are tested; test the next sphere } until its walk ends.  Finished lanes
//     write (key, index) to their slot, queue it for shading by material class and take the next begun walk from READY, `drain`
//     lanes at a time, so the walk loop runs with (almost) every lane walking -- whatever the shading side is doing.
//   * shading runs in BATCHES of up to 64 slots of one class with every lane active, as in spt_pool.hip: HIT (DIFF / SPEC,
//     smallpt.cpp:208-223), HITR (glass, :225-263), GEN (next camera sample / pending transmitted child / new task,
//     :304-340, :252).  The batch's lanes then BEGIN the new ray -- ray test, always-tested spheres (the walls and the light of a
//     Cornell box), walk set-up (spt_grid.h) -- and push it onto READY.
//
// RNG (D7), summation order (D9), sin/cos (D17), depth cap (D18), zero-weight cut (D19) and every arithmetic expression are those
// of spt_grid.hip / spt_pool.hip / the oracle: results are bit-identical, bounce counts included.
#include "spt_device.h"
#define SPT_GRID_DEVICE_ONLY
#include "spt_grid.h"
#include "spt_kernel.h"

namespace spt {

constexpr int kQBlock = 1024;                                    // threads per workgroup (one workgroup per CU shares the LDS tables)
constexpr uint32_t kQEpsBias = 0x38D1B717u + 1u;                 // bits(1e-4f) + 1
constexpr uint32_t kQInfKey = 0x60AD78ECu - kQEpsBias;           // key of 1e20f (maths.h:16)
#ifndef QX_SLOT_F4
#define QX_SLOT_F4 8
#endif
// float4 per slot in global memory: 6 are used, 8 make a slot exactly one 128-byte line.  At 96 bytes half the slots straddle two lines and
// a batch fetched 1.5 lines per slot through the fabric (the 75 MB of slots are beyond the L2): 128-byte slots are 4.6 % faster (383 -> 366 ms
// on config 5) although the working set grows to 100 MB -- the kernel feels its 5 TB/s of slot traffic.
constexpr int kQSlotF4 = QX_SLOT_F4;
constexpr uint32_t kQSlotBytes = 16u * kQSlotF4;
constexpr uint32_t kQFew = 4u;                                   // a batch that leaves at most this many rays answers them with the whole wave
constexpr uint32_t kQFin = 0xFFFFu;                              // staged header of a border cell / the walker's cur when its walk has ended
constexpr int kQStackF4 = 4;                                     // one pending child = one 64-byte line

__device__ __forceinline__ uint32_t lane_id_q() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t rank_q(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t uniq(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// intersectAnalytic of one sphere record {c, r*r} on integer keys (scene.cpp:129-140, smallpt.cpp:59-65), as in spt_grid.hip
__device__ __forceinline__ uint32_t sphere_key_q(const float4 g, f3 o, f3 d)
{
    const f3 op = mk(g.x - o.x, g.y - o.y, g.z - o.z);                                  // :132
    const float bb = dot(op, d);                                                        // :133
    const float det = bb * bb - dot(op, op) + g.w;                                      // :133 (g.w = r*r)
    const float sd = sqrt_rsq(det);                                                     // :134
    const uint32_t key1 = __float_as_uint(bb - sd) - kQEpsBias;                         // :135
    const uint32_t key2 = __float_as_uint(bb + sd) - kQEpsBias;
    return key1 < key2 ? key1 : key2;
}

// LDS reads at a byte address the lane holds in a register (the kernel's dynamic LDS starts at address 0 -- this translation unit has no
// static __shared__ --, so offsets into s_lds ARE addresses; spt_gpool_launch checks it): no base is added, ds_read_u16 zero-extends.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float __attribute__((ext_vector_type(4))) lds_v4f;
// (ds_read_u16 written out: the compiler's own selection masks the zero-extended result once more; the wait inside the statement keeps
// the compiler's counted waits conservative)
__device__ __forceinline__ uint32_t lds_u16(uint32_t a) { uint32_t v; asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); return v; }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)a; }
__device__ __forceinline__ float4 lds_f4(uint32_t a)
{
    const lds_v4f v = *(const __attribute__((address_space(3))) lds_v4f*)(uintptr_t)a;
    return make_float4(v.x, v.y, v.z, v.w);
}
#else
__device__ __forceinline__ uint32_t lds_u16(uint32_t) { return 0u; }
__device__ __forceinline__ uint32_t lds_u32(uint32_t) { return 0u; }
__device__ __forceinline__ float4 lds_f4(uint32_t) { return make_float4(0.f, 0.f, 0.f, 0.f); }
#endif

// The exhaustive loop of smallpt.cpp:54-70 for ONE ray by the whole wave: lane l tests spheres l, l + 64, ... in ascending order with
// strict '<' (the lowest index among a lane's equal keys), then the wave takes the lexicographic minimum of (key, index) -- the key the
// sequential loop ends with and, among the spheres that produce it, the lowest index (:61).  o / d are wave-uniform.  Used where a wave
// has only a few rays to answer (the end of a launch, a lone 4096-bounce path in a closed white ball): n / 64 tests per lane instead
// of the always-tested list and a walk in one lane.
__device__ __forceinline__ void wave_closest_sphere(const float4* geom, uint32_t n, f3 o, f3 d, uint32_t lane, uint32_t& key_out, uint32_t& idx_out)
{
    uint32_t bk = kQInfKey, bi = 0u;
    for (uint32_t i = lane; i < n; i += 64u) {
        const uint32_t key = sphere_key_q(geom[i], o, d);
        if (key < bk) { bk = key; bi = i; }
    }
#pragma unroll 1
    for (int off = 32; off > 0; off >>= 1) {                     // (rolled: this is the rare path, the main loop keeps its registers)
        const uint32_t k2 = (uint32_t)__shfl_xor((int)bk, off), i2 = (uint32_t)__shfl_xor((int)bi, off);
        const bool better = (k2 < bk) | ((k2 == bk) & (i2 < bi));
        bk = better ? k2 : bk; bi = better ? i2 : bi;
    }
    key_out = bk; idx_out = bk == kQInfKey ? 0u : bi;            // (a miss keeps index 0 like the sequential loop)
}

// packed word of a slot (Q1.w): [11:0] depth, [14:12] branch bits (D7), [15] weight-may-be-non-finite flag, [31:30] pending
// transmitted children of the slot's current sample
__device__ __forceinline__ uint32_t pack_q(uint32_t depth, uint32_t branchf, uint32_t sp) { return depth | (branchf << 12) | (sp << 30); }

enum { QC_GEN = 0, QC_HIT = 1, QC_HITR = 2 };

// Global slot record (6 float4 of a 128-byte line).  What every bounce reads and writes sits in the line's FIRST 64-byte sector:
//   Q0 {o.xyz, t_ok}  Q1 {d.xyz, packed}  Q2 {w.xyz, k1}  Q3 {rbase, -, near key, near index};
// what only the GEN batches (one in nine) and emissive hits touch in the second: Q4 {task + 1, next sample, -, -}  Q5 {block sum xyz, -}.
// (old layout, for the record:) Q3 {rbase, task + 1, next sample, -}
//                                Q4 {near key, near index, -, -}  Q5 {block sum xyz, -}
template <bool STATS>
__global__ __launch_bounds__(kQBlock) void gpoolkernel(const KParams K, const GridParams G, const uint32_t* __restrict__ g_cells,
                                                       const uint16_t* __restrict__ g_refs, const uint32_t* __restrict__ g_always, const QParams Q)
{
    // LDS of the workgroup: [cell references, always-tested list][sphere records][cell headers][materials][wave regions].
    // The references come first so that a walker's cur / end are BYTE addresses that fit the 16-bit halves of a staged cell header, and
    // a reference holds sphere index + gb16 (gb16 = offset of the sphere records / 16): the record's address is reference << 4, one
    // full-rate instruction, and references order like sphere indices (the lowest-index rule compares them as they are).
    extern __shared__ char s_lds[];
    const uint32_t ngeom = G.n ? G.n : 1u;
    const uint32_t geom_off = ((G.nrefs + G.nalways + 1u) * 2u + 15u) & ~15u;
    const uint32_t gb16 = geom_off >> 4;
    const uint32_t cells_off = geom_off + ngeom * 16u;
    const uint32_t mat_off = (cells_off + G.ncells * 4u + 15u) & ~15u;
    const uint32_t tables_end = mat_off + ngeom * 16u;
    uint16_t* const s_refs = reinterpret_cast<uint16_t*>(s_lds);                  // nrefs cell references, then the always-tested list, one spare
    float4* const s_geom = reinterpret_cast<float4*>(s_lds + geom_off);          // n x {centre, r * r}
    uint32_t* const s_cells = reinterpret_cast<uint32_t*>(s_lds + cells_off);    // staged cell headers (below)
    // materials: {color.xyz, Refl_t | emissive << 2} per sphere (16 of the host table's 48 bytes: pmax = fmaxf(color) and color * (1 / pmax)
    // are single IEEE operations that the shading batch repeats bit for bit; emission is read from global memory for emissive spheres only)
    float4* const s_mat = reinterpret_cast<float4*>(s_lds + mat_off);
    for (uint32_t i = threadIdx.x; i < G.n; i += blockDim.x) {
        s_geom[i] = K.geom[i];
        const float4 mc = K.mat[3 * i + 1];
        s_mat[i] = make_float4(mc.x, mc.y, mc.z, K.mat[3 * i].w);
    }
    // cell headers in the walker's form: byte address of the first reference | byte address behind the last << 16 -- the lane's
    // [cur, end) after two instructions --; a border cell reads kQFin: cur = 0xFFFF > end = 0, "the walk has ended" (the launch
    // checks 2 (nrefs + 1) < 0xFFFF)
    for (uint32_t i = threadIdx.x; i < G.ncells; i += blockDim.x) {
        const uint32_t h = g_cells[i];
        const uint32_t first = h >> kGridCountBits, cnt = h & ((1u << kGridCountBits) - 1u);
        s_cells[i] = h == kGridBorder ? kQFin : ((2u * first) | ((2u * (first + cnt)) << 16));
    }
    for (uint32_t i = threadIdx.x; i < G.nrefs; i += blockDim.x) s_refs[i] = (uint16_t)(g_refs[i] + gb16);
    for (uint32_t i = threadIdx.x; i <= G.nalways; i += blockDim.x) s_refs[G.nrefs + i] = (uint16_t)((i < G.nalways ? g_always[i] : 0u) + gb16);

    const uint32_t lane = lane_id_q();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t S = Q.S, R = Q.R;
    const uint32_t wave_bytes = R * 64u + 2u * S;
    float4* const RD0 = reinterpret_cast<float4*>(s_lds + tables_end + wave * wave_bytes);             // {o.xyz, near key}
    float4* const RD1 = RD0 + R;                                 // {d.xyz, near reference (index + gb16) | slot << 16}
    float4* const RD2 = RD1 + R;                                 // {tx, ty, tz, byte address of the cell header}
    float4* const RD3 = RD2 + R;                                 // {dtx, dty, dtz, header of the walk's start cell}
    uint8_t* const LH = reinterpret_cast<uint8_t*>(RD3 + R);    // S bytes: HIT list from index 0 up, HITR list from S - 1 down
    uint8_t* const LGN = LH + S;                                 // S bytes: GEN list
    const uint32_t wave_gid = blockIdx.x * (blockDim.x >> 6) + wave;
    float4* const slots = Q.slots + (size_t)wave_gid * S * kQSlotF4;
    char* const slot_bytes = reinterpret_cast<char*>(slots);     // (32-bit byte offsets from a wave-uniform base: no 64-bit vector arithmetic)
    float4* const gstack = reinterpret_cast<float4*>(K.stack) + (size_t)wave_gid * S * (3 * kQStackF4);
    auto stack_rec = [&](uint32_t e, uint32_t slot) -> float4* { return gstack + (slot * 3u + e) * kQStackF4; };

    // every slot starts on the GEN list as a finished, task-less slot
    for (uint32_t s = lane; s < S; s += 64u) {
        LGN[s] = (uint8_t)s;
        slots[s * kQSlotF4 + 1] = make_float4(0.f, 0.f, 1.f, __uint_as_float(0u));
        slots[s * kQSlotF4 + 4] = make_float4(__uint_as_float(0u), __uint_as_float(0u), 0.f, 0.f);
    }
    __syncthreads();

    // ---- wave-uniform state ----
    uint32_t nR = 0, nH = 0, nHR = 0, nG = S;                    // READY records, HIT / HITR / GEN list lengths
    uint32_t chunk_next = 0, chunk_end = 0;                      // this wave's private range of task ids
    bool queue_empty = false;
    unsigned long long nbounce = 0;
    uint32_t nkill = 0;                                          // per lane
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    bool timed_out = false;
    uint32_t n_loop = 0;
    // statistics (STATS build)
    // (32-bit per wave -- a launch of the instrumented build lasts less than 2^32 clocks --: as 64-bit scalars they cost spill lanes and scratch)
    uint32_t st_iter = 0, st_act = 0, st_exch = 0, st_redo = 0;
    uint32_t st_test = 0, st_step = 0;                           // per lane
    uint32_t st_bat[3] = {0, 0, 0}, st_lan[3] = {0, 0, 0};
    uint32_t ph[4] = {0, 0, 0, 0}, ph_t = 0;                     // wave time: walk, exchange, generation batches, shading batches
#define QSTAMP(i) if (STATS) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime(); ph[i] += t_ - ph_t; ph_t = t_; }

    // ---- the lane's walk (registers): the fields of a GridWalk (spt_grid.h), the ray, the nearest (key, index) so far and the references
    // [cur, end) of the current cell that are still to test (byte addresses in s_refs).  cur == end: the cell is exhausted, the lane
    // steps; cur > end: the lane is not walking -- (1, 0): it holds no walk, (kQFin, 0): its walk has ended and waits for the exchange.
    // wci is the byte address of the current cell's header and wsx / wsy / wsz step it; near_i is a reference (sphere index + gb16). ----
    f3 wo = mk(0, 0, 0), wd = mk(0, 0, 1);
    float wtx = 0.f, wty = 0.f, wtz = 0.f, wdx = 0.f, wdy = 0.f, wdz = 0.f;
    int32_t wsx = 0, wsy = 0, wsz = 0;
    uint32_t wci = 0, cur = 1, end = 0, near_key = kQInfKey, near_i = 0, wslot = 0;

    // the index steps a walker lane needs when it takes a begun walk over: in VECTOR registers (in scalar ones they push the loop's
    // masks into spill lanes; re-read from the kernel-argument segment they put a scalar-memory wait into every exchange)
    int32_t stride_y = 4 * G.stride_y, stride_z = 4 * G.stride_z;      // (of byte addresses)
    asm volatile("" : "+v"(stride_y), "+v"(stride_z));

    if (STATS) ph_t = (uint32_t)__builtin_amdgcn_s_memtime();
    for (;;) {
        if ((++n_loop & 63u) == 0u && K.watchdog_ticks != 0ull && __builtin_amdgcn_s_memtime() - t_start > K.watchdog_ticks) { timed_out = true; break; }
        uint32_t nAct = (uint32_t)__popcll(__ballot(cur <= end));
        uint32_t nFin = (uint32_t)__popcll(__ballot(cur == kQFin));
        const uint32_t nEmp = 64u - nAct - nFin;
        bool idle = true;                                        // nothing was done in this round: the wave's tasks are finished

        if ((nFin != 0u && (nFin >= Q.drain || nAct == 0u)) || (nR != 0u && nEmp + nFin != 0u && (nEmp + nFin >= Q.drain || nAct == 0u))) {
            // =============== exchange: finished walkers hand their hits over, empty lanes take begun walks ===============
            if (STATS) ++st_exch;
            idle = false;
            // (finished lanes are empty lanes already -- cur > end --, so the begun walks can be fetched before the hits are handed over:
            // the LDS reads of both halves are in flight together)
            const unsigned long long me = __ballot(cur > end);
            const uint32_t ne = (uint32_t)__popcll(me);
            const uint32_t k = ne < nR ? ne : nR;
            const uint32_t rk = rank_q(me);
            const bool take = cur > end && rk < k;
            const uint32_t pos = nR - 1u - rk;
            float4 r0, r1, r2, r3;                               // (read and used by the taking lanes only)
            if (take) { r0 = RD0[pos]; r1 = RD1[pos]; r2 = RD2[pos]; r3 = RD3[pos]; }
            uint32_t hcls = 0u;                                  // 1: onto HIT, 2: onto HITR
            if (cur == kQFin) {
                cur = 1u;                                        // (end is 0 already)
                const uint32_t inst = near_i - gb16;
                *reinterpret_cast<uint2*>(slot_bytes + wslot * kQSlotBytes + 56u) = make_uint2(near_key, inst);
                hcls = (near_key != kQInfKey && (__float_as_uint(s_mat[inst].w) & 3u) == 2u) ? 2u : 1u;
            }
            const unsigned long long mh = __ballot(hcls == 1u), mr = __ballot(hcls == 2u);
            if (hcls != 0u) LH[hcls == 2u ? S - 1u - nHR - rank_q(mr) : nH + rank_q(mh)] = (uint8_t)wslot;
            nH += (uint32_t)__popcll(mh); nHR += (uint32_t)__popcll(mr);
            if (take) {
                // a begun walk as grid_walk_begin left it (spt_grid.h); the index steps follow from the direction's signs as there
                wo = mk(r0.x, r0.y, r0.z); near_key = __float_as_uint(r0.w);
                wd = mk(r1.x, r1.y, r1.z);
                const uint32_t pk = __float_as_uint(r1.w);
                near_i = pk & 0xFFFFu; wslot = pk >> 16;
                wtx = r2.x; wty = r2.y; wtz = r2.z; wci = __float_as_uint(r2.w);
                wdx = r3.x; wdy = r3.y; wdz = r3.z;
                wsx = !(wd.x > 0.0f) ? -4 : 4; wsy = !(wd.y > 0.0f) ? -stride_y : stride_y; wsz = !(wd.z > 0.0f) ? -stride_z : stride_z;
                const uint32_t h0 = __float_as_uint(r3.w);       // the start cell (never a border cell)
                cur = h0 & 0xFFFFu; end = h0 >> 16;
            }
            nR -= k;
            nAct += k; nFin = 0u;
            QSTAMP(1)
        }

        const uint32_t nFree = R - nR;
        const bool starving = nR < Q.drain && nAct + Q.drain <= 64u;
        const uint32_t aH = nH < nFree ? nH : nFree, aHR = nHR < nFree ? nHR : nFree, aG = nG < nFree ? nG : nFree;
        uint32_t cls = QC_HITR, amax = aHR;
        if (aH > amax) { cls = QC_HIT; amax = aH; }
        if (aG > amax) { cls = QC_GEN; amax = aG; }
        const uint32_t cap = amax < 64u ? amax : 64u;            // lanes of the largest batch available
        const uint32_t minb = Q.min_batch < 64u ? Q.min_batch : 64u;

        // =============== a batch of up to 64 slots of one class: pop the slots and issue the loads of their state; the walk below runs
        // while they are in flight, the batch's code after it ===============
        const bool run_batch = cap != 0u && (cap >= (starving ? minb : 64u) || nAct == 0u);
        const uint32_t b = run_batch ? cap : 0u;
        const bool valid = lane < b;
        uint32_t slot = 0;
        float4 q0 = make_float4(0.f, 0.f, 0.f, __builtin_inff()), q1 = make_float4(0.f, 0.f, 1.f, 0.f), q2 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 q3 = make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 q4 = make_uint2(kQInfKey, 0u), qt = make_uint2(0u, 0u);
        if (run_batch) {
            idle = false;
            if (cls == QC_GEN) { nG -= b; if (valid) slot = LGN[nG + lane]; }
            else if (cls == QC_HIT) { nH -= b; if (valid) slot = LH[nH + lane]; }
            else { if (valid) slot = LH[S - nHR + lane]; nHR -= b; }
            if (STATS) { ++st_bat[cls]; st_lan[cls] += b; }
            const float4* const lq = slots + slot * kQSlotF4;
            // (every lane loads -- idle lanes read slot 0, never use it --: a merge with default values would need the data at once;
            // and every class loads the five rows (a GEN batch needs two of them): a merge at a branch would, too)
            q0 = lq[0]; q1 = lq[1]; q2 = lq[2];
            q3.x = reinterpret_cast<const float*>(lq + 3)[0];                             // rbase (a register loaded and never read would be reused by the walk: a wait)
            q4 = reinterpret_cast<const uint2*>(lq + 3)[1];                               // near key, near index
            qt = reinterpret_cast<const uint2*>(lq + 4)[0];                               // task + 1, next sample (GEN batches)
        }

        if (nAct != 0u) {
            // =============== walk: the fused STEP + TEST body until only `thr` lanes are left walking ===============
            // (the walk stops for an exchange -- `drain` lanes finished, or that many free for the begun walks that wait -- or for a
            // batch that is worth running once the walkers starve; all of these are "at most thr lanes still walk".  With a batch
            // pending it lasts a few iterations only: as long as the batch's loads are in flight.)
            idle = false;
            uint32_t thr = 0u;
            if (nAct + nFin > Q.drain) thr = nAct + nFin - Q.drain;           // finished lanes' = nFin + (nAct - act) >= drain
            if ((nR != 0u || (cap >= minb && nR < Q.drain)) && 64u - Q.drain > thr) thr = 64u - Q.drain;
            thr = uniq(thr);                                     // (wave-uniform by construction; the loop's exit test stays scalar)
            uint32_t iters = uniq(run_batch ? Q.walk_iters : 0xFFFFFFFFu);
            for (;;) {
                if (STATS) { ++st_iter; st_act += (uint32_t)__popcll(__ballot(cur <= end)); }
                if (cur == end) {
                    // ---- STEP: all spheres of the cell are tested; leave it (spt_grid.h (3)) ----
                    const float m = __builtin_fminf(wtx, __builtin_fminf(wty, wtz));   // grid_walk_exit
                    const float near_t = __uint_as_float(near_key + kQEpsBias);    // 1e20 while nothing is hit
                    uint32_t h = kQFin;
                    if (m < near_t) {                            // else: every cell up to the hit has been visited
                        grid_walk_step(wtx, wty, wtz, wdx, wdy, wdz, wsx, wsy, wsz, wci, m);
                        h = lds_u32(wci);                        // kQFin: the ray has left the table
                    }
                    cur = h & 0xFFFFu; end = h >> 16;
                    if (STATS) st_step += 1;                     // (per lane; reduced at the end)
                }
                if (cur < end) {
                    // ---- TEST: the next sphere of the lane's cell ----
                    const uint32_t ti = lds_u16(cur);
                    cur += 2u;
                    const uint32_t key = sphere_key_q(lds_f4(ti << 4), wo, wd);
                    // a sphere may be listed in several cells and cells are not visited in index order: lowest index among equal keys --
                    // (key, reference) pairs compared as 64-bit numbers, one instruction
                    const bool better = (((unsigned long long)key << 32) | ti) < (((unsigned long long)near_key << 32) | near_i);
                    near_key = better ? key : near_key;
                    near_i = better ? ti : near_i;
                    if (STATS) st_test += 1;
                }
                asm volatile("s_sub_u32 %0, %0, 1" : "+s"(iters) :: "scc");
                if (uniq((uint32_t)__popcll(__ballot(cur <= end))) <= thr || iters == 0u) break;
            }
            QSTAMP(0)
        }

        if (run_batch) {
        float4* const sq = slots + slot * kQSlotF4;
        // per-lane path registers handed from the class code to the begin of the new ray
        f3 o = mk(0, 0, 0), d = mk(0, 0, 1), w = mk(0, 0, 0);
        uint32_t depth = 0, branchf = 0, rbase = 0, k1 = 0, sp = 0;
        bool has_ray = false;
        bool to_gen = false;                                     // the slot's path ended: back onto the GEN list
        uint32_t requeue = 0;                                    // 1 / 2: the redo found a hit of the other class: onto HIT / HITR with the new answer

        if (cls == QC_GEN) {
            // ================= GEN: continue the slot's task (smallpt.cpp:304-340, :252 pop) =================
            sp = __float_as_uint(q1.w) >> 30;
            uint32_t task1 = qt.x, snext = qt.y;                                       // task + 1 (0: the slot has no task)
            uint32_t send = 0;
            if (task1 != 0u) {
                const uint32_t sbeg = ((task1 - 1u) & ((1u << K.nb_log2) - 1u)) * K.sb;
                send = sbeg + K.sb < K.samps ? sbeg + K.sb : K.samps;
            }
            bool gen = false, need_task = false;
            if (valid) {
                if (sp > 0u) {                                   // pending transmitted child (the reflected subtree is done)
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
                if (need_task && task1 != 0u) { const float4 a = sq[5]; K.cells[task1 - 1u] = make_float4(a.x, a.y, a.z, 0.0f); }
                // wave-private chunks of task ids; only the refill touches the global queue word
                const uint32_t cntn = (uint32_t)__popcll(need_mask);
                const uint32_t rk = rank_q(need_mask);
                const uint32_t avail = chunk_end - chunk_next;
                const uint32_t base_old = chunk_next;
                uint32_t base_new = 0;
                if (cntn > avail) {
                    if (!queue_empty) {
                        const int leader = __ffsll((long long)need_mask) - 1;
                        uint32_t nb = 0;
                        if ((int)lane == leader) nb = atomicAdd(K.queue, 64u);
                        base_new = uniq(__shfl(nb, leader));
                        if (base_new >= K.ntasks) { queue_empty = true; }
                    } else {
                        base_new = 0xFFFFFF00u;                  // nothing left: a queue position that stands for no task (deal_task)
                    }
                    chunk_next = base_new + (cntn - avail);
                    chunk_end = base_new + 64u;
                    if (queue_empty) { chunk_next = chunk_end = 0; }
                } else {
                    chunk_next += cntn;
                }
                if (need_task) {
                    const uint32_t nt = deal_task(rk < avail ? base_old + rk : base_new + (rk - avail), K.ntasks);   // (spt_device.h: a pixel's blocks go to different waves)
                    if (nt < K.ntasks) {
                        task1 = nt + 1u; gen = true;
                        snext = (nt & ((1u << K.nb_log2) - 1u)) * K.sb;
                        sq[5] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }                                            // else the slot retires: it is pushed onto no list
                }
            }
            if (gen) {
                // ---- camera ray of sample `snext` of the cell (smallpt.cpp:325-340 / :745-760), as in spt_grid.hip ----
                typedef const __attribute__((address_space(4))) KParams* KArgs;        // K is the first kernel argument
                KArgs kc = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kc));
                const f3 cam_o = mk(kc->cam_o[0], kc->cam_o[1], kc->cam_o[2]);
                const f3 cam_d = mk(kc->cam_d[0], kc->cam_d[1], kc->cam_d[2]);
                const f3 cam_cx = mk(kc->cam_cx[0], kc->cam_cx[1], kc->cam_cx[2]);
                const f3 cam_cy = mk(kc->cam_cy[0], kc->cam_cy[1], kc->cam_cy[2]);
                const uint32_t task = task1 - 1u;
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
                if (kc->sampler == 0u) {
                    const float r1 = 2 * u1;                                  // tent filter :327-330
                    const float q1s = sqrt_rsq(r1 < 1 ? r1 : 2 - r1);
                    const float dx = r1 < 1 ? q1s - 1 : 1 - q1s;
                    const float r2 = 2 * u2;
                    const float q2s = sqrt_rsq(r2 < 1 ? r2 : 2 - r2);
                    const float dy = r2 < 1 ? q2s - 1 : 1 - q2s;
                    // :331-332 in double like the reference; a / w as the exact Markstein sequence (tools/verify_exact_math.c)
                    const double tx = ((double)sx + .5 + (double)dx) / 2.0 + (double)px;
                    const double ty = ((double)sy + .5 + (double)dy) / 2.0 + (double)py;
                    const double qx0 = tx * kc->inv_w, qy0 = ty * kc->inv_h;
                    const double qx = __builtin_fma(__builtin_fma(-qx0, (double)kc->w, tx), kc->inv_w, qx0);
                    const double qy = __builtin_fma(__builtin_fma(-qy0, (double)kc->h, ty), kc->inv_h, qy0);
                    ax = (float)(qx - .5); ay = (float)(qy - .5);
                } else {
                    const float jx = ((float)sx + u1) * 0.5f, jy = ((float)sy + u2) * 0.5f;      // :750
                    const float fx = 0.5f * (2 * jx - 1), fy = 0.5f * (2 * jy - 1);              // :753-758
                    const float nx = (((float)px + 0.5f) + fx) * kc->inv_wf;                     // :628-631
                    const float ny = (((float)py + 0.5f) + fy) * kc->inv_hf;
                    ax = 2.f * nx - 1.f; ay = 2.f * ny - 1.f;                                    // :633
                }
                const f3 dd = cam_cx * ax + cam_cy * ay + cam_d;
                const float inv = rcp_exact(sqrt_exact(dot(dd, dd)));
                o = cam_o + dd * kc->cam_push;                                                  // :333
                d = dd * inv;                                                                   // normalize(d)
                w = mk(1, 1, 1); depth = 0; branchf = 0; rbase = k0;                             // :338-339
                ++snext;
                has_ray = true;
            }
            if (has_ray) {
                sq[2] = make_float4(w.x, w.y, w.z, __uint_as_float(k1));
                sq[3].x = __uint_as_float(rbase);
                reinterpret_cast<uint2*>(sq + 4)[0] = make_uint2(task1, snext);
            }
            QSTAMP(2)
        } else {
            // ================= HIT / HITR: shade the slot's closest hit (smallpt.cpp:168-263 under D2-D6, D18, D19) =================
            const f3 ro = mk(q0.x, q0.y, q0.z), din = mk(q1.x, q1.y, q1.z);
            w = mk(q2.x, q2.y, q2.z); k1 = __float_as_uint(q2.w);
            rbase = __float_as_uint(q3.x);
            const uint32_t pk = __float_as_uint(q1.w);
            depth = pk & 0xFFFu; branchf = (pk >> 12) & 0xFu; sp = pk >> 30;
            uint32_t hkey = q4.x, inst = q4.y;
            // A walk's answer (hit or miss) stands only inside the ray's valid range (spt_grid.h (1): a direction whose length has drifted
            // over a chain of mirror bounces is valid up to t_ok only): otherwise the exhaustive loop of smallpt.cpp:54-70 answers.
            const bool redo = valid && __uint_as_float(hkey + kQEpsBias) > q0.w;
            const unsigned long long mredo = __ballot(redo);
            if (mredo != 0ull) {
                if (STATS) st_redo += (uint32_t)__popcll(mredo);
                if (redo) { hkey = kQInfKey; inst = 0u; }
                if ((uint32_t)__popcll(mredo) <= 32u) {          // ray by ray with the whole wave (wave_closest_sphere)
                    unsigned long long todo = mredo;
                    while (todo != 0ull) {
                        const int rl = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        const f3 wro = mk(__shfl(ro.x, rl), __shfl(ro.y, rl), __shfl(ro.z, rl)), wrd = mk(__shfl(din.x, rl), __shfl(din.y, rl), __shfl(din.z, rl));
                        uint32_t kk, ii;
                        wave_closest_sphere(s_geom, G.n, wro, wrd, lane, kk, ii);
                        if ((int)lane == rl) { hkey = kk; inst = ii; }
                    }
                } else {
                    for (uint32_t i = 0; i < G.n; ++i) {
                        const float4 g = s_geom[i];
                        if (redo) {
                            const uint32_t key = sphere_key_q(g, ro, din);
                            if (key < hkey) { hkey = key; inst = i; }
                        }
                    }
                }
                if (redo && hkey != kQInfKey) {
                    const bool is_r = (__float_as_uint(s_mat[inst].w) & 3u) == 2u;
                    if (is_r != (cls == QC_HITR)) requeue = is_r ? 2u : 1u;
                }
                if (requeue != 0u) {                             // the other class's batch shades it: the answer is final (t_ok = inf)
                    sq[0] = make_float4(q0.x, q0.y, q0.z, __builtin_inff());
                    reinterpret_cast<uint2*>(sq + 3)[1] = make_uint2(hkey, inst);
                }
            }
            const bool live = valid && requeue == 0u;
            if (live) to_gen = true;                             // unless the path goes on (below)
            if (live && hkey != kQInfKey) {                                                 // else :168 miss (D13)
                const float t = __uint_as_float(hkey + kQEpsBias);
                const float4 gh = s_geom[inst];
                const float4 mc = s_mat[inst];                                              // color.xyz, refl | emissive << 2
                const uint32_t rb = __float_as_uint(mc.w);
                const float pmax = __builtin_fmaxf(__builtin_fmaxf(mc.x, mc.y), mc.z);      // :177 (the host table's mat[3 i + 1].w)
                const f3 hx = ro + din * t;                                                 // scene.cpp:137
                const f3 n = normalize<false>(mk(hx.x - gh.x, hx.y - gh.y, hx.z - gh.z));   // scene.cpp:124
                const f3 nl = dot(n, din) < 0 ? n : neg(n);                                 // :174 (D2)
                f3 f = mk(mc.x, mc.y, mc.z);                                                // :175
                if ((rb & 4u) != 0u || (branchf & 8u) != 0u) {                              // :179 (D4); + w*0 is skipped, exact for finite w
                    const float4 a = sq[5], me = K.mat[3 * inst + 0];
                    sq[5] = make_float4(a.x + w.x * me.x, a.y + w.y * me.y, a.z + w.z * me.z, 0.f);
                }
                bool cont = true;
                if (depth > 5u) {                                                           // :188 (D5)
                    if (rng_draw(rbase, k1) < pmax) f = f * rcp_exact<true>(pmax);          // :192 color * (1 / pmax): the host table's third row
                    else cont = false;                                                      // :196
                }
                if (cont) {
                    const f3 off = nl * 0.02f;                                              // :172 (D3)
                    f3 no = hx + off, nd, nf = f;
                    if (cls == QC_HIT) {
                        if ((rb & 3u) == 0u) {                                              // DIFF :208-215
                            const uint32_t u1bits = rng_draw_bits(rbase + kGolden, k1);
                            const float r2 = rng_draw(rbase + 2u * kGolden, k1);
                            const float r2s = sqrt_rsq(r2);
                            float sn, cs;
                            sincos2pi_bits(u1bits, sn, cs);                                  // D17
                            const f3 ww = nl;
                            const bool ay = __builtin_fabsf(ww.x) >= 0.1f;                  // (double)fabs(w.x) > .1, :211
                            const f3 ur = mk(ay ? ww.z : 0.f, ay ? 0.f : -ww.z, ay ? -ww.x : ww.y);
                            const float s2 = ay ? ww.x : ww.y;
                            const float qu = ww.z * ww.z + s2 * s2;                          // dot(ur, ur) with the zero term dropped
                            const f3 uu = ur * rcp_exact<false>(sqrt_rsq<true, true>(qu));
                            const f3 vv = cross(ww, uu);
                            nd = normalize<false>(uu * cs * r2s + vv * sn * r2s + ww * sqrt_rsq<true, true>(1 - r2));   // :212
                        } else {
                            nd = din - n * 2.0f * dot(n, din);                              // SPEC :218-223
                        }
                    } else {
                        nd = din - n * 2.0f * dot(n, din);                                  // :218 reflRay
                        const bool into = dot(n, nl) > 0;                                   // :225
                        const float nnt = into ? 1.0f / 1.5f : 1.5f / 1.0f;                 // :228
                        const float ddn = dot(din, nl);                                     // :229
                        const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn);                // :230
                        if (!(cos2t < 0)) {                                                 // else TIR :232-236
                            const f3 tdir = normalize<true>(din * nnt - n * ((into ? 1.0f : -1.0f) * (ddn * nnt + sqrt_exact(cos2t))));   // :238
                            const float R0 = (0.5f * 0.5f) / (2.5f * 2.5f);                 // :240-242
                            const float cc = 1 - (into ? -ddn : dot(tdir, n));              // :243
                            const float c2 = cc * cc;                                       // :244
                            const float Re = R0 + (1 - R0) * c2 * c2 * cc;                  // :245
                            const float Tr = 1 - Re;                                        // :246
                            const f3 xin = hx - off;                                        // D3
                            if (depth <= 2u) {                                              // :248 split (D6)
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
                                }
                                nf = f * Re;
                            } else {
                                const float Pr = 0.25f + 0.5f * Re;                         // :256
                                const bool pick_refl = rng_draw(rbase + kGolden, k1) < Pr;  // :257
                                const float inv = rcp_exact(pick_refl ? Pr : 1.f - Pr);     // :259 / :263
                                nf = f * (pick_refl ? Re : Tr) * inv;
                                if (!pick_refl) { no = xin; nd = tdir; }
                            }
                        }
                    }
                    // extend() smallpt.cpp:120-123 + D18 + D19
                    w = w * nf;
                    o = no; d = nd;
                    ++depth;
                    rbase += 4u * kGolden;
                    if (depth >= SPT_K_MAX_DEPTH) ++nkill;
                    else if (!(w.x == 0.f && w.y == 0.f && w.z == 0.f)) { has_ray = true; to_gen = false; }
                    if (!(__builtin_fabsf(w.x) < __builtin_inff() && __builtin_fabsf(w.y) < __builtin_inff() && __builtin_fabsf(w.z) < __builtin_inff())) branchf |= 8u;
                }
            }
            if (has_ray) {
                sq[2] = make_float4(w.x, w.y, w.z, __uint_as_float(k1));
                sq[3].x = __uint_as_float(rbase);
            } else if (to_gen) {
                sq[1].w = __uint_as_float(pack_q(depth, branchf, sp));   // the count of the sample's pending transmitted children for the GEN visit
            }
            QSTAMP(3)
        }

        // ================= BEGIN: ray test, always-tested spheres, start of the walk (spt_grid.h (1), (4)) =================
        nbounce += (unsigned long long)__popcll(__ballot(has_ray));
        {
            typedef const __attribute__((address_space(4))) GridParams* GArgs;
            GArgs gp = (GArgs)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(KParams));
            asm volatile("" : "+s"(gp));
            const GridParams& GB = *(const GridParams*)gp;
            bool ok = false;
            float t_ok = __builtin_inff();
            uint32_t bkey = kQInfKey, bi = gb16;                 // (a reference: index + gb16) index 0 with the inf key: never taken for a hit
            // A batch that produced only a few rays (kQFew: the end of a launch; one roulette-immune path -- colour (1,1,1) mirror or glass --
            // bouncing in a closed ball up to the depth cap) does not walk: the whole wave answers each ray with the exhaustive loop.
            const unsigned long long mray = __ballot(has_ray);
            const bool few = (uint32_t)__popcll(mray) <= kQFew;
            if (has_ray) {
                ok = grid_ray_ok(GB, o.x, o.y, o.z, d.x, d.y, d.z, t_ok) && !few;
                if (!ok) t_ok = __builtin_inff();                // the exhaustive loop's answer needs no range
                sq[0] = make_float4(o.x, o.y, o.z, t_ok);
                sq[1] = make_float4(d.x, d.y, d.z, __uint_as_float(pack_q(depth, branchf, sp)));
            }
            if (!few) {
                for (uint32_t k = 0; k < G.nalways; ++k) {       // ascending indices, strict '<' (smallpt.cpp:61)
                    const uint32_t i = s_refs[G.nrefs + k];
                    const float4 g = lds_f4(i << 4);
                    if (has_ray && ok) {
                        const uint32_t key = sphere_key_q(g, o, d);
                        if (key < bkey) { bkey = key; bi = i; }
                    }
                }
            }
            unsigned long long bad = __ballot(has_ray && !ok);
            if (bad != 0ull) {                                   // spt_grid.h (4): the exhaustive loop of smallpt.cpp:54-70, in place
                if (STATS) st_redo += (uint32_t)__popcll(bad);
                if ((uint32_t)__popcll(bad) <= 32u) {
                    // ... ray by ray with the whole wave (wave_closest_sphere)
                    while (bad != 0ull) {
                        const int rl = __ffsll((long long)bad) - 1;
                        bad &= bad - 1ull;
                        const f3 ro = mk(__shfl(o.x, rl), __shfl(o.y, rl), __shfl(o.z, rl)), rd = mk(__shfl(d.x, rl), __shfl(d.y, rl), __shfl(d.z, rl));
                        uint32_t kk, ii;
                        wave_closest_sphere(s_geom, G.n, ro, rd, lane, kk, ii);
                        if ((int)lane == rl) { bkey = kk; bi = ii + gb16; }
                    }
                } else {
                    // ... or every lane its own ray when most of the wave needs it
                    for (uint32_t i = 0; i < G.n; ++i) {
                        const float4 g = s_geom[i];
                        if (has_ray && !ok) {
                            const uint32_t key = sphere_key_q(g, o, d);
                            if (key < bkey) { bkey = key; bi = i + gb16; }
                        }
                    }
                }
                if (has_ray && !ok) {
                    reinterpret_cast<uint2*>(sq + 3)[1] = make_uint2(bkey, bi - gb16);
                    requeue = (bkey != kQInfKey && (__float_as_uint(s_mat[bi - gb16].w) & 3u) == 2u) ? 2u : 1u;
                }
            }
            const bool begun = has_ray && ok;
            const unsigned long long mb = __ballot(begun);
            if (begun) {
                GridWalk gw;
                grid_walk_begin(GB, o.x, o.y, o.z, d.x, d.y, d.z, gw);
                const uint32_t pos = nR + rank_q(mb);
                RD0[pos] = make_float4(o.x, o.y, o.z, __uint_as_float(bkey));
                RD1[pos] = make_float4(d.x, d.y, d.z, __uint_as_float(bi | (slot << 16)));
                const uint32_t cb = cells_off + 4u * gw.ci;
                RD2[pos] = make_float4(gw.tx, gw.ty, gw.tz, __uint_as_float(cb));
                RD3[pos] = make_float4(gw.dtx, gw.dty, gw.dtz, __uint_as_float(lds_u32(cb)));   // (the staged header)
            }
            nR += (uint32_t)__popcll(mb);
        }
        // ================= push the slots that did not begin a walk =================
        {
            const unsigned long long mg = __ballot(to_gen), mh = __ballot(requeue == 1u), mr = __ballot(requeue == 2u);
            if (to_gen) LGN[nG + rank_q(mg)] = (uint8_t)slot;
            if (requeue == 1u) LH[nH + rank_q(mh)] = (uint8_t)slot;
            if (requeue == 2u) LH[S - 1u - nHR - rank_q(mr)] = (uint8_t)slot;
            nG += (uint32_t)__popcll(mg); nH += (uint32_t)__popcll(mh); nHR += (uint32_t)__popcll(mr);
        }
        QSTAMP(cls == QC_GEN ? 2 : 3)
        }

        if (idle) break;
    }
#undef QSTAMP

    // stats: wave reduction then one atomic per wave
    unsigned long long nk = nkill, ns = st_step, nt = st_test;   // (64-bit from here: sums over the wave)
    for (int off = 32; off > 0; off >>= 1) { nk += __shfl_down(nk, off); if (STATS) { ns += __shfl_down(ns, off); nt += __shfl_down(nt, off); } }
    // (the counters' address is read from the kernel-argument segment here: kept in scalar registers across the main loop it costs spill lanes)
    typedef const __attribute__((address_space(4))) KParams* KArgsE;
    KArgsE ke = (KArgsE)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ke));
    unsigned long long* const ctr = ke->counters;
    if (lane == 0) {
        atomicAdd(&ctr[0], nbounce);
        if (nk) atomicAdd(&ctr[1], nk);
        if (timed_out) atomicAdd(&ctr[8], 1ull);
        if (STATS) {
            atomicAdd(&ctr[2], ns); atomicAdd(&ctr[3], nt);
            atomicAdd(&ctr[4], (unsigned long long)st_iter); atomicAdd(&ctr[5], (unsigned long long)st_act);
            atomicAdd(&ctr[6], (unsigned long long)st_redo); atomicAdd(&ctr[7], (unsigned long long)st_exch);
            for (int i = 0; i < 3; ++i) { atomicAdd(&ctr[10 + i], (unsigned long long)st_bat[i]); atomicAdd(&ctr[13 + i], (unsigned long long)st_lan[i]); }
            for (int i = 0; i < 4; ++i) atomicAdd(&ctr[16 + i], (unsigned long long)ph[i]);
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            atomicAdd(&ctr[20], t_end - t_start);
            atomicMax(&ctr[21], t_end - t_start);             // longest wave
        }
    }
}

}  // namespace spt

// LDS of one workgroup: the grid tables + 16 bytes of material per sphere, then per wave R begun walks of 56 bytes and two byte lists of S entries
extern "C" size_t spt_gpool_lds_bytes(const spt::GridParams* G, uint32_t waves, uint32_t S, uint32_t R)
{
    const size_t ngeom = G->n ? G->n : 1u;
    const size_t geom_off = (((size_t)G->nrefs + G->nalways + 1u) * 2u + 15u) & ~(size_t)15u;
    const size_t mat_off = (geom_off + ngeom * 16u + (size_t)G->ncells * 4u + 15u) & ~(size_t)15u;
    return mat_off + ngeom * 16u + (size_t)waves * ((size_t)R * 64u + 2u * (size_t)S);
}
extern "C" size_t spt_gpool_slot_floats(uint32_t blocks, uint32_t waves, uint32_t S) { return (size_t)blocks * waves * S * (spt::kQSlotF4 * 4u); }
extern "C" size_t spt_gpool_stack_floats(uint32_t blocks, uint32_t waves, uint32_t S) { return (size_t)blocks * waves * S * (3u * spt::kQStackF4 * 4u); }

extern "C" hipError_t spt_gpool_launch(const spt::KParams* K, const spt::GridParams* G, const uint32_t* d_cells, const uint16_t* d_refs,
                                       const uint32_t* d_always, const spt::QParams* Q, uint32_t blocks, uint32_t threads, int stats, hipStream_t stream)
{
    if (threads == 0 || threads > (uint32_t)spt::kQBlock || (threads & 63u)) return hipErrorInvalidValue;
    if (Q->S == 0 || Q->S > 256u || (Q->S & 15u) || (Q->R & 3u) || Q->R == 0 || Q->R > 0xFFFFu || Q->drain == 0 || Q->drain > 64u || G->nrefs >= 0x7FFEu) return hipErrorInvalidValue;
    if ((((size_t)G->nrefs + G->nalways + 1u) * 2u + 15u) / 16u + G->n > 0xFFFFu) return hipErrorInvalidValue;   // references (index + gb16) are 16 bits
    const size_t lds = spt_gpool_lds_bytes(G, threads / 64u, Q->S, Q->R);
    if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
    const void* fn = stats ? reinterpret_cast<const void*>(&spt::gpoolkernel<true>) : reinterpret_cast<const void*>(&spt::gpoolkernel<false>);
    hipFuncAttributes fa{};
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return e;
    if (fa.sharedSizeBytes != 0) return hipErrorInvalidValue;   // the kernel's LDS addressing assumes its dynamic LDS starts at 0
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (stats) hipLaunchKernelGGL(spt::gpoolkernel<true>, dim3(blocks), dim3(threads), lds, stream, *K, *G, d_cells, d_refs, d_always, *Q);
    else hipLaunchKernelGGL(spt::gpoolkernel<false>, dim3(blocks), dim3(threads), lds, stream, *K, *G, d_cells, d_refs, d_always, *Q);
    return hipGetLastError();
}
