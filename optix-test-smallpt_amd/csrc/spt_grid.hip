// spt_grid.hip -- persistent path-tracing kernel for LARGE sphere tables on gfx950 (MI355X): the closest hit of
// intersectGlobalSpheres (smallpt.cpp:54-70 over scene.cpp:129-140) goes through a uniform grid that is exhaustive-equivalent by
// construction (spt_grid.h), everything else -- shadePaths (smallpt.cpp:154-267), camera rays, RNG (D7), accumulation order (D9),
// sin/cos (D17), depth cap (D18), zero-weight cut (D19) -- is the arithmetic of spt_kernel.hip / the oracle, bit for bit.
//
//   * ONE workgroup of 1024 threads per CU shares one LDS copy of the whole structure: sphere records {c, r*r} (16 B), cell headers
//     (4 B: first reference << 13 | count, a one-cell border of sentinels), 16-bit sphere references, the always-tested list.
//     Config 5 (1024 spheres): 16 + 25 + 8 KB of the CU's 160 KB; every lookup of the walk is an LDS read, none goes to memory.
//   * a lane owns a path (registers) and is in one of four states: NONE (needs a camera ray / a pending glass child / a task),
//     FRESH (has a new ray), WALK (inside the grid), HIT (closest hit known, waits for shading).  The wave runs the phases
//     regenerate -> begin walks -> walk -> shade in a loop; inside the walk phase a lane either TESTS the next sphere of its cell
//     or STEPS to the next cell, and each iteration runs whichever of the two more lanes want.  The walk phase is left as soon as
//     fewer lanes are still walking than wait for shading / regeneration: the stragglers keep their walk state and continue in
//     the next round together with the rays the others have produced meanwhile, so neither a long ray nor a crowded cell holds
//     the other 63 lanes for long.
//   * rays the grid may not take (origin too far from the box for the error bound, direction whose squared length has drifted from
//     1: spt_grid.h (1)) run the exhaustive loop in place; spheres more than 16 x the median radius (the walls and the light of a
//     Cornell box) are tested for every ray before the walk, which also bounds it.
#include "spt_device.h"
#define SPT_GRID_DEVICE_ONLY
#include "spt_grid.h"
#include "spt_kernel.h"

namespace spt {

constexpr int kGridBlock = 1024;
constexpr uint32_t kGEpsBias = 0x38D1B717u + 1u;                 // bits(1e-4f) + 1
constexpr uint32_t kGInfKey = 0x60AD78ECu - kGEpsBias;           // key of 1e20f (maths.h:16)
constexpr int kGChunk = 64;                                      // task ids fetched from the global queue per atomic

enum : uint32_t { M_NONE = 0, M_FRESH = 1, M_WALK = 2, M_HIT = 3 };

__device__ __forceinline__ uint32_t lane_id_g() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// intersectAnalytic of one sphere record {c, r*r} on integer keys (scene.cpp:129-140, smallpt.cpp:59-65): key(t) = bits(t) - (bits(eps) + 1),
// "t > eps && t < nearest" is one unsigned compare; det < 0 gives NaN roots whose keys lie above every valid one.
__device__ __forceinline__ uint32_t sphere_key_g(const float4 g, f3 o, f3 d)
{
    const f3 op = mk(g.x - o.x, g.y - o.y, g.z - o.z);                                  // :132
    const float bb = dot(op, d);                                                        // :133
    const float det = bb * bb - dot(op, op) + g.w;                                      // :133 (g.w = r*r)
    const float sd = sqrt_rsq(det);                                                     // :134
    const uint32_t key1 = __float_as_uint(bb - sd) - kGEpsBias;                         // :135
    const uint32_t key2 = __float_as_uint(bb + sd) - kGEpsBias;
    return key1 < key2 ? key1 : key2;
}

struct GPath { f3 o, d, w; uint32_t depth, branch, rbase; };

// STATS: walk statistics and per-phase wave time (s_memtime) for tools/bench_grid.py; the product build carries none of it
// GLOBAL_TABLES (round 4): sphere records, cell headers and references are read where they lie in global memory (L2 / Infinity-Cache
// resident: 16 384 spheres are 0.26 MB of records + 0.5 MB of grid) instead of one CU's LDS -- tables beyond the LDS keep the grid (the same
// walk, the same proof, spt_grid.h) instead of falling to the one-lane-per-path hierarchy.  The always-tested list is read from its own array.
// WHERE = 2: the sphere records stay in global memory, the cell headers and references -- two of the walk's three lookups per sphere --
// in LDS: tables whose records alone exceed the LDS but whose grid fits it.
template <bool STATS, int WHERE>
__global__ __launch_bounds__(kGridBlock) void gridkernel(const KParams K, const GridParams G, const uint32_t* __restrict__ g_cells,
                                                         const uint16_t* __restrict__ g_refs, const uint32_t* __restrict__ g_always, uint32_t leave_q)
{
    constexpr bool GLOBAL_TABLES = WHERE == 1, GLOBAL_GEOM = WHERE != 0;
    extern __shared__ float4 s_lds_geom[];                       // n sphere records (WHERE 0), then the grid tables (WHERE 0, 2)
    uint32_t* const s_lds_cells = reinterpret_cast<uint32_t*>(s_lds_geom + (GLOBAL_GEOM ? 0u : (G.n ? G.n : 1u)));
    uint16_t* const s_lds_refs = reinterpret_cast<uint16_t*>(s_lds_cells + G.ncells);   // nrefs cell references, then the always-tested list, one spare
    if (!GLOBAL_GEOM) for (uint32_t i = threadIdx.x; i < G.n; i += blockDim.x) s_lds_geom[i] = K.geom[i];
    if (!GLOBAL_TABLES) {
        for (uint32_t i = threadIdx.x; i < G.ncells; i += blockDim.x) s_lds_cells[i] = g_cells[i];
        for (uint32_t i = threadIdx.x; i < G.nrefs; i += blockDim.x) s_lds_refs[i] = g_refs[i];
        for (uint32_t i = threadIdx.x; i <= G.nalways; i += blockDim.x) s_lds_refs[G.nrefs + i] = i < G.nalways ? (uint16_t)g_always[i] : (uint16_t)0;
    }
    // (the template argument decides the address space at compile time: LDS reads or global loads, never generic ones)
    auto geom_at = [&](uint32_t i) -> float4 { return GLOBAL_GEOM ? K.geom[i] : s_lds_geom[i]; };
    auto cell_at = [&](uint32_t ci) -> uint32_t { return GLOBAL_TABLES ? g_cells[ci] : s_lds_cells[ci]; };
    auto ref_at = [&](uint32_t k) -> uint32_t { return GLOBAL_TABLES ? (uint32_t)g_refs[k] : (uint32_t)s_lds_refs[k]; };
    auto always_at = [&](uint32_t k) -> uint32_t { return GLOBAL_TABLES ? g_always[k] : (uint32_t)s_lds_refs[G.nrefs + k]; };
    __syncthreads();

    const uint32_t lane = lane_id_g();
    const uint32_t gthread = blockIdx.x * blockDim.x + threadIdx.x;
    float* const gstack = K.stack + (size_t)gthread * (3 * 12);  // 3 pending transmitted children x 12 words per thread
    // per-lane state
    uint32_t mode = M_NONE;
    bool task_valid = false, queue_empty = false;
    uint32_t task = 0, sp = 0, s_gen = 0, s_end = 0, px = 0, py = 0, cell = 0, p0 = 0, p1 = 0, k0 = 0, k1 = 0;
    GPath p{mk(0, 0, 0), mk(0, 0, 1), mk(0, 0, 0), 0u, 0u, 0u};
    f3 acc = mk(0, 0, 0);
    float wtx = 0.f, wty = 0.f, wtz = 0.f, wdx = 0.f, wdy = 0.f, wdz = 0.f;   // the lane's walk (GridWalk, spt_grid.h) in separate registers
    int32_t wsx = 0, wsy = 0, wsz = 0;
    uint32_t wci = 0;
    uint32_t cur = 0, end = 0;                                   // references of the current cell still to test; end = 0 outside the walk
    uint32_t near_key = kGInfKey, near_i = 0xFFFFFFFFu;
    float t_ok = 0.f;                                            // the lane's walk is valid up to this ray parameter (spt_grid.h (1))
    bool redo = false;                                           // the walk ended beyond it: phase B runs the exhaustive loop for this ray
    uint32_t nbounce = 0, nkill = 0;
    uint32_t chunk_next = 0, chunk_end = 0;                      // wave-uniform: this wave's private range of task ids
    // statistics (wave-uniform counters, lane 0 reports)
    unsigned long long n_steps = 0, n_tests = 0, n_step_iters = 0, n_test_iters = 0, n_fallback = 0, n_shade_lanes = 0;
    unsigned long long ph[5] = {0, 0, 0, 0, 0};                  // wave time in regeneration, walk begin, test bodies, step bodies, shading
    unsigned long long ph_t = 0;
#define GSTAMP(i) if (STATS) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - ph_t; ph_t = t_; }
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    bool timed_out = false;
    uint32_t n_rounds = 0;

    for (;;) {
        ++n_rounds;
        if ((n_rounds & 63u) == 0u && K.watchdog_ticks != 0ull && __builtin_amdgcn_s_memtime() - t_start > K.watchdog_ticks) { timed_out = true; break; }
        if (STATS) ph_t = __builtin_amdgcn_s_memtime();
        // ================= A: regeneration (smallpt.cpp:304-340, :252 pop) =================
        if (mode == M_NONE && sp > 0) {                          // pending transmitted child of the lane's current sample
            --sp;
            const float* e = gstack + sp * 12;
            p.o = mk(e[0], e[1], e[2]); p.d = mk(e[3], e[4], e[5]); p.w = mk(e[6], e[7], e[8]);
            const uint32_t db = __float_as_uint(e[9]);
            p.depth = db & 0xFFFFu; p.branch = db >> 16;
            p.rbase = rng_base(k0, p.branch, p.depth);
            mode = M_FRESH;
        }
        {
            // task completion + wave-aggregated fetch: a wave-private chunk of kGChunk task ids per atomic on the queue word
            const bool need_task = mode == M_NONE && s_gen == s_end && !queue_empty;
            const unsigned long long need_mask = __ballot(need_task);
            if (need_mask != 0ull) {
                if (need_task && task_valid) K.cells[task] = make_float4(acc.x, acc.y, acc.z, 0.0f);
                const uint32_t cnt = (uint32_t)__popcll(need_mask);
                const uint32_t rank = (uint32_t)__popcll(need_mask & ((1ull << lane) - 1ull));
                const uint32_t avail = chunk_end - chunk_next;
                uint32_t base_old = chunk_next, base_new = 0;
                if (cnt > avail) {
                    const int leader = __ffsll((long long)need_mask) - 1;
                    uint32_t nb = 0;
                    if ((int)lane == leader) nb = atomicAdd(K.queue, (uint32_t)kGChunk);
                    base_new = __builtin_amdgcn_readfirstlane(__shfl(nb, leader));
                    chunk_next = base_new + (cnt - avail);
                    chunk_end = base_new + (uint32_t)kGChunk;
                } else {
                    chunk_next += cnt;
                }
                const uint32_t base = rank < avail ? base_old : base_new - avail;
                if (need_task) {
                    task = deal_task(base + rank, K.ntasks);      // (spt_device.h: a pixel's blocks go to different waves)
                    task_valid = task < K.ntasks;
                    if (task_valid) {
                        // task = ((pixel * 4 + cell) << nb_log2) | block: one block of a jitter cell's samples (D9)
                        const uint32_t cellid = task >> K.nb_log2, blk = task & ((1u << K.nb_log2) - 1u);
                        const uint32_t pix_local = cellid >> 2;
                        cell = cellid & 3u;
                        const uint32_t ry = pix_local / K.w;
                        px = pix_local - ry * K.w;
                        py = K.row_begin + (ry >> K.rb_log2) * K.rb_stride + (ry & K.rb_mask);
                        const uint32_t pixel_idx = py * K.w + px;                      // GLOBAL index (smallpt.cpp:298)
                        p0 = mix32(pixel_idx + K.s0); p1 = mix32(pixel_idx ^ K.s1);
                        s_gen = blk * K.sb;
                        s_end = s_gen + K.sb < K.samps ? s_gen + K.sb : K.samps;
                        acc = mk(0, 0, 0);
                    } else {
                        queue_empty = true; s_gen = s_end = 0;
                    }
                }
            }
        }
        if (mode == M_NONE && task_valid && s_gen < s_end) {
            // camera ray of sample s_gen (smallpt.cpp:325-340 / :745-760), as in spt_kernel.hip phase C1.  The camera constants are
            // read from the kernel-argument segment HERE (the empty asm keeps the compiler from hoisting ~30 scalar loads out of the
            // main loop, where they would push the walk's scalars into spill lanes).
            typedef const __attribute__((address_space(4))) KParams* KArgs;        // K is the first kernel argument
            KArgs kc = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kc));
            const f3 cam_o = mk(kc->cam_o[0], kc->cam_o[1], kc->cam_o[2]);
            const f3 cam_d = mk(kc->cam_d[0], kc->cam_d[1], kc->cam_d[2]);
            const f3 cam_cx = mk(kc->cam_cx[0], kc->cam_cx[1], kc->cam_cx[2]);
            const f3 cam_cy = mk(kc->cam_cy[0], kc->cam_cy[1], kc->cam_cy[2]);
            const uint32_t index_in_pixel = cell * K.samps + s_gen;                    // :306
            k0 = mix32(p0 ^ (index_in_pixel * kGolden));
            k1 = mix32(p1 + index_in_pixel * 0x85EBCA6Bu);
            const float u1 = rng_draw(k0 + ((1u << 28) | 0u) * kGolden, k1);
            const float u2 = rng_draw(k0 + ((1u << 28) | 1u) * kGolden, k1);
            const uint32_t sx = cell & 1u, sy = cell >> 1;
            float ax, ay;
            if (kc->sampler == 0u) {
                const float r1 = 2 * u1;                                               // tent filter :327-330
                const float q1 = sqrt_rsq(r1 < 1 ? r1 : 2 - r1);
                const float dx = r1 < 1 ? q1 - 1 : 1 - q1;
                const float r2 = 2 * u2;
                const float q2 = sqrt_rsq(r2 < 1 ? r2 : 2 - r2);
                const float dy = r2 < 1 ? q2 - 1 : 1 - q2;
                // :331-332 in double like the reference; a / w as the exact Markstein sequence (tools/verify_exact_math.c)
                const double tx = ((double)sx + .5 + (double)dx) / 2.0 + (double)px;
                const double ty = ((double)sy + .5 + (double)dy) / 2.0 + (double)py;
                const double qx0 = tx * kc->inv_w, qy0 = ty * kc->inv_h;
                const double qx = __builtin_fma(__builtin_fma(-qx0, (double)kc->w, tx), kc->inv_w, qx0);
                const double qy = __builtin_fma(__builtin_fma(-qy0, (double)kc->h, ty), kc->inv_h, qy0);
                ax = (float)(qx - .5); ay = (float)(qy - .5);
            } else {
                const float jx = ((float)sx + u1) * 0.5f, jy = ((float)sy + u2) * 0.5f;  // :750
                const float fx = 0.5f * (2 * jx - 1), fy = 0.5f * (2 * jy - 1);          // :753-758
                const float nx = (((float)px + 0.5f) + fx) * kc->inv_wf;                   // :628-631
                const float ny = (((float)py + 0.5f) + fy) * kc->inv_hf;
                ax = 2.f * nx - 1.f; ay = 2.f * ny - 1.f;                                // :633
            }
            const f3 dd = cam_cx * ax + cam_cy * ay + cam_d;
            const float inv = rcp_exact(sqrt_exact(dot(dd, dd)));
            p.o = cam_o + dd * kc->cam_push;                                           // :333
            p.d = dd * inv;                                                            // normalize(d)
            p.w = mk(1, 1, 1); p.depth = 0; p.branch = 0; p.rbase = k0;                // :338-339
            ++s_gen;
            mode = M_FRESH;
        }
        if (__ballot(mode != M_NONE) == 0ull) break;             // no lane has a path, a pending child, a sample or a task left
        GSTAMP(0)

        // ================= B: new rays: ray test, always-tested spheres, start of the walk =================
        {
            const bool fresh = mode == M_FRESH;
            if (__ballot(fresh) != 0ull) {
                // the ~25 grid constants are read from the kernel-argument segment here (G is the argument after K), like the camera
                // constants of phase A: kept in scalar registers across the whole loop they push the walk's scalars into spill lanes
                typedef const __attribute__((address_space(4))) GridParams* GArgs;
                GArgs gp = (GArgs)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(KParams));
                asm volatile("" : "+s"(gp));
                const GridParams& GB = *(const GridParams*)gp;
                bool ok = false;
                // A wave with only a few new rays and nobody walking (the end of a launch; a roulette-immune path -- colour (1,1,1) mirror or
                // glass -- bouncing in a closed ball up to the depth cap) answers each with the exhaustive loop run by ALL its lanes: lane l
                // tests spheres l, l + 64, ..., then the lexicographic minimum of (key, index) over the wave (below).
                const unsigned long long mfresh = __ballot(fresh);
                const bool few = (uint32_t)__popcll(mfresh) <= 4u && __ballot(mode == M_WALK) == 0ull && G.n >= 64u;
                if (fresh) {
                    if (!redo) ++nbounce;                        // (a ray handed back by its walk was counted when it started)
                    near_key = kGInfKey; near_i = 0u;            // index 0 with the inf key: never replaced by another inf key, never taken for a hit
                    ok = grid_ray_ok(GB, p.o.x, p.o.y, p.o.z, p.d.x, p.d.y, p.d.z, t_ok) && !redo && !few;
                    if (!ok) t_ok = __builtin_inff();            // the exhaustive loop's answer needs no range
                    redo = false;
                }
                if (few) {
                    unsigned long long todo = mfresh;
                    if (STATS) n_fallback += (unsigned long long)__popcll(todo);
                    while (todo != 0ull) {
                        const int rl = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        const f3 ro = mk(__shfl(p.o.x, rl), __shfl(p.o.y, rl), __shfl(p.o.z, rl)), rd = mk(__shfl(p.d.x, rl), __shfl(p.d.y, rl), __shfl(p.d.z, rl));
                        uint32_t wk = kGInfKey, wi = 0u;
                        for (uint32_t i = lane; i < G.n; i += 64u) {
                            const uint32_t key = sphere_key_g(geom_at(i), ro, rd);
                            if (key < wk) { wk = key; wi = i; }
                        }
#pragma unroll 1
                        for (int off = 32; off > 0; off >>= 1) {
                            const uint32_t k2 = (uint32_t)__shfl_xor((int)wk, off), i2 = (uint32_t)__shfl_xor((int)wi, off);
                            const bool better = (k2 < wk) | ((k2 == wk) & (i2 < wi));
                            wk = better ? k2 : wk; wi = better ? i2 : wi;
                        }
                        if ((int)lane == rl) { near_key = wk; near_i = wk == kGInfKey ? 0u : wi; }
                    }
                }
                if (!few)
                for (uint32_t k = 0; k < G.nalways; ++k) {       // the walls and the light of a Cornell box: ascending indices, strict '<' (smallpt.cpp:61)
                    const uint32_t i = always_at(k);
                    const float4 g = geom_at(i);
                    if (fresh && ok) {
                        const uint32_t key = sphere_key_g(g, p.o, p.d);
                        if (key < near_key) { near_key = key; near_i = i; }
                    }
                }
                const unsigned long long bad = few ? 0ull : __ballot(fresh && !ok);
                if (bad != 0ull) {                               // spt_grid.h (4): the exhaustive loop of smallpt.cpp:54-70
                    if (STATS) n_fallback += (unsigned long long)__popcll(bad);
                    for (uint32_t i = 0; i < G.n; ++i) {
                        const float4 g = geom_at(i);
                        if (fresh && !ok) {
                            const uint32_t key = sphere_key_g(g, p.o, p.d);
                            if (key < near_key) { near_key = key; near_i = i; }
                        }
                    }
                }
                if (fresh) {
                    if (ok) {
                        GridWalk w;
                        grid_walk_begin(GB, p.o.x, p.o.y, p.o.z, p.d.x, p.d.y, p.d.z, w);
                        wtx = w.tx; wty = w.ty; wtz = w.tz; wdx = w.dtx; wdy = w.dty; wdz = w.dtz; wsx = w.sx; wsy = w.sy; wsz = w.sz; wci = w.ci;
                        const uint32_t h = cell_at(wci);          // the start cell is clamped into the table: never a border cell
                        cur = h >> kGridCountBits; end = cur + (h & ((1u << kGridCountBits) - 1u));
                        mode = M_WALK;
                    } else {
                        mode = M_HIT;
                    }
                }
            }
        }

        GSTAMP(1)
        // ================= C: walk =================
        {
            uint32_t nwalk = (uint32_t)__popcll(__ballot(mode == M_WALK));
            uint32_t nidle = (uint32_t)__popcll(__ballot(mode == M_HIT));   // lanes whose hit waits for shading
            while (nwalk != 0u && nwalk * 16u >= nidle * leave_q) {      // leave_q = 0: every walk runs to its end
                const bool wt = cur < end;                       // end = 0 for lanes outside the walk
                const uint32_t nt = (uint32_t)__popcll(__ballot(wt));
                if (2u * nt >= nwalk) {
                    // ---- TEST: the next sphere of the lane's cell ----
                    if (STATS) { ++n_test_iters; n_tests += nt; }
                    if (wt) {
                        const uint32_t i = ref_at(cur);
                        ++cur;
                        const uint32_t key = sphere_key_g(geom_at(i), p.o, p.d);
                        // a sphere may be listed in several cells and cells are not visited in index order: lowest index among equal keys
                        const bool better = (key < near_key) | ((key == near_key) & (i < near_i));
                        near_key = better ? key : near_key;
                        near_i = better ? i : near_i;
                    }
                    GSTAMP(2)
                } else {
                    // ---- STEP: leave the cell (all its spheres are tested) ----
                    if (STATS) { ++n_step_iters; n_steps += nwalk - nt; }
                    if (mode == M_WALK && !wt) {
                        const float m = __builtin_fminf(wtx, __builtin_fminf(wty, wtz));   // grid_walk_exit
                        const float near_t = __uint_as_float(near_key + kGEpsBias);    // 1e20 while nothing is hit
                        bool stop = !(m < near_t);               // spt_grid.h (3): every cell up to the hit has been visited
                        if (!stop) {
                            grid_walk_step(wtx, wty, wtz, wdx, wdy, wdz, wsx, wsy, wsz, wci, m);
                            const uint32_t h = cell_at(wci);
                            stop = h == kGridBorder;             // left the table
                            cur = h >> kGridCountBits; end = cur + (h & ((1u << kGridCountBits) - 1u));
                        }
                        if (stop) { mode = M_HIT; end = 0; }      // end = 0: "cur < end" is false outside the walk
                    }
                    const uint32_t still = (uint32_t)__popcll(__ballot(mode == M_WALK));
                    nidle += nwalk - still;
                    nwalk = still;
                    GSTAMP(3)
                }
            }
        }

        // ================= D: shadePaths for the lanes whose closest hit is known (smallpt.cpp:168-263 under D2-D6, D18, D19) =================
        // A walk's answer (hit or miss) stands only inside the ray's valid range (spt_grid.h (1): a direction whose length has drifted
        // over a chain of mirror bounces is valid up to t_ok only); otherwise the exhaustive loop takes the ray over in the next round.
        if (mode == M_HIT && __uint_as_float(near_key + kGEpsBias) > t_ok) { mode = M_FRESH; redo = true; }
        if (mode == M_HIT) {
            if (STATS) ++n_shade_lanes;
            mode = M_NONE;
            if (near_key != kGInfKey) {                                                // else :168 miss (D13)
                const uint32_t inst = near_i;
                const float t = __uint_as_float(near_key + kGEpsBias);
                const float4 gh = geom_at(inst);
                const float4 me = K.mat[3 * inst + 0], mc = K.mat[3 * inst + 1];
                const int refl = __float_as_int(me.w) & 3;
                const f3 hx = p.o + p.d * t;                                           // scene.cpp:137
                const f3 n = normalize<false>(mk(hx.x - gh.x, hx.y - gh.y, hx.z - gh.z));   // scene.cpp:124
                const f3 nl = dot(n, p.d) < 0 ? n : neg(n);                            // :174 (D2)
                f3 f = mk(mc.x, mc.y, mc.z);                                           // :175
                acc = acc + p.w * mk(me.x, me.y, me.z);                                // :179 (D4)
                bool cont = true;
                if (p.depth > 5) {                                                     // :188 (D5)
                    if (rng_draw(p.rbase, k1) < mc.w) { const float4 mf = K.mat[3 * inst + 2]; f = mk(mf.x, mf.y, mf.z); }
                    else cont = false;
                }
                if (cont) {
                    const f3 off = nl * 0.02f;                                         // :172 (D3)
                    f3 no = hx + off, nd, nf = f;
                    if (refl == 0) {                                                   // DIFF :208-215
                        const uint32_t u1bits = rng_draw_bits(p.rbase + kGolden, k1);
                        const float r2 = rng_draw(p.rbase + 2u * kGolden, k1);
                        const float r2s = sqrt_rsq(r2);
                        float sn, cs;
                        sincos2pi_bits(u1bits, sn, cs);                                // D17
                        const f3 ww = nl;
                        const bool ay = __builtin_fabsf(ww.x) >= 0.1f;                // (double)fabs(w.x) > .1, :211
                        const f3 ur = mk(ay ? ww.z : 0.f, ay ? 0.f : -ww.z, ay ? -ww.x : ww.y);
                        const float s2 = ay ? ww.x : ww.y;
                        const float qu = ww.z * ww.z + s2 * s2;                        // dot(ur, ur) with the zero term dropped
                        const f3 uu = ur * rcp_exact<false>(sqrt_rsq<true, true>(qu));
                        const f3 vv = cross(ww, uu);
                        nd = normalize<false>(uu * cs * r2s + vv * sn * r2s + ww * sqrt_rsq<true, true>(1 - r2));   // :212
                    } else {
                        nd = p.d - n * 2.0f * dot(n, p.d);                             // :218 reflRay
                        if (refl == 2) {                                               // REFR :225-263
                            const bool into = dot(n, nl) > 0;
                            const float nnt = into ? 1.0f / 1.5f : 1.5f / 1.0f;
                            const float ddn = dot(p.d, nl);
                            const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
                            if (!(cos2t < 0)) {                                        // else TIR :232-236
                                const f3 tdir = normalize<true>(p.d * nnt - n * ((into ? 1.0f : -1.0f) * (ddn * nnt + sqrt_exact(cos2t))));   // :238
                                const float R0 = (0.5f * 0.5f) / (2.5f * 2.5f);
                                const float cc = 1 - (into ? -ddn : dot(tdir, n));
                                const float c2 = cc * cc;
                                const float Re = R0 + (1 - R0) * c2 * c2 * cc;
                                const float Tr = 1 - Re;
                                const f3 xin = hx - off;
                                if (p.depth <= 2) {                                    // :248 split (D6)
                                    const f3 tw = p.w * (f * Tr);
                                    if (!(tw.x == 0.f && tw.y == 0.f && tw.z == 0.f)) {
                                        float* e = gstack + sp * 12;
                                        e[0] = xin.x; e[1] = xin.y; e[2] = xin.z; e[3] = tdir.x; e[4] = tdir.y; e[5] = tdir.z;
                                        e[6] = tw.x; e[7] = tw.y; e[8] = tw.z;
                                        e[9] = __uint_as_float((p.depth + 1u) | ((p.branch | (1u << p.depth)) << 16));
                                        ++sp;
                                    }
                                    nf = f * Re;
                                } else {
                                    const float Pr = 0.25f + 0.5f * Re;
                                    const bool pick_refl = rng_draw(p.rbase + kGolden, k1) < Pr;
                                    const float inv = rcp_exact(pick_refl ? Pr : 1.f - Pr);
                                    nf = f * (pick_refl ? Re : Tr) * inv;
                                    if (!pick_refl) { no = xin; nd = tdir; }
                                }
                            }
                        }
                    }
                    // extend() smallpt.cpp:120-123 + D18 + D19
                    p.w = p.w * nf;
                    p.o = no; p.d = nd;
                    ++p.depth;
                    p.rbase += 4u * kGolden;
                    if (p.depth >= SPT_K_MAX_DEPTH) ++nkill;
                    else if (!(p.w.x == 0.f && p.w.y == 0.f && p.w.z == 0.f)) mode = M_FRESH;
                }
            }
        }
        GSTAMP(4)
    }
#undef GSTAMP

    // stats: wave reduction then one atomic per wave
    unsigned long long nb = nbounce, nk = nkill, ns = n_shade_lanes;
    for (int off = 32; off > 0; off >>= 1) { nb += __shfl_down(nb, off); nk += __shfl_down(nk, off); if (STATS) ns += __shfl_down(ns, off); }
    if (lane == 0) {
        atomicAdd(&K.counters[0], nb);
        if (nk) atomicAdd(&K.counters[1], nk);
        if (timed_out) atomicAdd(&K.counters[8], 1ull);
        if (STATS) {
            atomicAdd(&K.counters[2], n_steps); atomicAdd(&K.counters[3], n_tests);
            atomicAdd(&K.counters[4], n_step_iters); atomicAdd(&K.counters[5], n_test_iters);
            atomicAdd(&K.counters[6], n_fallback); atomicAdd(&K.counters[7], (unsigned long long)n_rounds);
            atomicAdd(&K.counters[9], ns);
            for (int i = 0; i < 5; ++i) atomicAdd(&K.counters[10 + i], ph[i]);
            atomicAdd(&K.counters[15], __builtin_amdgcn_s_memtime() - t_start);
        }
    }
}

}  // namespace spt

extern "C" size_t spt_grid_lds_bytes(const spt::GridParams* G)
{
    return (size_t)(G->n ? G->n : 1u) * 16u + (size_t)G->ncells * 4u + (((size_t)G->nrefs + G->nalways + 2u) / 2u) * 4u;
}
extern "C" int spt_grid_block_threads(void) { return spt::kGridBlock; }
extern "C" size_t spt_grid_stack_floats(uint32_t blocks, uint32_t threads) { return (size_t)blocks * threads * 36u; }

// LDS of the grid tables alone (WHERE = 2: the sphere records stay in global memory)
extern "C" size_t spt_grid_lds_bytes_tables(const spt::GridParams* G)
{
    return (size_t)G->ncells * 4u + (((size_t)G->nrefs + G->nalways + 2u) / 2u) * 4u + 16u;
}

template <bool STATS, int WHERE>
static hipError_t launch_grid(const spt::KParams* K, const spt::GridParams* G, const uint32_t* d_cells, const uint16_t* d_refs, const uint32_t* d_always,
                              uint32_t blocks, uint32_t threads, uint32_t leave_q, size_t lds, hipStream_t stream)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spt::gridkernel<STATS, WHERE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((spt::gridkernel<STATS, WHERE>), dim3(blocks), dim3(threads), lds, stream, *K, *G, d_cells, d_refs, d_always, leave_q);
    return hipGetLastError();
}

// where: 0 = every table in LDS, 1 = every table in global memory, 2 = sphere records in global memory, cell headers and references in LDS
extern "C" hipError_t spt_grid_launch(const spt::KParams* K, const spt::GridParams* G, const uint32_t* d_cells, const uint16_t* d_refs,
                                      const uint32_t* d_always, uint32_t blocks, uint32_t threads, uint32_t leave_q, int stats, int where, hipStream_t stream)
{
    if (threads == 0 || threads > (uint32_t)spt::kGridBlock || (threads & 63u) || where < 0 || where > 2) return hipErrorInvalidValue;
    const size_t lds = where == 1 ? 0 : (where == 2 ? spt_grid_lds_bytes_tables(G) : spt_grid_lds_bytes(G));
    if (where == 0) return stats ? launch_grid<true, 0>(K, G, d_cells, d_refs, d_always, blocks, threads, leave_q, lds, stream) : launch_grid<false, 0>(K, G, d_cells, d_refs, d_always, blocks, threads, leave_q, lds, stream);
    if (where == 1) return stats ? launch_grid<true, 1>(K, G, d_cells, d_refs, d_always, blocks, threads, leave_q, lds, stream) : launch_grid<false, 1>(K, G, d_cells, d_refs, d_always, blocks, threads, leave_q, lds, stream);
    return stats ? launch_grid<true, 2>(K, G, d_cells, d_refs, d_always, blocks, threads, leave_q, lds, stream) : launch_grid<false, 2>(K, G, d_cells, d_refs, d_always, blocks, threads, leave_q, lds, stream);
}
