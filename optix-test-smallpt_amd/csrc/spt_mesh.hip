// spt_mesh.hip -- triangle-mesh primitives of the reference on gfx950 (MI355X): the Intersector seam
// (addTriangleMesh / build / traceRays, smallpt.cpp:427-473; the OptiX Prime variant :475-603 is what it replaces) and
// the path tracer over a mesh scene.
//
//   * triIntersect (scene.cpp:52-70), intersect(ro, rd, mesh) (scene.cpp:95-116), CPUIntersector::intersect
//     (smallpt.cpp:443-458) and makeHit(instId, mesh, meshHit) (scene.cpp:73-93) with exactly their arithmetic: the
//     per-triangle constants v0, v1-v0, v2-v0 and n = cross(v1-v0, v2-v0) are evaluated once on the host (the same single
//     IEEE operations the reference repeats per call); d = 1.0 / dot(rd, n) -- a double division rounded to float in the
//     reference -- equals the correctly rounded binary32 reciprocal for every input (a double-rounding slip would need
//     x * m = 1 with a 25-bit m, i.e. x a power of two, where both are exact), so it is rcp_exact(|x|) with the sign restored.
//   * closest hit = brute force over all triangles of all instances in (instance, triangle) order with strict '<' and
//     dist > 0, which is what the per-mesh loop + the per-instance loop of the reference select.  Triangle records
//     (48 B) are staged through LDS in tiles by the whole workgroup and read back wave-uniformly (broadcast).
//   * meshkernel: one lane = one path, one task = one D9 sample block of a jitter cell, same RNG / accumulation order /
//     shading decisions (D1-D19) as the sphere kernels; hit.n is the interpolated, un-normalised vertex normal exactly
//     as makeHit returns it.  This is the reference's actual direction of travel (OptiX triangles); the exhaustive loop is
//     the default and the correctness anchor.
//   * SPT_ACCEL_BVH (the default of mesh scenes since round 4; spt_bvh.h, spt_tribvh.h): the same query through structures that provably reach
//     every triangle whose report beats or ties the answer -- same triIntersect arithmetic on the triangles they reach, same selection rule
//     (smallest t > 0, lowest global index among equal t) --: the stand-in for the OptiX Prime traversal of smallpt.cpp:475-603, and the
//     exhaustive loop's Hit for every ray.  SPT_ACCEL_BVH_FAST: the spatial hierarchy alone.
#include "spt_device.h"
#include "spt_kernel.h"

namespace spt {

constexpr int kMeshBlock = 256;
constexpr int kTile = 768;                                       // triangles per LDS tile: 768 x 48 B = 36 KB
constexpr uint32_t kMeshInfKey = 0x60AD78ECu - 1u;               // key of 1e20f (maths.h:16); key(t) = bits(t) - 1

__device__ __forceinline__ uint32_t lane_id_m() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// 1.0 / x of scene.cpp:62 (see the header): correctly rounded reciprocal with the sign handled outside the fast path
__device__ __forceinline__ float rcp_signed(float x)
{
    const float r = rcp_exact<true>(__builtin_fabsf(x));
    return __uint_as_float(__float_as_uint(r) | (__float_as_uint(x) & 0x80000000u));
}

// triIntersect against one staged triangle record; returns t (1e20 when the barycentrics reject it) and u, v.
// Branch-free: the four rejection tests of scene.cpp:67 are evaluated together (same truth value as the short-circuit form).
__device__ __forceinline__ float tri_test(const float4 r0, const float4 r1, const float4 r2, f3 ro, f3 rd, float& u, float& v)
{
    const f3 v0 = mk(r0.x, r0.y, r0.z), e1 = mk(r1.x, r1.y, r1.z), e2 = mk(r2.x, r2.y, r2.z), n = mk(r0.w, r1.w, r2.w);
    const f3 rov0 = ro - v0;                                                     // :58
    const f3 q = cross(rov0, rd);                                                // :61
    const float d = rcp_signed(dot(rd, n));                                      // :62
    u = d * dot(neg(q), e2);                                                     // :63
    v = d * dot(q, e1);                                                          // :64
    const float t = d * dot(neg(n), rov0);                                       // :65
    const bool rej = (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | ((u + v) > 1.0f);    // :67
    return rej ? 1e20f : t;
}

// Closest hit over every triangle (see the header).  All threads of the workgroup must call this together (the tiles are
// staged cooperatively); `active` lanes trace their ray.  Returns the global triangle index or 0xFFFFFFFF.
__device__ __forceinline__ uint32_t closest_triangle(const float4* __restrict__ tris, uint32_t ntris, float4* s_tile,
                                                     bool active, f3 ro, f3 rd, float& t_out)
{
    uint32_t near_key = kMeshInfKey, near_tri = 0xFFFFFFFFu;
    // "t > 0 && t < nearest" (scene.cpp:105, smallpt.cpp:449) as one unsigned compare on key = bits(t) - 1: +0 wraps to
    // the top, negative and NaN keys lie above the key of 1e20; ascending order + strict '<' = lowest index wins ties
    auto consider = [&](const float4 r0, const float4 r1, const float4 r2, uint32_t index) {
        float u, v;
        const float t = tri_test(r0, r1, r2, ro, rd, u, v);
        const uint32_t key = __float_as_uint(t) - 1u;
        const bool better = key < near_key;
        near_key = better ? key : near_key;
        near_tri = better ? index : near_tri;
    };
    for (uint32_t base = 0; base < ntris; base += kTile) {
        const uint32_t cnt = ntris - base < (uint32_t)kTile ? ntris - base : (uint32_t)kTile;
        __syncthreads();                                       // the previous tile is no longer read
        for (uint32_t i = threadIdx.x; i < 3u * cnt; i += blockDim.x) s_tile[i] = tris[3u * (size_t)base + i];
        __syncthreads();
        if (active) {
            uint32_t k = 0;
            for (; k + 4 <= cnt; k += 4) {                     // four records (12 broadcast reads) in flight per LDS round trip
                const float4* r = s_tile + 3 * k;
                const float4 a0 = r[0], a1 = r[1], a2 = r[2], b0 = r[3], b1 = r[4], b2 = r[5];
                const float4 c0 = r[6], c1 = r[7], c2 = r[8], d0 = r[9], d1 = r[10], d2 = r[11];
                consider(a0, a1, a2, base + k);
                consider(b0, b1, b2, base + k + 1);
                consider(c0, c1, c2, base + k + 2);
                consider(d0, d1, d2, base + k + 3);
            }
            for (; k < cnt; ++k) consider(s_tile[3 * k], s_tile[3 * k + 1], s_tile[3 * k + 2], base + k);
        }
    }
    t_out = __uint_as_float(near_key + 1u);
    return near_tri;
}

// The same exhaustive query for a FEW rays of the workgroup (at most kFewRays): instead of 256 lanes each walking all triangles
// for their own ray -- a full pass of the tile loop even when one path of the workgroup is still alive, which is what the end of
// a low-spp frame looks like -- the rays are listed in LDS and, ray by ray, the 256 threads share the triangles (thread t tests
// triangles t, t + 256, ...), keep their best (key, global index) and the workgroup takes the minimum: smallest key, lowest index
// among equal keys = what the reference's ascending loops with strict '<' select.  Same tri_test, so the same answer, in
// ntris / 256 tests per thread and ray.  Up to 128 rays this beats the tile loop (measured: the shipped scene at 4 spp 13.4 -> 3.1 ms,
// at 256 spp 136 -> 128 ms; with 256 it loses, 229 ms).  All threads of the workgroup call it together; nact = number of active lanes (> 0).
constexpr int kFewRays = 128;
__device__ __forceinline__ uint32_t closest_triangle_few(const float4* __restrict__ tris, uint32_t ntris, float4* s_tile, uint32_t nact,
                                                         bool active, f3 ro, f3 rd, float& t_out)
{
    uint32_t* const s_u = reinterpret_cast<uint32_t*>(s_tile);   // [0] counter, [4 .. 4 + 8 kFewRays) rays, then per-wave partial minima, then results
    float* const s_f = reinterpret_cast<float*>(s_tile);
    unsigned long long* const s_part = reinterpret_cast<unsigned long long*>(s_u + 4 + 8 * kFewRays);   // kMeshBlock / 64 entries
    unsigned long long* const s_res = s_part + kMeshBlock / 64;                                            // kFewRays entries
    __syncthreads();                                             // the tile area is no longer read by a previous query
    if (threadIdx.x == 0) s_u[0] = 0u;
    __syncthreads();
    uint32_t slot = 0;
    if (active) {
        slot = atomicAdd(&s_u[0], 1u);
        float* r = s_f + 4 + 8 * slot;
        r[0] = ro.x; r[1] = ro.y; r[2] = ro.z; r[3] = rd.x; r[4] = rd.y; r[5] = rd.z;
    }
    __syncthreads();
    for (uint32_t k = 0; k < nact; ++k) {
        const float* r = s_f + 4 + 8 * k;
        const f3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
        uint32_t best_key = kMeshInfKey, best_tri = 0xFFFFFFFFu;
        for (uint32_t i = threadIdx.x; i < ntris; i += kMeshBlock) {
            float u, v;
            const float t = tri_test(tris[3 * (size_t)i], tris[3 * (size_t)i + 1], tris[3 * (size_t)i + 2], o, d, u, v);
            const uint32_t key = __float_as_uint(t) - 1u;
            if (key < best_key) { best_key = key; best_tri = i; }     // ascending i per thread: strict '<' keeps the lowest index
        }
        unsigned long long m = ((unsigned long long)best_key << 32) | best_tri;   // lexicographic (key, index)
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long x = __shfl_down(m, off); m = x < m ? x : m; }
        if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long a = s_part[0];
            for (int w = 1; w < kMeshBlock / 64; ++w) a = s_part[w] < a ? s_part[w] : a;
            s_res[k] = a;
        }
        __syncthreads();
    }
    uint32_t near_key = kMeshInfKey, near_tri = 0xFFFFFFFFu;
    if (active) { const unsigned long long a = s_res[slot]; near_key = (uint32_t)(a >> 32); near_tri = (uint32_t)a; }
    if (near_key >= kMeshInfKey) { near_key = kMeshInfKey; near_tri = 0xFFFFFFFFu; }
    t_out = __uint_as_float(near_key + 1u);
    return near_tri;
}

// The same query through the hierarchy (spt_bvh.h).  Per lane: a stack of <= 32 child references in LDS (entry e of thread
// t at s_stack[e * blockDim + t]: conflict-free), near child first, both children's padded boxes tested against the ray
// segment [0, current nearest] with widened slabs (a box is only skipped when the ray misses it by more than the widening;
// NaN from 0 * inf drops out of v_min/v_max, which is the conservative side).  A triangle replaces the current hit when its
// key is smaller, or equal with a lower global index: the (instance, triangle)-ascending strict '<' of the reference's loops.
// Since round 4 the query is exhaustive-equivalent for every ray (spt_tribvh.h): the spatial walk inflates the child boxes per ray and
// finds every report whose error is bounded; the rays for which triIntersect's determinant is zero to rounding -- in a regular
// triangle's plane, or near the supporting line of a thin triangle's long edge -- find those triangles through a cone tree over the
// triangles' planes and a table (or tree) of the long edges' lines.  The walks and node tests are the host / device functions of spt_tribvh.h, which the CPU harness
// (tests/sanitize/tribvh_main.cpp) runs against the exhaustive loop.
constexpr uint32_t kCoopTriangleRays = 49152;                   // meshkernel<1>: a wave with few live rays answers them with the exhaustive loop
constexpr uint32_t kMeshArgOffset = (uint32_t)((sizeof(KParams) + alignof(MParams) - 1) / alignof(MParams) * alignof(MParams));   // meshkernel(KParams, MParams)
struct LdsStack {
    uint32_t* base;                                            // entry e of thread t at base[e * kMeshBlock + t]: conflict-free
    __device__ __forceinline__ void push(uint32_t sp, uint32_t v) { base[sp * kMeshBlock + threadIdx.x] = v; }
    __device__ __forceinline__ uint32_t pop(uint32_t sp) const { return base[sp * kMeshBlock + threadIdx.x]; }
};

// KOFF = byte offset of the MParams argument in the kernel-argument segment: what only the plane / line structures need is read from
// there where it is used (scalar loads) instead of sitting in SGPRs through the whole bounce loop (cf. the camera constants of meshkernel).
template <uint32_t KOFF>
__device__ __forceinline__ uint32_t closest_triangle_bvh(const MParams& M, uint32_t* s_stack, bool active, bool camera_ray, f3 ro, f3 rd, float& t_out)
{
    uint32_t near_key = kMeshInfKey, near_tri = 0xFFFFFFFFu;
    if (active) {
        float tcut = 1e20f;                                        // widened distance of the current nearest hit
        LdsStack st{s_stack};
        auto consider = [&](const float4* r, uint32_t g) {
            float u, v;
            const float t = tri_test(r[0], r[1], r[2], ro, rd, u, v);
            const uint32_t key = __float_as_uint(t) - 1u;
            if (key < near_key || (key == near_key && g < near_tri)) {          // key == kMeshInfKey never replaces: near_tri would have to be larger
                if (key < kMeshInfKey) { near_key = key; near_tri = g; tcut = t * 1.0001f; }
            }
        };
        TriQuery q;
        tri_query(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, q);
        const float ivx = __builtin_amdgcn_rcpf(rd.x), ivy = __builtin_amdgcn_rcpf(rd.y), ivz = __builtin_amdgcn_rcpf(rd.z);   // 1 ulp; inside the widening
        auto leaf = [&](uint32_t first, uint32_t cnt) {
            for (uint32_t k = 0; k < cnt; ++k) consider(M.bvh_tris + 3 * (size_t)(first + k), M.bvh_index[first + k]);
        };
        auto by_index = [&](uint32_t g) { consider(M.tris + 3 * (size_t)g, g); };
        if (!M.bvh_cones) {                                        // SPT_ACCEL_BVH_FAST: the plain hierarchy (documented exceptions)
            tri_walk_boxes<false>(M.bvh_nodes, nullptr, ro.x, ro.y, ro.z, ivx, ivy, ivz, q.h[0], q.h[1], q.h[2], tcut, st, leaf);
        } else {
            tri_walk_boxes<true>(M.bvh_nodes, M.bvh_cones, ro.x, ro.y, ro.z, ivx, ivy, ivz, q.h[0], q.h[1], q.h[2], tcut, st, leaf);
            typedef const __attribute__((address_space(4))) MParams* MArgs;
            MArgs mc = (MArgs)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + KOFF);
            asm volatile("" : "+s"(mc));
            if (camera_ray) {                                      // the planes through the camera point are listed (spt_bvh.h camera_planes)
                const uint32_t ncam = mc->ncam;
                const uint32_t* cam_planes = mc->cam_planes;
                for (uint32_t k = 0; k < ncam; ++k) by_index(cam_planes[k]);
            } else if (mc->plane_nodes) {
                tri_walk_planes(mc->plane_nodes, q, st, by_index);
            }
            if (mc->flat_lines) tri_scan_lines(mc->flat_lines, mc->flat_line_index, mc->nline_slots, q, st, by_index);
            else if (mc->line_nodes) tri_walk_lines(mc->line_nodes, q, st, by_index);
        }
    }
    t_out = __uint_as_float(near_key + 1u);
    return near_tri;
}

// ---- sphere tables through a hierarchy (spt_set_sphere_accel; spt_bvh.h build_sphere_bvh) -----------------------------------
// intersectAnalytic of one sphere record {c, r*r} on the integer keys of the sphere kernels (scene.cpp:129-140, smallpt.cpp:59-65):
// key(t) = bits(t) - (bits(eps) + 1); returns the smaller of the two root keys (NaN roots give keys above every valid one).
constexpr uint32_t kSphEpsBias = 0x38D1B717u + 1u;               // bits(1e-4f) + 1
constexpr uint32_t kSphInfKey = 0x60AD78ECu - kSphEpsBias;       // key of 1e20f
__device__ __forceinline__ uint32_t sphere_key(const float4 g, f3 o, f3 d)
{
    const f3 op = mk(g.x - o.x, g.y - o.y, g.z - o.z);                                  // :132
    const float bb = dot(op, d);                                                        // :133
    const float det = bb * bb - dot(op, op) + g.w;                                      // :133 (g.w = r*r)
    const float sd = sqrt_exact(det);                                                   // :134 (NaN for det < 0: both keys lose)
    const uint32_t key1 = __float_as_uint(bb - sd) - kSphEpsBias;                       // :135
    const uint32_t key2 = __float_as_uint(bb + sd) - kSphEpsBias;
    return key1 < key2 ? key1 : key2;
}

// Closest sphere by traversal.  EXHAUSTIVE-EQUIVALENT BY CONSTRUCTION, unlike the triangle hierarchy: intersectAnalytic divides
// by nothing, so its rounding error is bounded.  With u = 2^-24, e = c - o, |d|^2 = 1 + eta and the reported t of a sphere, the
// point p = o + t d satisfies | |p - c|^2 - r*r | <= 101 u (|e|^2 + r*r) + |eta| t^2 (error budget of the
// ten operations; the eta term because the formula assumes a unit direction).  A child box holds c +- r of its spheres; with D =
// the distance from o to the box's farthest corner (>= sqrt(|e|^2 + r*r) for every sphere inside, and t <= 2.1 D) the box
// inflated by pad = (2^-7 + 2.1 sqrt(|eta| + 8u)) D contains p with 2^-8 D to spare: 2^-8 D covers the rounding budget
// (sqrt of 2^-16 D^2, a 2.5x reserve on 101 u), 2.1 sqrt(|eta|) D the direction's length, the spare half absorbs the rounding
// of this slab test (~3u D) and of the measured eta (8u).  Hence a sphere whose reported key beats the final answer is never
// skipped: its box is entered at a parameter <= its t <= the current nearest.  NaN from 0 * inf drops out of v_min/v_max (no
// constraint).  (One pad for the whole traversal from the root's extent was measured: fatter boxes, 4-12 % slower.)
__device__ __forceinline__ uint32_t closest_sphere_bvh(const KParams& K, const MParams& M, const float4* nodes, const float4* leaf_geom, const uint32_t* leaf_index,
                                                    uint32_t* s_stack, bool active, f3 ro, f3 rd, float& t_out)
{
    uint32_t near_key = kSphInfKey, near_i = 0xFFFFFFFFu;
    if (active) {
        float tcut = 1e20f;
        auto consider = [&](const float4 g, uint32_t index) {
            const uint32_t key = sphere_key(g, ro, rd);
            if (key < near_key || (key == near_key && index < near_i)) {       // ascending index + strict '<' of smallpt.cpp:61
                if (key < kSphInfKey) { near_key = key; near_i = index; tcut = __uint_as_float(key + kSphEpsBias) * 1.0001f; }
            }
        };
        for (uint32_t k = 0; k < M.nalways; ++k) { const uint32_t i = M.always[k]; consider(K.geom[i], i); }
        const f3 iv = mk(__builtin_amdgcn_rcpf(rd.x), __builtin_amdgcn_rcpf(rd.y), __builtin_amdgcn_rcpf(rd.z));
        // inflation per unit of D for THIS ray: 2^-7 for the rounding budget, plus 2.1 sqrt(|eta|) for a direction whose squared
        // length is 1 + eta (intersectAnalytic assumes 1; mirror reflections are not renormalised and drift over a long chain):
        // the reported point then misses the sphere by |eta| t^2, t <= 2.1 D
        const float kpad = (1.0f / 128.0f + 2.1f * __builtin_amdgcn_sqrtf(__builtin_fabsf(dot(rd, rd) - 1.0f) + 0x1p-21f)) * 1.001f;
        uint32_t sp = 0;
        int cur = 0;
        for (;;) {
            if (cur >= 0) {
                const float4* nd = nodes + 4 * (size_t)cur;
                const float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
                // child boxes relative to the origin; farthest-corner distance; inflation
                const float l0x = a.x - ro.x, l1x = a.w - ro.x, l0y = a.y - ro.y, l1y = b.x - ro.y, l0z = a.z - ro.z, l1z = b.y - ro.z;
                const float r0x = b.z - ro.x, r1x = c.y - ro.x, r0y = b.w - ro.y, r1y = c.z - ro.y, r0z = c.x - ro.z, r1z = c.w - ro.z;
                const float lmx = __builtin_fmaxf(__builtin_fabsf(l0x), __builtin_fabsf(l1x)), lmy = __builtin_fmaxf(__builtin_fabsf(l0y), __builtin_fabsf(l1y)),
                            lmz = __builtin_fmaxf(__builtin_fabsf(l0z), __builtin_fabsf(l1z));
                const float rmx = __builtin_fmaxf(__builtin_fabsf(r0x), __builtin_fabsf(r1x)), rmy = __builtin_fmaxf(__builtin_fabsf(r0y), __builtin_fabsf(r1y)),
                            rmz = __builtin_fmaxf(__builtin_fabsf(r0z), __builtin_fabsf(r1z));
                // kpad D with D >= the Euclidean distance: the (1 ulp) v_sqrt_f32 of the sum of squares (the 1.001 sits in kpad)
                const float lp = __builtin_amdgcn_sqrtf(lmx * lmx + lmy * lmy + lmz * lmz) * kpad;
                const float rp = __builtin_amdgcn_sqrtf(rmx * rmx + rmy * rmy + rmz * rmz) * kpad;
                const float lx0 = (l0x - lp) * iv.x, lx1 = (l1x + lp) * iv.x, ly0 = (l0y - lp) * iv.y, ly1 = (l1y + lp) * iv.y, lz0 = (l0z - lp) * iv.z, lz1 = (l1z + lp) * iv.z;
                const float rx0 = (r0x - rp) * iv.x, rx1 = (r1x + rp) * iv.x, ry0 = (r0y - rp) * iv.y, ry1 = (r1y + rp) * iv.y, rz0 = (r0z - rp) * iv.z, rz1 = (r1z + rp) * iv.z;
                const float ln = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(lx0, lx1), __builtin_fminf(ly0, ly1)), __builtin_fminf(lz0, lz1));
                const float lf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(lx0, lx1), __builtin_fmaxf(ly0, ly1)), __builtin_fmaxf(lz0, lz1));
                const float rn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(rx0, rx1), __builtin_fminf(ry0, ry1)), __builtin_fminf(rz0, rz1));
                const float rf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(rx0, rx1), __builtin_fmaxf(ry0, ry1)), __builtin_fmaxf(rz0, rz1));
                // an empty child (inverted box: mn = +inf, mx = -inf) gives tn = +inf, tf = -inf (or NaN): never entered
                const bool hl = (lf >= 0.f) & (ln <= lf) & (ln <= tcut);
                const bool hr = (rf >= 0.f) & (rn <= rf) & (rn <= tcut);
                const int lref = __float_as_int(d.x), rref = __float_as_int(d.y);
                if (hl & hr) {
                    const bool left_first = ln <= rn;
                    s_stack[sp * kMeshBlock + threadIdx.x] = (uint32_t)(left_first ? rref : lref);
                    ++sp;
                    cur = left_first ? lref : rref;
                    continue;
                }
                if (hl | hr) { cur = hl ? lref : rref; continue; }
            } else {
                const uint32_t code = (uint32_t)~cur, first = code >> 4, cnt = code & 15u;
                for (uint32_t k = 0; k < cnt; ++k) consider(leaf_geom[first + k], leaf_index[first + k]);
            }
            if (sp == 0u) break;
            --sp;
            cur = (int)s_stack[sp * kMeshBlock + threadIdx.x];
        }
    }
    t_out = __uint_as_float(near_key + kSphEpsBias);
    return near_i;
}

struct MeshHit { float dist; uint32_t inst, tri; f3 x, n; float u, v; };

// Sphere::makeHit(SphereHit) (scene.cpp:118-127): x = o + d t (scene.cpp:137), n = normalize(x - center)
__device__ __forceinline__ MeshHit make_sphere_hit(const KParams& K, uint32_t i, float t, f3 ro, f3 rd)
{
    MeshHit h;
    const float4 g = K.geom[i];
    h.x = ro + rd * t;
    h.n = normalize<true>(mk(h.x.x - g.x, h.x.y - g.y, h.x.z - g.z));
    h.dist = t; h.inst = i; h.tri = 0u; h.u = 0.f; h.v = 0.f;
    return h;
}

// makeHit(instId, mesh, meshHit), scene.cpp:73-93, for the winning triangle (u, v re-evaluated from its record)
__device__ __forceinline__ MeshHit make_hit(const MParams& M, uint32_t tri, float t, f3 ro, f3 rd)
{
    MeshHit h;
    float u, v;
    (void)tri_test(M.tris[3 * (size_t)tri], M.tris[3 * (size_t)tri + 1], M.tris[3 * (size_t)tri + 2], ro, rd, u, v);
    const uint4 ix = M.tri_index[tri];                                           // {i1, i2, i3 (global vertex ids), instance}
    const float w = 1.f - u - v;                                                 // :82
    const float4 p1 = M.verts[2 * (size_t)ix.x], p2 = M.verts[2 * (size_t)ix.y], p3 = M.verts[2 * (size_t)ix.z];
    const float4 n1 = M.verts[2 * (size_t)ix.x + 1], n2 = M.verts[2 * (size_t)ix.y + 1], n3 = M.verts[2 * (size_t)ix.z + 1];
    h.x = mk(p1.x, p1.y, p1.z) * w + mk(p2.x, p2.y, p2.z) * u + mk(p3.x, p3.y, p3.z) * v;   // :88
    h.n = mk(n1.x, n1.y, n1.z) * w + mk(n2.x, n2.y, n2.z) * u + mk(n3.x, n3.y, n3.z) * v;   // :89
    h.dist = t; h.inst = ix.w; h.tri = tri - M.inst_first_tri[ix.w]; h.u = u; h.v = v;
    return h;
}

// Intersector::traceRays (smallpt.cpp:460-470 / :553-587): one Hit (scene.h:31-43, 44 bytes) per ray.
template <bool BVH>
__global__ __launch_bounds__(kMeshBlock) void trace_rays(const MParams M, const float* __restrict__ rays, uint64_t nrays, float* __restrict__ hits)
{
    extern __shared__ float4 s_tile[];
    const uint64_t i = (uint64_t)blockIdx.x * kMeshBlock + threadIdx.x;
    const bool active = i < nrays;
    f3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    if (active) { ro = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]); rd = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]); }
    float t;
    const uint32_t tri = BVH ? closest_triangle_bvh<0u>(M, reinterpret_cast<uint32_t*>(s_tile), active, false, ro, rd, t)
                             : closest_triangle(M.tris, M.ntris, s_tile, active, ro, rd, t);
    if (!active) return;
    float* h = hits + 11 * i;
    if (tri == 0xFFFFFFFFu) {                                                    // Hit{}: dist = inf (smallpt.cpp:454-455)
        h[0] = 1e20f;
        for (int k = 1; k < 11; ++k) h[k] = 0.f;
        return;
    }
    const MeshHit m = make_hit(M, tri, t, ro, rd);
    h[0] = m.dist; h[1] = __uint_as_float(m.inst); h[2] = __uint_as_float(m.tri);
    h[3] = m.x.x; h[4] = m.x.y; h[5] = m.x.z; h[6] = m.n.x; h[7] = m.n.y; h[8] = m.n.z; h[9] = m.u; h[10] = m.v;
}

// ---- path tracer over the mesh scene ------------------------------------------------------------------------------
struct MPath { f3 o, d, w; uint32_t depth, branch, rbase; };

// GEOM 0: triangles, exhaustive; 1: triangles through the hierarchy; 2: a sphere table through its hierarchy (staging a small
// tree into LDS behind the stacks was measured: 30 % slower -- two workgroups per CU and conflicting per-lane ds_read_b128)
template <int GEOM>
__global__ __launch_bounds__(kMeshBlock) void meshkernel(const KParams K, const MParams M)
{
    extern __shared__ float4 s_tile[];
    const uint32_t lane = lane_id_m();
    const uint32_t gthread = blockIdx.x * kMeshBlock + threadIdx.x;
    float* const gstack = K.stack + (size_t)gthread * (3 * 12);                  // 3 pending children x 12 words per thread

    bool alive = false, task_valid = false, queue_empty = false;
    uint32_t task = 0, sp = 0, s_gen = 0, s_end = 0, px = 0, py = 0, cell = 0, p0 = 0, p1 = 0, k0 = 0, k1 = 0;
    MPath p{mk(0, 0, 0), mk(0, 0, 1), mk(0, 0, 0), 0u, 0u, 0u};
    f3 acc = mk(0, 0, 0);
    unsigned long long nbounce = 0, nkill = 0;

    for (;;) {
        // ---- continue the lane's task: pending transmitted child, next sample of the block, or a new task ----
        if (!alive && sp > 0) {
            --sp;
            const float* e = gstack + sp * 12;
            p.o = mk(e[0], e[1], e[2]); p.d = mk(e[3], e[4], e[5]); p.w = mk(e[6], e[7], e[8]);
            const uint32_t db = __float_as_uint(e[9]);
            p.depth = db & 0xFFFFu; p.branch = db >> 16;
            p.rbase = rng_base(k0, p.branch, p.depth);
            alive = true;
        }
        if (!alive && s_gen == s_end && !queue_empty) {
            if (task_valid) K.cells[task] = make_float4(acc.x, acc.y, acc.z, 0.0f);
            uint32_t qpos = atomicAdd(K.queue, 1u);           // the triangle loop dwarfs this atomic
            if (GEOM == 1 && M.strips) {
                // through the triangle hierarchy the lanes of a wave hold an 8 x 8 TILE of pixels (spt_deal.h deal_task_tiles; scenes without
                // mirror / glass materials, MParams::strips); a tile's part beyond the image is a hole inside the range: fetch again
                const uint32_t S = 4u << K.nb_log2, rows = K.ntasks / (S * K.w), qend = deal_tiles_end(K.w, rows, S);
                task = deal_task_tiles(qpos, K.w, rows, S);
                while (task == 0xFFFFFFFFu && qpos < qend) { qpos = atomicAdd(K.queue, 1u); task = deal_task_tiles(qpos, K.w, rows, S); }
            } else {
                task = deal_task(qpos, K.ntasks);             // (spt_deal.h: a pixel's blocks go to different waves)
            }
            task_valid = task < K.ntasks;
            if (task_valid) {
                const uint32_t cellid = task >> K.nb_log2, blk = task & ((1u << K.nb_log2) - 1u);
                const uint32_t pix_local = cellid >> 2;
                cell = cellid & 3u;
                const uint32_t ry = pix_local / K.w;
                px = pix_local - ry * K.w; py = K.row_begin + (ry >> K.rb_log2) * K.rb_stride + (ry & K.rb_mask);
                const uint32_t pixel_idx = py * K.w + px;                        // GLOBAL index (smallpt.cpp:298)
                p0 = mix32(pixel_idx + K.s0); p1 = mix32(pixel_idx ^ K.s1);
                s_gen = blk * K.sb;
                s_end = s_gen + K.sb < K.samps ? s_gen + K.sb : K.samps;
                acc = mk(0, 0, 0);
            } else {
                queue_empty = true; s_gen = s_end = 0;
            }
        }
        if (!alive && task_valid && s_gen < s_end) {
            // camera ray of sample s_gen (smallpt.cpp:325-340 / :745-760), as in spt_kernel.hip phase C1.  The camera constants are read
            // from the kernel-argument segment here (the empty asm keeps ~30 scalar loads from being hoisted out of the bounce loop,
            // where they would push its scalars into spill lanes; K is the first kernel argument)
            typedef const __attribute__((address_space(4))) KParams* KArgs;
            KArgs kc = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kc));
            const f3 cam_o = mk(kc->cam_o[0], kc->cam_o[1], kc->cam_o[2]);
            const f3 cam_d = mk(kc->cam_d[0], kc->cam_d[1], kc->cam_d[2]);
            const f3 cam_cx = mk(kc->cam_cx[0], kc->cam_cx[1], kc->cam_cx[2]);
            const f3 cam_cy = mk(kc->cam_cy[0], kc->cam_cy[1], kc->cam_cy[2]);
            const uint32_t index_in_pixel = cell * K.samps + s_gen;              // :306
            k0 = mix32(p0 ^ (index_in_pixel * kGolden));
            k1 = mix32(p1 + index_in_pixel * 0x85EBCA6Bu);
            const float u1 = rng_draw(k0 + ((1u << 28) | 0u) * kGolden, k1);
            const float u2 = rng_draw(k0 + ((1u << 28) | 1u) * kGolden, k1);
            const uint32_t sx = cell & 1u, sy = cell >> 1;
            float ax, ay;
            if (kc->sampler == 0u) {
                const float r1 = 2 * u1;
                const float q1 = sqrt_rsq(r1 < 1 ? r1 : 2 - r1);
                const float dx = r1 < 1 ? q1 - 1 : 1 - q1;
                const float r2 = 2 * u2;
                const float q2 = sqrt_rsq(r2 < 1 ? r2 : 2 - r2);
                const float dy = r2 < 1 ? q2 - 1 : 1 - q2;
                const double tx = ((double)sx + .5 + (double)dx) / 2.0 + (double)px;
                const double ty = ((double)sy + .5 + (double)dy) / 2.0 + (double)py;
                const double qx0 = tx * kc->inv_w, qy0 = ty * kc->inv_h;
                const double qx = __builtin_fma(__builtin_fma(-qx0, (double)kc->w, tx), kc->inv_w, qx0);
                const double qy = __builtin_fma(__builtin_fma(-qy0, (double)kc->h, ty), kc->inv_h, qy0);
                ax = (float)(qx - .5); ay = (float)(qy - .5);
            } else {
                const float jx = ((float)sx + u1) * 0.5f, jy = ((float)sy + u2) * 0.5f;
                const float fx = 0.5f * (2 * jx - 1), fy = 0.5f * (2 * jy - 1);
                const float nx = (((float)px + 0.5f) + fx) * kc->inv_wf;
                const float ny = (((float)py + 0.5f) + fy) * kc->inv_hf;
                ax = 2.f * nx - 1.f; ay = 2.f * ny - 1.f;
            }
            const f3 dd = cam_cx * ax + cam_cy * ay + cam_d;
            const float inv = rcp_exact(sqrt_exact(dot(dd, dd)));
            p.o = cam_o + dd * kc->cam_push;
            p.d = dd * inv;
            p.w = mk(1, 1, 1); p.depth = 0; p.branch = 0; p.rbase = k0;
            ++s_gen;
            alive = true;
        }
        // exhaustive: the workgroup stages the triangle tiles together, so it leaves together; the traversals need no barrier and
        // a wave must not wait for the slowest ray of its three neighbours every bounce
        uint32_t nalive = 0;                                     // exhaustive mode: paths alive in the whole workgroup
        if (GEOM == 0) { nalive = (uint32_t)__syncthreads_count(alive ? 1 : 0); if (nalive == 0u) break; }
        else if (__ballot(alive) == 0ull) break;

        // ---- sphere table, a wave with only a few rays left (the end of a launch; a roulette-immune path -- colour (1,1,1) mirror or glass --
        // bouncing in a closed ball up to the depth cap): the whole wave answers each ray with the exhaustive loop of smallpt.cpp:54-70
        // -- lane l tests spheres l, l + 64, ... (coalesced records, ascending, strict '<'), then the lexicographic minimum of
        // (key, index) over the wave = the loop's key and the lowest index that gives it (:61) -- instead of one lane chasing
        // dependent node loads through the hierarchy.  Up to 16 384 spheres (256 tests per lane). ----
        float t;
        uint32_t coop_tri = 0xFFFFFFFFu;
        bool coop = false;
        if (GEOM == 2) {
            unsigned long long todo = __ballot(alive);
            coop = todo != 0ull && (uint32_t)__popcll(todo) <= 4u && K.n >= 64u && K.n <= 16384u;
            if (coop) {
                while (todo != 0ull) {
                    const int rl = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const f3 ro = mk(__shfl(p.o.x, rl), __shfl(p.o.y, rl), __shfl(p.o.z, rl)), rd = mk(__shfl(p.d.x, rl), __shfl(p.d.y, rl), __shfl(p.d.z, rl));
                    uint32_t wk = kSphInfKey, wi = 0xFFFFFFFFu;
                    for (uint32_t i = lane; i < K.n; i += 64u) {
                        const uint32_t key = sphere_key(K.geom[i], ro, rd);
                        if (key < wk) { wk = key; wi = i; }
                    }
#pragma unroll 1
                    for (int off = 32; off > 0; off >>= 1) {
                        const uint32_t k2 = (uint32_t)__shfl_xor((int)wk, off), i2 = (uint32_t)__shfl_xor((int)wi, off);
                        const bool better = (k2 < wk) | ((k2 == wk) & (i2 < wi));
                        wk = better ? k2 : wk; wi = better ? i2 : wi;
                    }
                    if ((int)lane == rl) { coop_tri = wk < kSphInfKey ? wi : 0xFFFFFFFFu; t = __uint_as_float(wk + kSphEpsBias); }
                }
            }
        }
        // ---- triangle hierarchy, the same situation (a lone mirror / glass chain keeps one lane of a wave busy for thousands of bounces, each a
        // chain of dependent node loads: the Cornell-like mesh scene of tests/test_meshes.py took 745 ms through the hierarchy against
        // 116 ms through the exhaustive loop before this): the wave answers each of its <= 4 rays with the exhaustive loop of
        // scene.cpp:95-116 -- lane l tests triangles l, l + 64, ... (ascending, strict '<'), then the lexicographic minimum of (key,
        // index) over the wave -- which is the answer by definition.  Taken while (live rays) x (triangles) <= kCoopTriangleRays, i.e. at most
        // 768 tests per lane: about what a wave's pass through the hierarchy costs whatever the number of its live lanes. ----
        if (GEOM == 1) {
            unsigned long long todo = __ballot(alive);
            coop = todo != 0ull && (uint32_t)__popcll(todo) * M.ntris <= kCoopTriangleRays;
            if (coop) {
                while (todo != 0ull) {
                    const int rl = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const f3 ro = mk(__shfl(p.o.x, rl), __shfl(p.o.y, rl), __shfl(p.o.z, rl)), rd = mk(__shfl(p.d.x, rl), __shfl(p.d.y, rl), __shfl(p.d.z, rl));
                    uint32_t wk = kMeshInfKey, wi = 0xFFFFFFFFu;
                    auto test = [&](const float4 r0, const float4 r1, const float4 r2, uint32_t i) {
                        float u, v;
                        const float tt = tri_test(r0, r1, r2, ro, rd, u, v);
                        const uint32_t key = __float_as_uint(tt) - 1u;
                        const bool better = key < wk;                                   // ascending i per lane: strict '<' keeps the lowest index
                        wk = better ? key : wk; wi = better ? i : wi;
                    };
                    uint32_t i = lane;
                    for (; i + 192u < M.ntris; i += 256u) {                             // four records per lane in flight per round trip (a chain of bounces is latency)
                        const float4* r = M.tris + 3 * (size_t)i;
                        const float4 a0 = r[0], a1 = r[1], a2 = r[2], b0 = r[192], b1 = r[193], b2 = r[194];
                        const float4 c0 = r[384], c1 = r[385], c2 = r[386], d0 = r[576], d1 = r[577], d2 = r[578];
                        test(a0, a1, a2, i); test(b0, b1, b2, i + 64u); test(c0, c1, c2, i + 128u); test(d0, d1, d2, i + 192u);
                    }
                    for (; i < M.ntris; i += 64u) test(M.tris[3 * (size_t)i], M.tris[3 * (size_t)i + 1], M.tris[3 * (size_t)i + 2], i);
#pragma unroll 1
                    for (int off = 32; off > 0; off >>= 1) {
                        const uint32_t k2 = (uint32_t)__shfl_xor((int)wk, off), i2 = (uint32_t)__shfl_xor((int)wi, off);
                        const bool better = (k2 < wk) | ((k2 == wk) & (i2 < wi));
                        wk = better ? k2 : wk; wi = better ? i2 : wi;
                    }
                    if ((int)lane == rl) { coop_tri = wk < kMeshInfKey ? wi : 0xFFFFFFFFu; t = __uint_as_float(wk + 1u); }
                }
            }
        }
        // ---- closest hit over all triangles (whole workgroup; idle lanes only help staging) ----
        const uint32_t tri = GEOM == 2 ? (coop ? coop_tri : closest_sphere_bvh(K, M, M.bvh_nodes, M.bvh_tris, M.bvh_index, reinterpret_cast<uint32_t*>(s_tile), alive, p.o, p.d, t))
                           : GEOM == 1 ? (coop ? coop_tri : closest_triangle_bvh<kMeshArgOffset>(M, reinterpret_cast<uint32_t*>(s_tile), alive, M.cam_cull != 0u && p.depth == 0u, p.o, p.d, t))
                           : nalive <= (uint32_t)kFewRays ? closest_triangle_few(M.tris, M.ntris, s_tile, nalive, alive, p.o, p.d, t)
                                       : closest_triangle(M.tris, M.ntris, s_tile, alive, p.o, p.d, t);
        if (alive) {
            ++nbounce;
            if (tri == 0xFFFFFFFFu) {
                alive = false;                                                   // smallpt.cpp:168 miss (D13)
            } else {
                // ---- shadePaths, smallpt.cpp:170-263 under D2-D6, D18, D19 ----
                const MeshHit h = GEOM == 2 ? make_sphere_hit(K, tri, t, p.o, p.d) : make_hit(M, tri, t, p.o, p.d);
                const float4* const mats = GEOM == 2 ? K.mat : M.mats;
                const float4 me = mats[3 * h.inst + 0], mc = mats[3 * h.inst + 1];
                const int refl = __float_as_int(me.w) & 3;
                const f3 n = h.n;                                                // :173, un-normalised
                const f3 nl = dot(n, p.d) < 0 ? n : neg(n);                      // :174 (D2)
                f3 f = mk(mc.x, mc.y, mc.z);                                     // :175
                acc = acc + p.w * mk(me.x, me.y, me.z);                          // :179 (D4)
                bool cont = true;
                if (p.depth > 5) {                                               // :188 (D5)
                    if (rng_draw(p.rbase, k1) < mc.w) { const float4 mf = mats[3 * h.inst + 2]; f = mk(mf.x, mf.y, mf.z); }
                    else cont = false;
                }
                if (cont) {
                    const f3 off = nl * 0.02f;                                   // :172 (D3)
                    f3 no = h.x + off, nd, nf = f;
                    if (refl == 0) {                                             // DIFF :208-215
                        const uint32_t u1bits = rng_draw_bits(p.rbase + kGolden, k1);
                        const float r2 = rng_draw(p.rbase + 2u * kGolden, k1);
                        const float r2s = sqrt_exact(r2);
                        float sn, cs;
                        sincos2pi_bits(u1bits, sn, cs);                          // D17
                        const f3 ww = nl;
                        const bool ay = __builtin_fabsf(ww.x) >= 0.1f;          // (double)fabs(w.x) > .1, :211
                        // cross((0,1,0), w) = (w.z, 0, -w.x); cross((1,0,0), w) = (0, -w.z, w.y): products with the axis' zeros only
                        // add signed zeros that never reach a non-zero value or a comparison
                        const f3 ur = mk(ay ? ww.z : 0.f, ay ? 0.f : -ww.z, ay ? -ww.x : ww.y);
                        const f3 uu = normalize<true>(ur);
                        const f3 vv = cross(ww, uu);
                        nd = normalize<true>(uu * cs * r2s + vv * sn * r2s + ww * sqrt_exact(1 - r2));   // :212
                    } else {
                        nd = p.d - n * 2.0f * dot(n, p.d);                       // :218 reflRay
                        if (refl == 2) {                                         // REFR :225-263
                            const bool into = dot(n, nl) > 0;
                            const float nnt = into ? 1.0f / 1.5f : 1.5f / 1.0f;
                            const float ddn = dot(p.d, nl);
                            const float cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
                            if (!(cos2t < 0)) {
                                const f3 tdir = normalize<true>(p.d * nnt - n * ((into ? 1.0f : -1.0f) * (ddn * nnt + sqrt_exact(cos2t))));
                                const float R0 = (0.5f * 0.5f) / (2.5f * 2.5f);
                                const float cc = 1 - (into ? -ddn : dot(tdir, n));
                                const float c2 = cc * cc;
                                const float Re = R0 + (1 - R0) * c2 * c2 * cc;
                                const float Tr = 1 - Re;
                                const f3 xin = h.x - off;
                                if (p.depth <= 2) {                              // :248 split (D6)
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
                    if (p.depth >= SPT_K_MAX_DEPTH) { ++nkill; cont = false; }
                    else cont = !(p.w.x == 0.f && p.w.y == 0.f && p.w.z == 0.f);
                }
                alive = cont;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) { nbounce += __shfl_down(nbounce, off); nkill += __shfl_down(nkill, off); }
    if (lane == 0) {
        atomicAdd(&K.counters[0], nbounce);
        if (nkill) atomicAdd(&K.counters[1], nkill);
    }
}

}  // namespace spt

// exhaustive: one tile of triangle records; hierarchy: 32 stack entries per thread
extern "C" size_t spt_mesh_lds_bytes(int bvh) { return bvh ? (size_t)spt::kMeshBlock * 32u * 4u : (size_t)spt::kTile * 48u; }
extern "C" size_t spt_mesh_stack_floats(uint32_t blocks) { return (size_t)blocks * spt::kMeshBlock * 36u; }

extern "C" hipError_t spt_mesh_launch(const spt::KParams* K, const spt::MParams* M, uint32_t blocks, hipStream_t stream)
{
    if (M->sphere_mode) hipLaunchKernelGGL(spt::meshkernel<2>, dim3(blocks), dim3(spt::kMeshBlock), spt_mesh_lds_bytes(1), stream, *K, *M);
    else if (M->bvh_nodes) hipLaunchKernelGGL(spt::meshkernel<1>, dim3(blocks), dim3(spt::kMeshBlock), spt_mesh_lds_bytes(1), stream, *K, *M);
    else hipLaunchKernelGGL(spt::meshkernel<0>, dim3(blocks), dim3(spt::kMeshBlock), spt_mesh_lds_bytes(0), stream, *K, *M);
    return hipGetLastError();
}

extern "C" hipError_t spt_mesh_trace_rays(const spt::MParams* M, const float* d_rays, uint64_t nrays, float* d_hits, hipStream_t stream)
{
    const uint64_t blocks = (nrays + spt::kMeshBlock - 1) / spt::kMeshBlock;
    if (M->bvh_nodes) hipLaunchKernelGGL(spt::trace_rays<true>, dim3((unsigned)blocks), dim3(spt::kMeshBlock), spt_mesh_lds_bytes(1), stream, *M, d_rays, nrays, d_hits);
    else hipLaunchKernelGGL(spt::trace_rays<false>, dim3((unsigned)blocks), dim3(spt::kMeshBlock), spt_mesh_lds_bytes(0), stream, *M, d_rays, nrays, d_hits);
    return hipGetLastError();
}
