// spt_tribvh.h -- what makes the triangle hierarchy (SPT_ACCEL_BVH of spt_set_mesh_accel) return the exhaustive loop's hit for EVERY
// ray: the constants, the per-ray query and the node tests shared by the gfx950 traversal (spt_mesh.hip), the host builder
// (spt_bvh.cpp) and the CPU harness that runs the very same functions against the exhaustive loop (tests/sanitize/tribvh_main.cpp).
//
// The problem.  triIntersect (scene.cpp:52-70) divides by det = dot(rd, cross(e1, e2)) without a cut-off (:62).  When det is zero
// to rounding the quotients u, v, t are noise, and noise passes the barycentric test (:67) with a probability that is not small:
// the exhaustive loops of the reference (scene.cpp:95-116, smallpt.cpp:443-458) then report a triangle at a distance that has
// nothing to do with where the triangle is, and no bounding volume contains such a "hit".  Rounds 2 and 3 documented the rays
// concerned as an exception; this round removes it.  Every test below is the reference's arithmetic on a triangle of the scene,
// so testing MORE triangles never changes the answer; the only obligation is that every triangle whose reported key beats or ties
// the final answer is tested.  Three structures share that obligation.
//
// Notation: u = 2^-24; triangle record v0, e1 = fl(v1 - v0), e2 = fl(v2 - v0), n = fl(cross(e1, e2)), all taken as exact data;
// n* = e1 x e2 exactly, E = |e1||e2|, g = E / |n*| (1 / sine of the angle at v0), e = the longer of |e1|, |e2|; r = ro - v0, R = |r|.
// True values det* = rd.n*, u* = -(r x rd).e2 / det*, v* = (r x rd).e1 / det*, t* = -(n*.r) / det*: ro + t* rd = v0 + u* e1 + v* e2.
//
// (0) rounding of the four dot products as tri_test evaluates them (one rounding per operation, no contraction):
//       |det   - det*      | <= 7.1 u |rd| E        (3 u |rd||n| for the dot, 4 u |rd| E for n against n*)
//       |num_u - u* det*   | <= 8.1 u R |rd| |e2|   (cross(rov0, rd): 4 u, rov0 = fl(ro - v0): 1 u, the dot: 3 u)
//       |num_v - v* det*   | <= 8.1 u R |rd| |e1|
//       |num_t - t* det*   | <= 8.1 u E R
//     and u_c = num_u / det (1 + 2.1 u) etc.  A triangle is ACCEPTED when 0 <= u_c, 0 <= v_c, u_c <= 1, fl(u_c + v_c) <= 1 and
//     t_c > 0; then |num_u|, |num_v| <= |det| (1 + 3 u).  E = 0 (an edge of length zero) gives det = 0, d = inf and t = NaN:
//     never accepted, such triangles are in no structure.
//
// (1) REGULAR triangles (g <= kTriThinG), rays with |rd.n*| >= M u |rd| E for some M >= M0 = 2^11 (tau0 = kTriBand = M0 u = 2^-13: the
//     direction makes an angle of at least tau0 g with the plane).  Then det = det* (1 + th), |th| <= 7.1 / M, and with (0):
//       |u_c - u*| |e1| <= (8 |e1| + 8.2 R) / M,  |v_c - v*| |e2| <= (8 |e2| + 8.2 R) / M,  |t_c - t*| |rd| <= (16.1 R + 17 e) / M,
//     so the reported point P_c = ro + t_c rd lies within (32.5 R + 33 e) / M of the triangle: the steeper the ray meets the plane, the
//     smaller the error.  The spatial hierarchy (binned SAH, boxes padded by e / 4 + 1e-4 of the largest coordinate at build time)
//     knows, per child, a CONE of its triangles' normals (axis a, chord radius kappa, either orientation), the largest g and the
//     longest edge e_max below it.  Every triangle inside has |nh.rdh| >= c := |a.rdh| - kappa, i.e. M >= c / (u g_max) when that is
//     >= M0; otherwise its triangles either have M >= M0 or belong to (2).  With D = distance from ro to the box's farthest corner
//     (R <= D) the child box is inflated by
//       pad = 37 u (D + 1.016 e_max) / max(c / g_max, tau0) * 1.03
//     -- 2e-6 D for a patch facing the ray, 1.8 % of D for one seen edge-on or one whose normals spread widely (the upper levels).
//     37 = 32.5 + 4.5: the slab test itself (differences, products with the 1-ulp v_rcp of rd) misplaces a face by up to 3 u D, which
//     must not eat into the bound where the pad is at its smallest (c = 1); the 3 % cover the 1-ulp v_sqrt in D and the rounding of
//     a.rdh (3 u against tau0 = 2^-13).  Coordinates and edges are assumed to stay clear of underflow and overflow (1e-15 ... 1e15).
//     P_c is in every inflated ancestor box at the parameter t_c <= the current nearest: the triangle is reached.
//
// (2) REGULAR triangles, rays with |rd.n*| < M0 u |rd| E (the direction lies in the plane to within tau = kTriBand g; M = M0 below).  Acceptance
//     bounds the in-plane part of the line's moment m = r x rd about v0: |m.e2| <= |det| + 8.1 u R |rd||e2| and the same with e1,
//     hence (dual basis of e1, e2, lengths |e_j| / |n*|)  |m_par| <= g u |rd| [(M + 7.1)(|e1| + |e2|) + 16.2 R].  Written with the
//     plane's unit normal nh, r = r_n nh + r_par, rd = d_n nh + d_par:  m_par = nh x (r_n d_par - d_n r_par), so
//       |r_n| <= tau (R + 2 e) (1 + 2^-10):   the ORIGIN lies in the plane to within tau (R + 2 e)   -- (B)
//     besides                |nh.rd| < tau |rd|                                                              -- (A).
//     A ray that satisfies (A) but not (B) is rejected by tri_test; one that violates (A) is case (1).  The triangles with (A) and
//     (B) are found by a second tree over the regular triangles, clustered by normal and position, whose children carry a CONE of
//     normals (axis a, chord radius kappa; nh and -nh are the same plane), a reference point p, sigma >= |nh_i.(v0_i - p)| (how far
//     the planes pass from p: p is the least-squares meeting point of the cluster's planes), rho >= |v0_i - p|, the largest tau
//     and te >= tau_i (rho + 2 e_i).  With dp = p - ro = s rdh + w (w orthogonal to rdh):  (A) gives |a.rdh| <= kappa + tau;
//     nh_i.(ro - v0_i) = -nh_i.dp + nh_i.(p - v0_i) and nh_i.dp = s (nh_i.rdh) + nh_i.w turn (B) into
//       |a.w| <= kappa |w| + sigma + 2.002 tau |dp| + te.
//     A leaf child is one triangle (a = nh, p = v0, kappa = sigma = 0: its exact condition).  A ray generically lies in NO plane
//     of a smooth surface, and a patch facing the ray fails (A), a patch seen edge-on from beside fails (B): the walk ends early.
//
// (3) THIN triangles (g > kTriThinG: the angle at v0 is below 1 / 32 or above pi - 1 / 32; the 2 * 2L needles makeSphereTriMesh puts
//     at the poles, scene.cpp:13-27, have g ~ 10^6 and a normal that is noise for every ray; also n* = 0).  With eh the unit
//     direction of the longer edge e_L, acceptance gives |m.e_L| <= |det|(1 + 3u) + 8.1 u R |rd||e_L| and |det| <= |rd|(|n*| + 7.1 u E):
//       |(r x rdh).eh| <= e_S (1 / g + 7.2 u) + 8.1 u R       (e_S = the shorter edge; e_S / g = the triangle's height over e_L)
//     -- the ray's LINE passes within that of the infinite line through v0 along eh (times the sine of their angle), wherever along
//     it: the case round 3 left open (a line crossing a needle's supporting line 43 units beyond its tip).  A third tree over the
//     thin triangles: children carry a cone of the directions eh (axis a, kappa, either orientation), a reference point p (the
//     least-squares meeting point of the cluster's lines: the pole, for the needles of a pole) and lam >= dist(p, line_i) + a_i +
//     8.1 u |v0_i - p|, a_i = e_S (1 / g + 7.2 u).  With dp = p - ro and m = dp x rdh (the ray's moment about p, |m| = its distance
//     from p):  (r x rdh).eh = -m.eh + ((p - v0) x rdh).eh and the last term is at most dist(p, line_i), so a child is entered when
//       |a.m| <= kappa |m| + lam + 32 u |dp|.
//     The needles of a pole are a fan of lines through one point: a ray's moment picks two azimuths of it, the walk is logarithmic.
//
// Float evaluation of the node tests: rdh carries 2 u; a.w = a.dp - s (a.rdh), |w|^2 = |dp|^2 - s^2 and dp x rdh cancel to within
// 6 u |dp| -- both sides of (B) carry 16 u |dp| (tau >= 2^-13 makes the 2.002 -> 2.004 of its |dp| term 2.4e-7 |dp| more), the line
// test 32 u |dp| (8.1 u |dp| of which is the bound's).  The builder works in double and rounds kappa, sigma, lam, te up against the
// float axes and points it stores.
#ifndef SPT_TRIBVH_H
#define SPT_TRIBVH_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SPT_THD __host__ __device__ __forceinline__
#else
#define SPT_THD inline
#endif

// The node tests are not the reference's arithmetic (tri_test is, and lives elsewhere): they may contract a * b + c into one fused operation
// -- fewer VALU instructions, smaller rounding error; the thresholds' slack covers either form (the CPU harness runs the unfused one).
#if defined(__clang__)
#define SPT_TRI_FUSE _Pragma("clang fp contract(fast)")
#else
#define SPT_TRI_FUSE
#endif

namespace spt {

constexpr double kTriBand = 1.0 / 8192.0;         // tau0 = M0 u, M0 = 2^11: (1) / (2) split at |nh.rdh| = kTriBand * g
constexpr double kTriThinG = 32.0;                // g above which a triangle is THIN (3)
constexpr uint32_t kTriFlatLines = 16384;         // up to this many thin triangles are scanned as a table instead of walked as a tree (3)
constexpr float kTriPadK = 37.0f * 0x1p-24f * 1.03f;   // (1): pad = kTriPadK (D + e') / max((|a.rdh| - kappa) / g_max, tau0), e' = 1.016 e_max; 37 = 32.5 + 4.5 for the slab arithmetic

// Per ray (closest-hit query): origin and unit direction (a NaN / inf / zero direction makes every comparison below false: nothing is
// visited, and the exhaustive loop reports nothing either -- det = NaN or 0 gives t = NaN or inf).
struct TriQuery {
    float o[3];
    float h[3];                                    // rdh = rd / |rd|
};

SPT_THD float tri_rsq(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);               // 1 ulp: inside the slack of the thresholds
#else
    return 1.0f / __builtin_sqrtf(x);
#endif
}
SPT_THD float tri_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return __builtin_sqrtf(x);
#endif
}

SPT_THD void tri_query(float ox, float oy, float oz, float dx, float dy, float dz, TriQuery& q)
{
    SPT_TRI_FUSE
    const float inv = tri_rsq(dx * dx + dy * dy + dz * dz);
    q.o[0] = ox; q.o[1] = oy; q.o[2] = oz;
    q.h[0] = dx * inv; q.h[1] = dy * inv; q.h[2] = dz * inv;
}

// (2): a child of the plane tree = cone of normals (axis a, chord radius kappa, either orientation), reference point p, sigma >=
// |nh_i.(v0_i - p)|, the largest tau below it and te >= tau_i (rho + 2 e_i) 1.001 (rho >= |v0_i - p|).  With dp = p - ro = s rdh + w:
//   (A)  |a.rdh| <= kappa + tau          (B)  |a.w| <= kappa |w| + sigma + 2.002 tau |dp| + te
// An empty child has kappa = -1e30.  Float evaluation: a.w = a.dp - s (a.rdh) and |w|^2 = |dp|^2 - s^2 cancel: both sides carry 16 u |dp|.
SPT_THD bool tri_plane_child(const TriQuery& q, float ax, float ay, float az, float kappa, float px, float py, float pz, float sigma, float tau, float te)
{
    SPT_TRI_FUSE
    const float ah = ax * q.h[0] + ay * q.h[1] + az * q.h[2];
    const float dx = px - q.o[0], dy = py - q.o[1], dz = pz - q.o[2];
    const float s = dx * q.h[0] + dy * q.h[1] + dz * q.h[2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float aw = (ax * dx + ay * dy + az * dz) - s * ah;
    const float w = tri_sqrt(__builtin_fmaxf(d2 - s * s, 0.0f) + 0x1p-20f * d2);
    const float d = tri_sqrt(d2);
    const bool A = __builtin_fabsf(ah) <= kappa + tau * (1.0f + 0x1p-10f) + 0x1p-20f;
    const bool B = __builtin_fabsf(aw) <= kappa * w + sigma + (2.004f * tau + 0x1p-20f) * d + te;
    return A & B;
}

// (3): a child of the line tree = cone of the long edges' directions (axis a, chord radius kappa, either orientation), reference
// point p, lam >= dist(p, line_i) + a_i + 8.1 u |v0_i - p|.  With dp = p - ro and the ray's moment m = dp x rdh about p:
//   |a.m| <= kappa |m| + lam + 32 u |dp|        (8.1 u |dp| of the bound, the rest for the float cross product)
SPT_THD bool tri_line_child(const TriQuery& q, float ax, float ay, float az, float kappa, float px, float py, float pz, float lam)
{
    SPT_TRI_FUSE
    const float dx = px - q.o[0], dy = py - q.o[1], dz = pz - q.o[2];
    const float mx = dy * q.h[2] - dz * q.h[1], my = dz * q.h[0] - dx * q.h[2], mz = dx * q.h[1] - dy * q.h[0];
    const float am = ax * mx + ay * my + az * mz;
    const float m = tri_sqrt(mx * mx + my * my + mz * mz);
    const float d = tri_sqrt(dx * dx + dy * dy + dz * dz);
    return __builtin_fabsf(am) <= kappa * (m * 1.001f + 0x1p-19f * d) + lam + 0x1p-19f * d;
}

// (1): one child of the spatial hierarchy against the ray segment [0, tcut]: entry parameter in `tn`.  The box [lo, hi] - ro is given
// component-wise; iv = 1 / rd per component (inf for a zero component: NaN from 0 * inf drops out of min / max, the conservative
// side); (hx, hy, hz) = rdh; the child's cone = axis (ax, ay, az), chord radius kappa, iq = 1 / g_max, ee = 1.016 e_max.
// EXACT = false: the plain hierarchy of SPT_ACCEL_BVH_FAST -- no inflation, the build-time padding only.
template <bool EXACT>
SPT_THD bool tri_box_child(float l0x, float l1x, float l0y, float l1y, float l0z, float l1z, float ivx, float ivy, float ivz, float tcut,
                           float hx, float hy, float hz, float ax, float ay, float az, float kappa, float iq, float ee, float& tn)
{
    SPT_TRI_FUSE
    float p = 0.0f;
    if (EXACT) {
        const float mx = __builtin_fmaxf(__builtin_fabsf(l0x), __builtin_fabsf(l1x));
        const float my = __builtin_fmaxf(__builtin_fabsf(l0y), __builtin_fabsf(l1y));
        const float mz = __builtin_fmaxf(__builtin_fabsf(l0z), __builtin_fabsf(l1z));
        const float dfar = tri_sqrt(mx * mx + my * my + mz * mz) * 1.001f;                 // D (1 ulp sqrt)
        const float c = (__builtin_fabsf(ax * hx + ay * hy + az * hz) - kappa) * iq;       // every triangle inside: |nh.rdh| / g >= c
        const float m = __builtin_fmaxf(c, (float)kTriBand);
#if defined(__HIP_DEVICE_COMPILE__)
        p = (dfar + ee) * kTriPadK * __builtin_amdgcn_rcpf(m);
#else
        p = (dfar + ee) * kTriPadK / m;
#endif
    }
    const float x0 = (l0x - p) * ivx, x1 = (l1x + p) * ivx;
    const float y0 = (l0y - p) * ivy, y1 = (l1y + p) * ivy;
    const float z0 = (l0z - p) * ivz, z1 = (l1z + p) * ivz;
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    return (tf >= 0.0f) & (tn <= tf * 1.0001f) & (tn <= tcut);      // an empty child (inverted box) gives tn = +inf or NaN: never entered
}

// ---- the three walks.  Stack: push(sp, v) / pop(sp) of a per-ray stack of <= 32 child references (LDS on the device, an array
// in the harness).  The trees' depth bound (spt_bvh.h kBvhMaxDepth) is what keeps it from overflowing.

// (1) spatial hierarchy, near child first.  leaf(first, count) tests the leaf-order triangles [first, first + count) and lowers
// `tcut` (1.0001 x the current nearest distance) when one of them becomes the answer.
template <bool EXACT, class Stack, class Leaf>
SPT_THD void tri_walk_boxes(const float4* __restrict__ nodes, const float4* __restrict__ cones, float ox, float oy, float oz, float ivx, float ivy, float ivz,
                            float hx, float hy, float hz, float& tcut, Stack& st, Leaf&& leaf)
{
    uint32_t sp = 0;
    int cur = 0;                                               // the root is always node 0
    for (;;) {
        if (cur >= 0) {
            const float4* nd = nodes + 4 * (size_t)cur;
            const float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
            float4 cl = a, cr = a, ce = a;                     // (unused by the fast form)
            if (EXACT) {
                const float4* cn = cones + 3 * (size_t)cur;    // {left axis, kappa} {right axis, kappa} {left 1/g, left e', right 1/g, right e'}
                cl = cn[0]; cr = cn[1]; ce = cn[2];
            }
            float ln, rn;
            const bool hl = tri_box_child<EXACT>(a.x - ox, a.w - ox, a.y - oy, b.x - oy, a.z - oz, b.y - oz, ivx, ivy, ivz, tcut, hx, hy, hz, cl.x, cl.y, cl.z, cl.w, ce.x, ce.y, ln);
            const bool hr = tri_box_child<EXACT>(b.z - ox, c.y - ox, b.w - oy, c.z - oy, c.x - oz, c.w - oz, ivx, ivy, ivz, tcut, hx, hy, hz, cr.x, cr.y, cr.z, cr.w, ce.z, ce.w, rn);
            const int lref = __builtin_bit_cast(int, d.x), rref = __builtin_bit_cast(int, d.y);
            if (hl & hr) {
                const bool left_first = ln <= rn;
                st.push(sp, (uint32_t)(left_first ? rref : lref));
                ++sp;
                cur = left_first ? lref : rref;
                continue;
            }
            if (hl | hr) { cur = hl ? lref : rref; continue; }
        } else {
            const uint32_t code = (uint32_t)~cur;
            leaf(code >> 4, code & 15u);
        }
        if (sp == 0u) break;
        --sp;
        cur = (int)st.pop(sp);
    }
}

// (2) / (3): cone trees; a child that is a single triangle is tested as soon as its (then exact) condition holds (tri(g): global
// triangle g).  Node layouts: spt_bvh.h.
template <class Stack, class Tri>
SPT_THD void tri_walk_planes(const float4* __restrict__ nodes, const TriQuery& q, Stack& st, Tri&& tri)
{
    uint32_t sp = 0;
    int cur = 0;
    for (;;) {
        const float4* nd = nodes + 6 * (size_t)cur;
        const float4 la = nd[0], lp = nd[1], ra = nd[2], rp = nd[3], t = nd[4], e = nd[5];
        bool hl = tri_plane_child(q, la.x, la.y, la.z, la.w, lp.x, lp.y, lp.z, lp.w, t.x, t.y);
        bool hr = tri_plane_child(q, ra.x, ra.y, ra.z, ra.w, rp.x, rp.y, rp.z, rp.w, t.z, t.w);
        const int lref = __builtin_bit_cast(int, e.x), rref = __builtin_bit_cast(int, e.y);
        if (hl & (lref < 0)) { tri((uint32_t)~lref); hl = false; }
        if (hr & (rref < 0)) { tri((uint32_t)~rref); hr = false; }
        if (hl & hr) { st.push(sp, (uint32_t)rref); ++sp; cur = lref; continue; }
        if (hl | hr) { cur = hl ? lref : rref; continue; }
        if (sp == 0u) break;
        --sp;
        cur = (int)st.pop(sp);
    }
}

template <class Stack, class Tri>
SPT_THD void tri_walk_lines(const float4* __restrict__ nodes, const TriQuery& q, Stack& st, Tri&& tri)
{
    uint32_t sp = 0;
    int cur = 0;
    for (;;) {
        const float4* nd = nodes + 5 * (size_t)cur;
        const float4 la = nd[0], lp = nd[1], ra = nd[2], rp = nd[3], e = nd[4];
        bool hl = tri_line_child(q, la.x, la.y, la.z, la.w, lp.x, lp.y, lp.z, lp.w);
        bool hr = tri_line_child(q, ra.x, ra.y, ra.z, ra.w, rp.x, rp.y, rp.z, rp.w);
        const int lref = __builtin_bit_cast(int, e.x), rref = __builtin_bit_cast(int, e.y);
        if (hl & (lref < 0)) { tri((uint32_t)~lref); hl = false; }
        if (hr & (rref < 0)) { tri((uint32_t)~rref); hr = false; }
        if (hl & hr) { st.push(sp, (uint32_t)rref); ++sp; cur = lref; continue; }
        if (hl | hr) { cur = hl ? lref : rref; continue; }
        if (sp == 0u) break;
        --sp;
        cur = (int)st.pop(sp);
    }
}

// ---- (3), table form.  A per-lane tree walk on a GPU is a chain of dependent, scattered loads; a loop over a table that every lane of a
// wave reads at the same index is not: read through the constant address space it is a scalar load (s_load_dwordx8 into SGPRs) and the
// test a few VALU instructions with SGPR operands.  Up to kTriFlatLines thin triangles are therefore kept as a table that every ray
// scans (layout and test below); the listed triangles are tested.  (The same form was tried for the PLANES of small scenes --
// one {nh, c0 g} per triangle, c0 = 2^-9, boxes inflated by 2^-8 D: a ray lists 0.3 % of the triangles, a wave of 64 unrelated rays
// a fifth of them, and testing the listed ones cost more than the whole exhaustive loop: 33 ms against 29 ms for a 1280 x 720 frame
// of the shipped 8192-triangle scene.  Not kept: profiles/r04_triangle_hierarchy.txt.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) float4 tri_flat_t;
#else
typedef const float4 tri_flat_t;
#endif
// The table is a sequence of GROUPS: a header {p, number of records} followed by that many records {eh, tol}; p is a point all the
// group's lines pass (nearly) through -- the pole, for the needles of a pole: their long edges all end there --, tol = dist(p, line) + a +
// 8.1 u |v0 - p|.  Per group the ray's moment about p, m = (p - ro) x rdh, is formed once, a record then costs a dot product:
// |eh.m| <= tol + 32 u |p - ro|   ((3) with a cone of one direction).  index[slot] = the record's global triangle.
template <class Stack, class Cand>
SPT_THD void tri_scan_lines(const float4* __restrict__ flat_, const uint32_t* __restrict__ index_, uint32_t nslots, const TriQuery& q, Stack& st, Cand&& cand)
{
    SPT_TRI_FUSE
    tri_flat_t* flat = (tri_flat_t*)flat_;
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) uint32_t* index = (const __attribute__((address_space(4))) uint32_t*)index_;
#else
    const uint32_t* index = index_;
#endif
    uint32_t sp = 0;
    for (uint32_t i = 0; i < nslots;) {
        const float4 h = flat[i];
        const uint32_t cnt = __builtin_bit_cast(uint32_t, h.w);
        const float dx = h.x - q.o[0], dy = h.y - q.o[1], dz = h.z - q.o[2];
        const float mx = dy * q.h[2] - dz * q.h[1], my = dz * q.h[0] - dx * q.h[2], mz = dx * q.h[1] - dy * q.h[0];
        const float thr = 0x1p-19f * tri_sqrt(dx * dx + dy * dy + dz * dz);
        for (uint32_t k = 1; k <= cnt; ++k) {
            const float4 e = flat[i + k];
            if (__builtin_fabsf(e.x * mx + e.y * my + e.z * mz) <= e.w + thr) {
                st.push(sp, index[i + k]);
                if (++sp == 32u) { while (sp) { --sp; cand(st.pop(sp)); } }
            }
        }
        i += cnt + 1u;
    }
    while (sp) { --sp; cand(st.pop(sp)); }
}

}  // namespace spt
#endif
