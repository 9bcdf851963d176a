// spt_bvh.h -- host-side builder of the optional bounding-volume hierarchy over a mesh scene's triangles (SPT_ACCEL_BVH).
//
// The reference's CPU path tests every triangle (scene.cpp:95-116, smallpt.cpp:443-458); its GPU path hands the same
// query to OptiX Prime (smallpt.cpp:475-603), whose acceleration structure is not in the repository.  This is the
// stand-in for the latter: a binary hierarchy with both children's boxes in the parent (one 64-byte node per visit),
// leaves of <= 4 triangles, depth <= kBvhMaxDepth so that a 32-entry per-lane stack can never overflow.
//
// Contract (include/smallpt_mi355x.h, spt_set_mesh_accel): the traversal evaluates the SAME triIntersect arithmetic on
// the triangles it visits and selects by the same rule (smallest t > 0, lowest (instance, triangle) index among equal
// t), so it returns the exhaustive loop's hit whenever it visits that triangle -- and since round 4 it provably visits
// every triangle whose reported key beats or ties the answer (spt_tribvh.h): this spatial hierarchy finds the reports
// whose error is bounded (its child boxes are inflated per ray), a cone tree over the triangles' PLANES finds the rays
// that lie in a regular triangle's plane to rounding (triIntersect has no determinant cut-off, scene.cpp:62, and reports
// noise there), a table (or cone tree) of the long edges' LINES finds the thin triangles (needles, zero area) whatever the ray.
#ifndef SPT_BVH_H
#define SPT_BVH_H
#include <hip/hip_runtime.h>

#include "spt_tribvh.h"

#include <cstdint>
#include <string>
#include <vector>

#ifndef SPT_BVH_LEAF_TRIS
#define SPT_BVH_LEAF_TRIS 12
#endif

namespace spt {

constexpr uint32_t kBvhMaxDepth = 32;      // children of the root are at depth 1; a leaf reference sits at depth <= 32
constexpr uint32_t kBvhLeafTris = SPT_BVH_LEAF_TRIS;   // triangles per leaf (measured on the shipped scene: 2 / 4 / 7 per leaf = 28.7 / 23.3 / 19.1 ms at 256^2 x 256 spp)
constexpr uint32_t kBvhLeafSpheres = 4;      // spheres per leaf (round 2: 2 ... 7 within 3 %)
constexpr uint32_t kBvhAlways = 32;          // sphere hierarchies: at most this many outsized spheres are tested for every ray

// Child reference: >= 0 = node index; < 0 = leaf, ~ref = (first leaf-order triangle << 4) | count (count 0 = empty).
struct Bvh {
    std::vector<float4> nodes;        // 4 x float4 per node: {lmin.xyz, lmax.x} {lmax.yz, rmin.xy} {rmin.z, rmax.xyz} {left, right, 0, 0}
    std::vector<float4> tris;         // 3 x float4 per triangle in leaf order (the records of MParams::tris)
    std::vector<uint32_t> index;      // global (instance-major) triangle index of every leaf-order triangle
    std::vector<uint32_t> always;     // sphere hierarchies: global indices (ascending) of the spheres kept out of the tree
    uint32_t depth = 0, leaves = 0;
    // Triangle hierarchies (spt_tribvh.h): the spatial tree above holds the REGULAR triangles; `planes` is the cone tree over their planes
    // (6 float4 per node: {left axis, kappa} {left p, sigma} {right axis, kappa} {right p, sigma} {left tau, left te, right tau, right te}
    // {left, right, 0, 0}), `lines` the one over the long edges of the THIN triangles (5 float4 per node: {left axis, kappa} {left p, lam}
    // {right axis, kappa} {right p, lam} {left, right, 0, 0}).  Leaves are single triangles: a negative reference r stands for the GLOBAL
    // triangle ~r (record 3 * ~r of the scene's table).  Triangles with an edge of length zero are in no structure (never accepted,
    // spt_tribvh.h (0)).
    std::vector<float4> planes, lines;
    // up to kTriFlatLines thin triangles are kept as a table instead of `lines` (spt_tribvh.h (3)): groups {p, count} {eh, tol} ...;
    // flat_line_index[slot] = the record's global triangle
    std::vector<float4> flat_lines;
    std::vector<uint32_t> flat_line_index;
    std::vector<float4> cones;        // spatial tree, 3 float4 per node: {left axis, kappa} {right axis, kappa} {left 1/g_max, left 1.016 e_max, right ..} (spt_tribvh.h (1))
    uint32_t regular_count = 0, thin_count = 0, dead_count = 0, ball_depth = 0;
    bool flat = false;                // the thin triangles are a table (flat_lines), not a tree (lines)
};

// recs: ntris x 3 records {v0, n.x} {v1 - v0, n.y} {v2 - v0, n.z}.  Throws std::runtime_error on non-finite vertices.
// form: how the thin triangles are kept -- 0 = by their number (a table up to kTriFlatLines), 1 = table, 2 = tree (tests).
void build_bvh(const float4* recs, uint32_t ntris, Bvh& out, int form = 0);
// Structural check used by the CPU tests: every triangle in exactly one leaf, every box contains its subtree's padded
// triangles, depth bound respected.  Returns false and a reason on failure.
bool validate_bvh(const float4* recs, uint32_t ntris, const Bvh& bvh, std::string& why);
// The regular triangles whose plane contains the point o to within (B) of spt_tribvh.h (2) (global indices, ascending): what replaces
// the plane tree for rays whose lines all pass through o and start at most `extra` from it (the rays of depth 0 of a frame: pinhole
// camera extra = 0, smallpt camera extra = push * |d|max).
void camera_planes(const float4* recs, uint32_t ntris, const float o[3], float extra, std::vector<uint32_t>& out);

// The same hierarchy over a sphere table (SPT_ACCEL_BVH of spt_set_sphere_accel): geom[i] = {centre, r*r}, radius[i] = r.
// Node boxes are the spheres' own extents (no padding: closest_sphere_bvh inflates every box it tests by a bound on the
// rounding error of intersectAnalytic for THAT ray, which is what makes the traversal provably exhaustive-equivalent);
// `tris` holds one float4 {centre, r*r} per sphere in leaf order.
void build_sphere_bvh(const float4* geom, const float* radius, uint32_t n, Bvh& out);
bool validate_sphere_bvh(const float4* geom, const float* radius, uint32_t n, const Bvh& bvh, std::string& why);

}  // namespace spt
#endif
