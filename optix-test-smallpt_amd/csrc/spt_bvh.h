// spt_bvh.h -- host-side builder of the optional bounding-volume hierarchy over a mesh scene's triangles (SPT_ACCEL_BVH).
//
// The reference's CPU path tests every triangle (scene.cpp:95-116, smallpt.cpp:443-458); its GPU path hands the same
// query to OptiX Prime (smallpt.cpp:475-603), whose acceleration structure is not in the repository.  This is the
// stand-in for the latter: a binary hierarchy with both children's boxes in the parent (one 64-byte node per visit),
// leaves of <= 4 triangles, depth <= kBvhMaxDepth so that a 32-entry per-lane stack can never overflow.
//
// Contract (include/smallpt_mi355x.h, spt_set_mesh_accel): the traversal evaluates the SAME triIntersect arithmetic on
// the triangles it visits and selects by the same rule (smallest t > 0, lowest (instance, triangle) index among equal
// t), so it returns the exhaustive loop's hit whenever it visits that triangle.  Boxes are padded by a quarter of the
// triangle's longest edge (+ 1e-4 of its largest coordinate) and the slab tests are widened, which covers every hit
// whose ray passes within rounding distance of its triangle; triIntersect has no determinant cut-off, so for a ray
// lying (to ~1e-7 rad) in a triangle's plane it can also report hits with no geometric relation to the triangle, which
// no bounding volume contains.  The exhaustive kernel therefore stays the default and the parity anchor; tests compare
// the two on millions of rays, adversarial ones included.
#ifndef SPT_BVH_H
#define SPT_BVH_H
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace spt {

constexpr uint32_t kBvhMaxDepth = 32;      // children of the root are at depth 1; a leaf reference sits at depth <= 32
constexpr uint32_t kBvhLeafTris = 4;         // primitives per leaf (triangles or spheres)
constexpr uint32_t kBvhAlways = 32;          // sphere hierarchies: at most this many outsized spheres are tested for every ray

// Child reference: >= 0 = node index; < 0 = leaf, ~ref = (first leaf-order triangle << 3) | count (count 0 = empty).
struct Bvh {
    std::vector<float4> nodes;        // 4 x float4 per node: {lmin.xyz, lmax.x} {lmax.yz, rmin.xy} {rmin.z, rmax.xyz} {left, right, 0, 0}
    std::vector<float4> tris;         // 3 x float4 per triangle in leaf order (the records of MParams::tris)
    std::vector<uint32_t> index;      // global (instance-major) triangle index of every leaf-order triangle
    std::vector<uint32_t> always;     // sphere hierarchies: global indices (ascending) of the spheres kept out of the tree
    uint32_t depth = 0, leaves = 0;
    // Triangle hierarchies: THIN triangles (area <= 2^-10 of the longest edge squared: the needles makeSphereTriMesh puts at the poles,
    // scene.cpp:13-27) live in a second hierarchy of the same layout that is traversed along the ray's whole LINE without a distance
    // cut.  triIntersect divides by dot(rd, cross(e1, e2)) (scene.cpp:62), which is rounding noise for such a triangle whatever the ray:
    // a ray passing within rounding distance of the needle gets a "hit" whose distance has nothing to do with where the needle is, so
    // the exhaustive loop's answer can only be reproduced by testing the needle whenever the line meets its padded box.
    std::vector<float4> thin_nodes, thin_tris;
    std::vector<uint32_t> thin_index;
    uint32_t thin_count = 0;
};
constexpr double kBvhThinRatio = 1.0 / 1024.0;   // |cross(e1, e2)| <= ratio * (longest edge)^2

// recs: ntris x 3 records {v0, n.x} {v1 - v0, n.y} {v2 - v0, n.z}.  Throws std::runtime_error on non-finite vertices.
void build_bvh(const float4* recs, uint32_t ntris, Bvh& out);
// Structural check used by the CPU tests: every triangle in exactly one leaf, every box contains its subtree's padded
// triangles, depth bound respected.  Returns false and a reason on failure.
bool validate_bvh(const float4* recs, uint32_t ntris, const Bvh& bvh, std::string& why);

// The same hierarchy over a sphere table (SPT_ACCEL_BVH of spt_set_sphere_accel): geom[i] = {centre, r*r}, radius[i] = r.
// Node boxes are the spheres' own extents (no padding: closest_sphere_bvh inflates every box it tests by a bound on the
// rounding error of intersectAnalytic for THAT ray, which is what makes the traversal provably exhaustive-equivalent);
// `tris` holds one float4 {centre, r*r} per sphere in leaf order.
void build_sphere_bvh(const float4* geom, const float* radius, uint32_t n, Bvh& out);
bool validate_sphere_bvh(const float4* geom, const float* radius, uint32_t n, const Bvh& bvh, std::string& why);

}  // namespace spt
#endif
