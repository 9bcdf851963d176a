// spt_grid.h -- uniform grid over a large sphere table (the default closest-hit structure for tables above the pool kernel's 24
// spheres): layout, host builder, and the traversal arithmetic shared by the gfx950 kernel (spt_grid.hip) and the CPU harness
// that checks it against the exhaustive loop (tests/sanitize/grid_main.cpp).
//
// What it replaces: intersectGlobalSpheres (smallpt.cpp:54-70) tests EVERY sphere for every ray.  The grid visits only the cells a
// ray crosses, evaluates the SAME intersectAnalytic arithmetic (scene.cpp:129-140, on the integer keys of the sphere kernels) on the
// spheres registered there and selects by the same rule (smallest t > eps, lowest index among equal t).  It returns the exhaustive
// loop's answer for EVERY ray, by construction:
//
//   (1) error of a reported hit.  With u = 2^-24, e = c - o, |d|^2 = 1 + eta, the point p = o + t d of a reported root t satisfies
//       | |p - c|^2 - r^2 | <= 101 u (|e|^2 + r^2) + |eta| t^2   (intersectAnalytic divides by nothing).
//       A ray takes the grid only if  D(o) := distance from o to the farthest corner of the grid box  <= Dmax; every in-grid centre
//       lies in the box, hence |e| <= Dmax and t <= 1.01 (|e| + r) <= 1.5 Dmax.  The walk of a ray is valid up to the parameter
//       t_ok with |eta| t_ok^2 <= 2^-19 (1.5 Dmax)^2:  t_ok = inf when | fl(d.d) - 1 | <= 2^-19 - 2^-22 (so |eta| <= 2^-19: every
//       fresh direction), else 0.99 * 1.5 Dmax * sqrt(2^-19 / (1.25 | fl(d.d) - 1 |)) (a mirror reflection is not renormalised,
//       smallpt.cpp:218, so over a chain of bounces eta drifts; inside a closed mirror ball the hits are a few radii away and stay
//       far below t_ok).  For every root t <= t_ok, with  E_j = 2^-17 (Dmax^2 + r_j^2) + 2^-19 (1.5 Dmax)^2  (128 u >= 101 u), the
//       reported point lies within  sqrt(r_j^2 + E_j)  of c_j.
//   (2) registration.  Sphere j is listed in every cell whose box is within R_j of c_j (Euclidean),  R_j = sqrt(r_j^2 + E_j) + dgrid,
//       dgrid = 2^-11 Dmax; the grid box is the union of the cubes c_j +- R_j.
//   (3) the walk.  tx/ty/tz = parameters at which the ray leaves the current cell along x / y / z, advanced by additions of cell / |d_a|.
//       Where the walk believes a face to be differs from where it is by two terms, per axis: (i) the accumulated rounding of the exit
//       parameters, <= 4 (steps + 4) u t with at most 3 * 128 steps and t <= 1.5 Dmax: 1.39e-4 Dmax; (ii) the rounding of the face
//       coordinate b = gmin + i * cell itself, <= u (|gmin| + 2 extent): the builder keeps every axis' extent at >= 10^-3 of its
//       coordinates' magnitude, and extent <= Dmax / 1.25, so <= 4.8e-5 Dmax.  Together 1.87e-4 Dmax per axis, 3.24e-4 Dmax as a distance
//       (sqrt(3)) < dgrid = 4.88e-4 Dmax.  (Round 3 had dgrid = 2^-12 Dmax and counted term (i) only, and that at 2^-13 Dmax -- it is
//       1.14 x that; the CPU harness never failed, but its scenes sat near the origin, where (ii) vanishes.  Round 4 doubles dgrid and
//       adds scenes 300 ... 600 extents away from the origin and 128-cell-long tables to tests/sanitize/grid_main.cpp.)  For every true
//       parameter t the walk is, at its computed time t, in a cell whose slab contains the true point up to that error in every
//       axis -- so a cell within dgrid of the reported point p_j, in which j is listed, has been
//       visited once the computed exit time of the current cell is >= t_j.  The walk stops when that exit time reaches the current
//       nearest t, or when it steps onto the one-cell border of sentinels around the table (no reported point lies outside the box,
//       see (2)); the start cell is clamped into the table, which only adds cells.  Origins outside the box need no special case.
//       Beyond t_ok nothing of this holds (a reported point may lie outside the box), so the walk's answer stands only if its
//       nearest t is <= t_ok -- then every root below it is within the valid range and was covered; a walk that ends with a
//       nearest t > t_ok (a miss included, also when it left the table) hands its ray to the exhaustive loop.
//   (4) everything else -- rays that fail the test of (1), spheres more than 16 x the median radius (walls, lights: tested for
//       every ray, like the hierarchy's always-list) -- goes through the exhaustive loop / is tested unconditionally.
// Extra tests can never change the answer (every test is the reference's arithmetic on a sphere of the table), so the only
// obligation is the one (1)-(3) discharge: every sphere whose key beats or ties the final answer has been tested.
#ifndef SPT_GRID_H
#define SPT_GRID_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SPT_HD __host__ __device__ __forceinline__
#else
#define SPT_HD inline
#endif

namespace spt {

constexpr uint32_t kGridBorder = 0xFFFFFFFFu;      // header of a border cell: the ray has left the table
constexpr uint32_t kGridCountBits = 13;            // header = first reference << 13 | number of references (<= 8191)
constexpr int32_t kGridMaxDim = 128;               // cells per axis
constexpr uint32_t kGridAlways = 1024;             // outsized spheres (more than 16 x the median radius) tested for every ray: at most this many

// Everything the traversal needs besides the tables; plain data, passed by value to the kernel.
struct GridParams {
    float gmin[3], cell[3], inv_cell[3], gmax[3];  // the table covers [gmin, gmax], gmax = gmin + dim * cell
    float dfar2_max;                               // (1): squared farthest-corner distance a ray origin may have (already shrunk by 2^-20)
    float eta_max;                                 // (1): | fl(d.d) - 1 | up to which a walk is valid at any parameter
    float tok_scale;                               // (1): t_ok = tok_scale / sqrt(1.25 | fl(d.d) - 1 |) beyond that
    int32_t dim[3];                                // interior cells per axis
    int32_t stride_y, stride_z;                    // the table has a one-cell border: dim[0] + 2 and (dim[0] + 2) * (dim[1] + 2)
    uint32_t ncells, nrefs, nalways, n;            // table sizes; n = spheres in the scene
};

struct GridWalk {
    float tx, ty, tz;                              // ray parameter at which the walk leaves the current cell along x / y / z
    float dtx, dty, dtz;                           // parameter per cell
    int32_t sx, sy, sz;                            // linear-index step per axis, sign included (+-1, +-stride_y, +-stride_z)
    uint32_t ci;                                   // linear index of the current cell in the bordered table
};

SPT_HD float grid_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);               // 1 ulp; the bound of (3) has room for it
#else
    return 1.0f / x;
#endif
}

// (1): may this ray use the grid, and up to which parameter is its walk valid?  NaN / inf in o or d fail the comparisons and
// take the exhaustive loop.
SPT_HD bool grid_ray_ok(const GridParams& G, float ox, float oy, float oz, float dx, float dy, float dz, float& t_ok)
{
    const float ax = __builtin_fmaxf(__builtin_fabsf(ox - G.gmin[0]), __builtin_fabsf(ox - G.gmax[0]));
    const float ay = __builtin_fmaxf(__builtin_fabsf(oy - G.gmin[1]), __builtin_fabsf(oy - G.gmax[1]));
    const float az = __builtin_fmaxf(__builtin_fabsf(oz - G.gmin[2]), __builtin_fabsf(oz - G.gmax[2]));
    const float far2 = ax * ax + ay * ay + az * az;
    const float eta = __builtin_fabsf((dx * dx + dy * dy + dz * dz) - 1.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    const float rs = __builtin_amdgcn_rsqf(1.25f * eta);       // 1 ulp; the 0.99 in tok_scale has room for it
#else
    const float rs = 1.0f / __builtin_sqrtf(1.25f * eta);
#endif
    t_ok = eta <= G.eta_max ? __builtin_inff() : G.tok_scale * rs;
    return (far2 <= G.dfar2_max) & (eta < 0.25f);               // (eta < 1/4: a direction that is no direction at all; also refuses NaN)
}

// One axis of a walk's constants: 1 / d, whether the ray moves along the axis at all, parameter per cell, direction of the index step.
// (Also what a walker lane of spt_gpool.hip rebuilds from the direction when it takes a begun walk over: a walk's state in LDS is
// {exit parameters, cell index} only.)
SPT_HD void grid_axis_rate(float d, float cell, float& iv, bool& moving, float& dt, bool& neg)
{
    iv = grid_rcp(d);
    moving = __builtin_fabsf(d) >= 0x1p-60f;                      // else the ray never crosses a face of this axis (and iv may be inf)
    dt = moving ? cell * __builtin_fabsf(iv) : 0.0f;
    neg = !(d > 0.0f);
}

// One axis of the start of a walk: cell index (clamped into the table), exit parameter, parameter per cell, index step.
SPT_HD void grid_axis_begin(float o, float d, float gmin, float cell, float inv_cell, int32_t dim,
                            int32_t& idx, float& t, float& dt, bool& neg)
{
    const float f = (o - gmin) * inv_cell;
    const int32_t i = (int32_t)__builtin_fminf(__builtin_fmaxf(f, 0.0f), (float)(dim - 1));   // clamped into the table (NaN -> 0)
    float iv;
    bool moving;
    grid_axis_rate(d, cell, iv, moving, dt, neg);
    const float b = gmin + (float)(i + (neg ? 0 : 1)) * cell;     // the face the ray leaves the cell through
    t = moving ? (b - o) * iv : __builtin_inff();
    idx = i;
}

SPT_HD void grid_walk_begin(const GridParams& G, float ox, float oy, float oz, float dx, float dy, float dz, GridWalk& w)
{
    int32_t ix, iy, iz;
    bool nx, ny, nz;
    grid_axis_begin(ox, dx, G.gmin[0], G.cell[0], G.inv_cell[0], G.dim[0], ix, w.tx, w.dtx, nx);
    grid_axis_begin(oy, dy, G.gmin[1], G.cell[1], G.inv_cell[1], G.dim[1], iy, w.ty, w.dty, ny);
    grid_axis_begin(oz, dz, G.gmin[2], G.cell[2], G.inv_cell[2], G.dim[2], iz, w.tz, w.dtz, nz);
    w.sx = nx ? -1 : 1; w.sy = ny ? -G.stride_y : G.stride_y; w.sz = nz ? -G.stride_z : G.stride_z;
    w.ci = (uint32_t)((ix + 1) + G.stride_y * (iy + 1) + G.stride_z * (iz + 1));
}

// Parameter at which the walk leaves its current cell (NaN never appears: the exit parameters are finite or +inf).
SPT_HD float grid_walk_exit(const GridWalk& w) { return __builtin_fminf(w.tx, __builtin_fminf(w.ty, w.tz)); }

// Steps into the next cell through the face reached at m = grid_walk_exit(w).
// (The kernel keeps the fields of a GridWalk in separate registers: selecting between sx / sy / sz through the struct would make
// the compiler index it in scratch memory.)
SPT_HD void grid_walk_step(float& tx, float& ty, float& tz, float dtx, float dty, float dtz, int32_t sx, int32_t sy, int32_t sz, uint32_t& ci, float m)
{
    const bool isx = tx == m;
    const bool isy = !isx & (ty == m);
    const bool isz = !isx & !isy;
    tx += isx ? dtx : 0.0f;
    ty += isy ? dty : 0.0f;
    tz += isz ? dtz : 0.0f;
    int32_t st = isy ? sy : sz;
    st = isx ? sx : st;
    ci += (uint32_t)st;
}

}  // namespace spt

#if !defined(SPT_GRID_DEVICE_ONLY)      // host builder (spt_grid.hip, the kernel's translation unit, leaves it out)
#include <string>
#include <vector>

namespace spt {

struct SphereGrid {
    GridParams P{};
    std::vector<uint32_t> cells;      // (dim + 2)^3 headers; kGridBorder on the border
    std::vector<uint16_t> refs;       // sphere indices, ascending inside a cell
    std::vector<uint32_t> always;     // ascending indices of the spheres tested for every ray
    double dmax = 0.0;                // Dmax of (1)
    uint32_t max_cell = 0;            // most references in one cell
    std::vector<float> reach;         // R_j of (2) per sphere (0 for the always-tested ones); kept for validation
    bool usable = false;              // false: the scene does not fit (reason in why); the caller keeps another kernel
    std::string why;
    size_t lds_bytes() const { return cells.size() * 4 + ((refs.size() + 1) / 2) * 4 + always.size() * 4; }
};

// geom[i] = {centre, r*r}, radius[i] = r.  cells_per_sphere: resolution (interior cells ~ that many times the in-grid spheres);
// lds_budget: bytes available for cells + references + the always-list (the kernel keeps them in LDS).  Throws on non-finite input.
void build_sphere_grid(const float4* geom, const float* radius, uint32_t n, double cells_per_sphere, size_t lds_budget, SphereGrid& out);
// Structural check used by the CPU tests: every sphere is listed in every cell its cube of (2) meets (clamped into the table),
// references ascending and in range, border intact, every centre inside the box.
bool validate_sphere_grid(const float4* geom, const float* radius, uint32_t n, const SphereGrid& g, std::string& why);

}  // namespace spt
#endif
#endif
