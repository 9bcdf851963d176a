// spt_grid.cpp -- host builder of the uniform grid over a sphere table (see spt_grid.h for the layout and the contract).
#include "spt_grid.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

namespace spt {
namespace {

inline float round_down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f; }
inline float round_up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f; }

// R_j of spt_grid.h (2) for a sphere of radius r under the ray test with farthest-corner distance dmax
double reach_of(double r, double dmax)
{
    const double E = std::ldexp(dmax * dmax + r * r, -17) + std::ldexp(2.25 * dmax * dmax, -19);
    return std::sqrt(r * r + E) * (1.0 + 1e-12) + std::ldexp(dmax, -11);      // + dgrid
}

// cells an interval [lo, hi] meets along one axis, clamped into the table (the device clamps its start cell the same way)
inline void cell_range(double lo, double hi, double gmin, double cell, int32_t dim, int32_t& i0, int32_t& i1)
{
    const double a = std::floor((lo - gmin) / cell), b = std::floor((hi - gmin) / cell);
    i0 = (int32_t)std::min(std::max(a, 0.0), (double)(dim - 1));
    i1 = (int32_t)std::min(std::max(b, 0.0), (double)(dim - 1));
}

// does the ball of radius R around c meet cell (x, y, z) of the table?  (2) of spt_grid.h: squared distance from c to the cell's box in
// double, with a relative reserve far above the double rounding of the box edges
inline bool ball_meets_cell(const double c[3], double R, const GridParams& P, int32_t x, int32_t y, int32_t z)
{
    const int32_t idx[3] = {x, y, z};
    double d2 = 0.0;
    for (int a = 0; a < 3; ++a) {
        const double lo = (double)P.gmin[a] + (double)idx[a] * (double)P.cell[a], hi = lo + (double)P.cell[a];
        const double d = std::max(0.0, std::max(lo - c[a], c[a] - hi));
        d2 += d * d;
    }
    return d2 <= R * R * (1.0 + 1e-9);
}

}  // namespace

void build_sphere_grid(const float4* geom, const float* radius, uint32_t n, double cells_per_sphere, size_t lds_budget, SphereGrid& out)
{
    out = SphereGrid{};
    for (uint32_t i = 0; i < n; ++i)
        if (!(std::isfinite(geom[i].x) && std::isfinite(geom[i].y) && std::isfinite(geom[i].z) && std::isfinite(radius[i])))
            throw std::runtime_error("sphere grid: a sphere has non-finite centre or radius");
    if (n > 0xFFFFu) { out.why = "more than 65535 spheres"; return; }
    // spheres far larger than the rest would be listed in every cell: they are tested for every ray instead
    std::vector<float> rs(n);
    for (uint32_t i = 0; i < n; ++i) rs[i] = std::fabs(radius[i]);
    std::vector<uint32_t> huge;
    if (n > 0) {
        std::vector<float> sorted(rs);
        std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
        const float cut = 16.0f * sorted[n / 2];
        for (uint32_t i = 0; i < n; ++i) if (rs[i] > cut) huge.push_back(i);
        if (huge.size() > kGridAlways) {
            std::sort(huge.begin(), huge.end(), [&](uint32_t x, uint32_t y) { return rs[x] > rs[y] || (rs[x] == rs[y] && x < y); });
            huge.resize(kGridAlways);
        }
        std::sort(huge.begin(), huge.end());
    }
    out.always = huge;
    std::vector<uint32_t> ids;
    {
        size_t h = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (h < huge.size() && huge[h] == i) { ++h; continue; }
            ids.push_back(i);
        }
    }
    out.reach.assign(n, 0.0f);
    GridParams& P = out.P;
    P.n = n; P.nalways = (uint32_t)huge.size();
    P.eta_max = 0x1p-19f - 0x1p-22f;

    // extent of the spheres themselves -> Dmax -> reaches -> the box (union of the cubes c +- R)
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    double dmax = 1.0;
    if (!ids.empty()) {
        double slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i : ids) {
            const double c[3] = {geom[i].x, geom[i].y, geom[i].z};
            for (int a = 0; a < 3; ++a) { slo[a] = std::min(slo[a], c[a] - rs[i]); shi[a] = std::max(shi[a], c[a] + rs[i]); }
        }
        const double diag0 = std::sqrt((shi[0] - slo[0]) * (shi[0] - slo[0]) + (shi[1] - slo[1]) * (shi[1] - slo[1]) + (shi[2] - slo[2]) * (shi[2] - slo[2]));
        dmax = 1.3 * diag0;
        auto cubes = [&](double d) {                      // union of the cubes c +- R(d); returns its diagonal
            for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
            for (uint32_t i : ids) {
                const double R = reach_of(rs[i], d);
                const double c[3] = {geom[i].x, geom[i].y, geom[i].z};
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a] - R); hi[a] = std::max(hi[a], c[a] + R); }
            }
            // (an axis is at least 10^-3 of its coordinates' magnitude wide, see below: part of the box the ray test must admit --
            // a thin table far from the origin, tests/sanitize/grid_main.cpp kind 8, had its box widened after Dmax was fixed)
            for (int a = 0; a < 3; ++a) { const double mag = std::max(1.0, std::fabs(lo[a]) + std::fabs(hi[a])); hi[a] = std::max(hi[a], lo[a] + 1e-3 * mag); }
            return std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        };
        for (int pass = 0; pass < 8; ++pass) {
            const double diag = cubes(dmax);
            if (diag <= 0.9 * dmax) break;                // origins anywhere in the box (and a little outside) pass the ray test
            dmax = 1.25 * diag;
        }
        (void)cubes(dmax);                                // the box belongs to the Dmax that is used (also when the loop ran out)
        if (!std::isfinite(dmax) || !(dmax < 1e15)) { out.why = "extent overflows"; return; }
    }
    out.dmax = dmax;
    for (uint32_t i : ids) out.reach[i] = round_up(reach_of(rs[i], dmax));
    // float box, grown by 1e-6 of its size and coordinates so that the binary32 roundings of gmin, cell and gmax stay outside the cubes
    double ext[3];
    for (int a = 0; a < 3; ++a) {
        const double mag = std::max(1.0, std::fabs(lo[a]) + std::fabs(hi[a]));
        const double e = std::max(hi[a] - lo[a], 1e-3 * mag);
        const double m = 1e-6 * std::max(e, mag);
        lo[a] -= m; ext[a] = e + 3.0 * m;
    }

    // resolution: cubic cells, about cells_per_sphere per in-grid sphere, shrunk until the tables fit the LDS budget
    double target = std::max(8.0, cells_per_sphere * (double)std::max<size_t>(ids.size(), 1));
    std::vector<uint32_t> count, first;
    for (;;) {
        const double vol = ext[0] * ext[1] * ext[2];
        const double s = std::cbrt(vol / target);
        int32_t dim[3];
        for (int a = 0; a < 3; ++a) dim[a] = (int32_t)std::min<double>(kGridMaxDim, std::max(1.0, std::ceil(ext[a] / s)));
        const size_t ncells = (size_t)(dim[0] + 2) * (dim[1] + 2) * (dim[2] + 2);
        for (int a = 0; a < 3; ++a) {
            P.dim[a] = dim[a];
            P.gmin[a] = round_down(lo[a]);
            P.cell[a] = round_up(ext[a] / dim[a] * (1.0 + 1e-6));
            P.inv_cell[a] = (float)(1.0 / (double)P.cell[a]);
            P.gmax[a] = round_up((double)P.gmin[a] + (double)dim[a] * (double)P.cell[a]);
        }
        P.stride_y = dim[0] + 2; P.stride_z = (dim[0] + 2) * (dim[1] + 2);
        bool fits = ncells * 4 <= lds_budget;
        size_t nrefs = 0;
        if (fits) {
            count.assign(ncells, 0u);
            for (uint32_t i : ids) {
                const double c[3] = {geom[i].x, geom[i].y, geom[i].z};
                int32_t r0[3], r1[3];
                for (int a = 0; a < 3; ++a) cell_range(c[a] - out.reach[i], c[a] + out.reach[i], P.gmin[a], P.cell[a], dim[a], r0[a], r1[a]);
                for (int32_t z = r0[2]; z <= r1[2]; ++z)
                    for (int32_t y = r0[1]; y <= r1[1]; ++y)
                        for (int32_t x = r0[0]; x <= r1[0]; ++x)
                            if (ball_meets_cell(c, out.reach[i], P, x, y, z)) { ++count[(size_t)(x + 1) + (size_t)P.stride_y * (y + 1) + (size_t)P.stride_z * (z + 1)]; ++nrefs; }
                if (nrefs * 2 > lds_budget) break;
            }
            const uint32_t cmax = count.empty() ? 0u : *std::max_element(count.begin(), count.end());
            fits = ncells * 4 + ((nrefs + 1) / 2) * 4 + huge.size() * 4 <= lds_budget && nrefs < (1u << (32 - kGridCountBits)) - 1u && cmax < (1u << kGridCountBits);
            out.max_cell = cmax;
        }
        if (fits) {
            first.assign(ncells + 1, 0u);
            for (size_t k = 0; k < ncells; ++k) first[k + 1] = first[k] + count[k];
            out.refs.assign(nrefs, 0);
            std::vector<uint32_t> fill(first.begin(), first.end() - 1);
            for (uint32_t i : ids) {                       // ascending sphere index => ascending inside every cell
                const double c[3] = {geom[i].x, geom[i].y, geom[i].z};
                int32_t r0[3], r1[3];
                for (int a = 0; a < 3; ++a) cell_range(c[a] - out.reach[i], c[a] + out.reach[i], P.gmin[a], P.cell[a], dim[a], r0[a], r1[a]);
                for (int32_t z = r0[2]; z <= r1[2]; ++z)
                    for (int32_t y = r0[1]; y <= r1[1]; ++y)
                        for (int32_t x = r0[0]; x <= r1[0]; ++x)
                            if (ball_meets_cell(c, out.reach[i], P, x, y, z))
                                out.refs[fill[(size_t)(x + 1) + (size_t)P.stride_y * (y + 1) + (size_t)P.stride_z * (z + 1)]++] = (uint16_t)i;
            }
            out.cells.assign(ncells, kGridBorder);
            for (int32_t z = 0; z < dim[2]; ++z)
                for (int32_t y = 0; y < dim[1]; ++y)
                    for (int32_t x = 0; x < dim[0]; ++x) {
                        const size_t k = (size_t)(x + 1) + (size_t)P.stride_y * (y + 1) + (size_t)P.stride_z * (z + 1);
                        out.cells[k] = (first[k] << kGridCountBits) | count[k];
                    }
            P.ncells = (uint32_t)ncells; P.nrefs = (uint32_t)nrefs;
            break;
        }
        if (target <= 8.0) { out.why = "the tables do not fit the LDS budget"; return; }
        target = std::max(8.0, target * 0.7);
    }
    // ray test (1): farthest-corner distance against Dmax, with room for the binary32 evaluation of the test itself
    const double d2 = dmax * dmax * (1.0 - std::ldexp(1.0, -18));
    P.dfar2_max = round_down(d2);
    P.tok_scale = round_down(0.99 * 1.5 * dmax * std::sqrt(std::ldexp(1.0, -19)));
    out.usable = true;
}

bool validate_sphere_grid(const float4* geom, const float* radius, uint32_t n, const SphereGrid& g, std::string& why)
{
    const GridParams& P = g.P;
    if (!g.usable) { why = "grid not usable: " + g.why; return false; }
    const size_t ncells = (size_t)(P.dim[0] + 2) * (P.dim[1] + 2) * (P.dim[2] + 2);
    if (g.cells.size() != ncells || P.ncells != ncells || P.nrefs != g.refs.size() || P.n != n) { why = "table sizes"; return false; }
    if (P.stride_y != P.dim[0] + 2 || P.stride_z != (P.dim[0] + 2) * (P.dim[1] + 2)) { why = "strides"; return false; }
    std::vector<char> is_always(n, 0);
    for (size_t k = 0; k < g.always.size(); ++k) {
        if (g.always[k] >= n || (k && g.always[k] <= g.always[k - 1])) { why = "always-list not ascending / out of range"; return false; }
        is_always[g.always[k]] = 1;
    }
    if (g.always.size() > kGridAlways || P.nalways != g.always.size()) { why = "always-list size"; return false; }
    for (int32_t z = -1; z <= P.dim[2]; ++z)
        for (int32_t y = -1; y <= P.dim[1]; ++y)
            for (int32_t x = -1; x <= P.dim[0]; ++x) {
                const bool border = x < 0 || y < 0 || z < 0 || x == P.dim[0] || y == P.dim[1] || z == P.dim[2];
                const uint32_t h = g.cells[(size_t)(x + 1) + (size_t)P.stride_y * (y + 1) + (size_t)P.stride_z * (z + 1)];
                if (border != (h == kGridBorder)) { why = "border"; return false; }
                if (border) continue;
                const uint32_t f = h >> kGridCountBits, c = h & ((1u << kGridCountBits) - 1u);
                if ((size_t)f + c > g.refs.size()) { why = "reference range"; return false; }
                for (uint32_t k = 0; k < c; ++k) {
                    if (g.refs[f + k] >= n || is_always[g.refs[f + k]]) { why = "reference out of range / to an always-tested sphere"; return false; }
                    if (k && g.refs[f + k] <= g.refs[f + k - 1]) { why = "references not ascending"; return false; }
                }
            }
    for (uint32_t i = 0; i < n; ++i) {
        if (is_always[i]) continue;
        const double c[3] = {geom[i].x, geom[i].y, geom[i].z};
        const double r = std::fabs((double)radius[i]);
        const double need = reach_of(r, g.dmax);
        if (!((double)g.reach[i] >= need)) { why = "reach of sphere " + std::to_string(i) + " below the bound"; return false; }
        int32_t r0[3], r1[3];
        for (int a = 0; a < 3; ++a) {
            if (!(c[a] - need >= (double)P.gmin[a] && c[a] + need <= (double)P.gmax[a])) { why = "cube of sphere " + std::to_string(i) + " outside the box"; return false; }
            cell_range(c[a] - need, c[a] + need, P.gmin[a], P.cell[a], P.dim[a], r0[a], r1[a]);
        }
        for (int32_t z = r0[2]; z <= r1[2]; ++z)
            for (int32_t y = r0[1]; y <= r1[1]; ++y)
                for (int32_t x = r0[0]; x <= r1[0]; ++x) {
                    if (!ball_meets_cell(c, need, P, x, y, z)) continue;
                    const uint32_t h = g.cells[(size_t)(x + 1) + (size_t)P.stride_y * (y + 1) + (size_t)P.stride_z * (z + 1)];
                    const uint32_t f = h >> kGridCountBits, cn = h & ((1u << kGridCountBits) - 1u);
                    if (!std::binary_search(g.refs.begin() + f, g.refs.begin() + f + cn, (uint16_t)i)) { why = "sphere " + std::to_string(i) + " missing from a cell of its ball"; return false; }
                }
    }
    // the ray test must admit every origin inside the box
    const double diag = std::sqrt(std::pow((double)P.gmax[0] - P.gmin[0], 2) + std::pow((double)P.gmax[1] - P.gmin[1], 2) + std::pow((double)P.gmax[2] - P.gmin[2], 2));
    if (!g.refs.empty() && !((double)P.dfar2_max >= diag * diag)) { why = "ray test rejects origins inside the box"; return false; }
    if (!((double)P.dfar2_max <= g.dmax * g.dmax)) { why = "ray test admits origins beyond Dmax"; return false; }
    return true;
}

}  // namespace spt
