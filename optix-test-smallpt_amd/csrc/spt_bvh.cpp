// spt_bvh.cpp -- binned-SAH builder of the optional triangle hierarchy (see spt_bvh.h for the layout and the contract).
#include "spt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <stdexcept>

namespace spt {
namespace {

struct Box {
    float mn[3], mx[3];
    void clear() { for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -std::numeric_limits<float>::infinity(); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
    double half_area() const
    {
        const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        return dx < 0 ? 0.0 : dx * dy + dy * dz + dz * dx;
    }
};

inline float round_down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -std::numeric_limits<float>::infinity()) : f; }
inline float round_up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, std::numeric_limits<float>::infinity()) : f; }

// Padded box of one triangle record: the vertices are v0, v0 + e1, v0 + e2 (e1, e2 carry one rounding of the reference's
// v1 - v0, v2 - v0, far inside the padding).
Box padded_box(const float4* r)
{
    const double v[3][3] = {{r[0].x, r[0].y, r[0].z},
                            {(double)r[0].x + r[1].x, (double)r[0].y + r[1].y, (double)r[0].z + r[1].z},
                            {(double)r[0].x + r[2].x, (double)r[0].y + r[2].y, (double)r[0].z + r[2].z}};
    double edge2 = 0.0, big = 0.0;
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c)
            if (!std::isfinite(v[k][c])) throw std::runtime_error("spt_set_mesh_accel: a triangle has non-finite vertices");
    for (int k = 0; k < 3; ++k) {
        const double* a = v[k];
        const double* b = v[(k + 1) % 3];
        edge2 = std::max(edge2, (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
        for (int c = 0; c < 3; ++c) big = std::max(big, std::fabs(a[c]));
    }
    if (!std::isfinite(edge2)) throw std::runtime_error("spt_set_mesh_accel: a triangle's extent overflows");
    const double pad = 0.25 * std::sqrt(edge2) + 1e-4 * big + 1e-30;
    Box b;
    for (int c = 0; c < 3; ++c) {
        b.mn[c] = round_down(std::min({v[0][c], v[1][c], v[2][c]}) - pad);
        b.mx[c] = round_up(std::max({v[0][c], v[1][c], v[2][c]}) + pad);
    }
    return b;
}

// internal levels a subtree of `count` triangles needs when split at the median: leaf references end up that much deeper
inline uint32_t levels_needed(uint32_t count, uint32_t leaf_max)
{
    uint32_t leaves = (count + leaf_max - 1) / leaf_max, l = 0;
    while ((1u << l) < leaves) ++l;
    return l;
}

struct Builder {
    const float4* recs;              // rec_f4 float4 per primitive, indexed by GLOBAL id
    uint32_t rec_f4 = 3;
    uint32_t leaf_max = kBvhLeafTris; // primitives per leaf (<= 15: four bits of the leaf reference)
    std::vector<uint32_t> ids;       // global id of every primitive handed to the builder (empty = identity)
    std::vector<Box> box;            // per primitive
    std::vector<float> cen;          // 3 per primitive
    std::vector<uint32_t> order;     // permutation being partitioned
    Bvh* out;
    // triangle hierarchies only (spt_tribvh.h (1)): unit normal, g and longest edge per primitive -> a cone per child in out->cones
    std::vector<double> nrm, gq, el;

    struct Cone { float a[3]; float kappa, iq, ee; };
    Cone range_cone(uint32_t lo, uint32_t hi) const
    {
        Cone c{{1.f, 0.f, 0.f}, 2.0f, 1.0f, 0.0f};
        if (lo == hi) return c;
        const double* ref = &nrm[3 * (size_t)order[lo]];
        double mean[3] = {0, 0, 0}, gmax = 1.0, emax = 0.0;
        for (uint32_t i = lo; i < hi; ++i) {
            const double* q = &nrm[3 * (size_t)order[i]];
            const double sgn = q[0] * ref[0] + q[1] * ref[1] + q[2] * ref[2] < 0.0 ? -1.0 : 1.0;      // either orientation of a normal
            for (int k = 0; k < 3; ++k) mean[k] += sgn * q[k];
            gmax = std::max(gmax, gq[order[i]]); emax = std::max(emax, el[order[i]]);
        }
        const double len = std::sqrt(mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]);
        if (len > 1e-6 * (hi - lo)) for (int k = 0; k < 3; ++k) c.a[k] = (float)(mean[k] / len);
        const double al = std::sqrt((double)c.a[0] * c.a[0] + (double)c.a[1] * c.a[1] + (double)c.a[2] * c.a[2]);   // the stored axis is a unit vector up to 1e-7
        double kap = 0.0;
        for (uint32_t i = lo; i < hi; ++i) {
            const double* q = &nrm[3 * (size_t)order[i]];
            double dm = 0.0, dp = 0.0;
            for (int k = 0; k < 3; ++k) { dm += (q[k] - c.a[k]) * (q[k] - c.a[k]); dp += (q[k] + c.a[k]) * (q[k] + c.a[k]); }
            kap = std::max(kap, std::sqrt(std::min(dm, dp)));
        }
        c.kappa = round_up(kap + std::fabs(al - 1.0) + 1e-6);
        c.iq = round_down(1.0 / gmax);
        c.ee = round_up(1.016 * emax);
        return c;
    }
    void store_cones(uint32_t node, const Cone& l, const Cone& r)
    {
        if (nrm.empty()) return;
        if (out->cones.size() < 3 * (size_t)(node + 1)) out->cones.resize(3 * (size_t)(node + 1));
        float4* cn = &out->cones[3 * (size_t)node];
        cn[0] = make_float4(l.a[0], l.a[1], l.a[2], l.kappa);
        cn[1] = make_float4(r.a[0], r.a[1], r.a[2], r.kappa);
        cn[2] = make_float4(l.iq, l.ee, r.iq, r.ee);
    }

    int32_t leaf_ref(uint32_t lo, uint32_t hi)
    {
        const uint32_t first = (uint32_t)out->index.size();
        std::sort(order.begin() + lo, order.begin() + hi);               // ascending global index inside a leaf (tidy; ties are broken by index anyway)
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t g = ids.empty() ? order[i] : ids[order[i]];
            out->index.push_back(g);
            for (uint32_t k = 0; k < rec_f4; ++k) out->tris.push_back(recs[rec_f4 * (size_t)g + k]);
        }
        ++out->leaves;
        return ~(int32_t)((first << 4) | (hi - lo));
    }

    // whole hierarchy over the primitives whose boxes / centroids are set; the root is always node 0
    void run()
    {
        const uint32_t n = (uint32_t)box.size();
        if (n >= (1u << 27)) throw std::runtime_error("hierarchy: too many primitives for the leaf encoding");
        order.resize(n);
        std::iota(order.begin(), order.end(), 0u);
        out->index.reserve(n); out->tris.reserve(rec_f4 * (size_t)n);
        if (n <= leaf_max) {
            // one leaf with everything, one empty leaf behind an inverted box
            out->nodes.resize(4);
            Box l = range_box(0, n), r; r.clear();
            if (n == 0) l.clear();
            const int32_t lr = leaf_ref(0, n), rr = ~(int32_t)0;
            out->nodes[0] = make_float4(l.mn[0], l.mn[1], l.mn[2], l.mx[0]);
            out->nodes[1] = make_float4(l.mx[1], l.mx[2], r.mn[0], r.mn[1]);
            out->nodes[2] = make_float4(r.mn[2], r.mx[0], r.mx[1], r.mx[2]);
            float4 refs = make_float4(0.f, 0.f, 0.f, 0.f);
            std::memcpy(&refs.x, &lr, 4); std::memcpy(&refs.y, &rr, 4);
            out->nodes[3] = refs;
            out->depth = 1;
            if (!nrm.empty()) store_cones(0, range_cone(0, n), range_cone(0, 0));
        } else {
            const int32_t root = build(0, n, 0);
            if (root != 0) throw std::runtime_error("hierarchy: internal error (root is not node 0)");
        }
        if (out->tris.empty()) out->tris.resize(rec_f4, make_float4(0.f, 0.f, 0.f, 0.f));   // never read; keeps the device buffers non-empty
        if (out->index.empty()) out->index.push_back(0u);
    }

    Box range_box(uint32_t lo, uint32_t hi) const
    {
        Box b; b.clear();
        for (uint32_t i = lo; i < hi; ++i) b.grow(box[order[i]]);
        return b;
    }

    // returns the split position in (lo, hi); children must fit into `levels_left` further internal levels each
    uint32_t split(uint32_t lo, uint32_t hi, uint32_t levels_left)
    {
        const uint32_t n = hi - lo;
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) { const float c = cen[3 * (size_t)order[i] + a]; cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c); }
        constexpr int kBins = 16;
        double best = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = 0;
        for (int a = 0; a < 3; ++a) {
            const double ext = (double)cmx[a] - cmn[a];
            if (!(ext > 0)) continue;
            Box bb[kBins]; uint32_t cnt[kBins] = {};
            for (auto& b : bb) b.clear();
            const double scale = kBins / ext;
            for (uint32_t i = lo; i < hi; ++i) {
                const uint32_t g = order[i];
                const int k = std::min(kBins - 1, (int)(((double)cen[3 * (size_t)g + a] - cmn[a]) * scale));
                bb[k].grow(box[g]); ++cnt[k];
            }
            Box acc; acc.clear();
            double right_area[kBins]; uint32_t right_cnt[kBins];
            uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); c += cnt[k]; right_area[k] = acc.half_area(); right_cnt[k] = c; }
            acc.clear(); c = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                acc.grow(bb[k]); c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const double cost = acc.half_area() * c + right_area[k + 1] * right_cnt[k + 1];
                if (cost < best) { best = cost; best_axis = a; best_bin = k; }
            }
        }
        if (best_axis >= 0) {
            const int a = best_axis;
            const double scale = kBins / ((double)cmx[a] - cmn[a]);
            auto mid = std::partition(order.begin() + lo, order.begin() + hi, [&](uint32_t g) {
                return std::min(kBins - 1, (int)(((double)cen[3 * (size_t)g + a] - cmn[a]) * scale)) <= best_bin;
            });
            const uint32_t m = (uint32_t)(mid - order.begin());
            if (m > lo && m < hi && levels_needed(m - lo, leaf_max) <= levels_left && levels_needed(hi - m, leaf_max) <= levels_left) return m;
        }
        // median along the widest centroid axis (or by index when all centroids coincide): depth stays logarithmic
        int a = 0;
        for (int k = 1; k < 3; ++k) if (cmx[k] - cmn[k] > cmx[a] - cmn[a]) a = k;
        const uint32_t m = lo + (n + 1) / 2;
        std::nth_element(order.begin() + lo, order.begin() + m, order.begin() + hi, [&](uint32_t x, uint32_t y) {
            const float cx = cen[3 * (size_t)x + a], cy = cen[3 * (size_t)y + a];
            return cx < cy || (cx == cy && x < y);
        });
        return m;
    }

    // builds the subtree over order[lo, hi) whose reference sits at depth `depth`; returns the reference
    int32_t build(uint32_t lo, uint32_t hi, uint32_t depth)
    {
        out->depth = std::max(out->depth, depth);
        if (hi - lo <= leaf_max) return leaf_ref(lo, hi);
        if (depth >= kBvhMaxDepth) throw std::runtime_error("spt_set_mesh_accel: hierarchy depth bound violated");
        const uint32_t node = (uint32_t)(out->nodes.size() / 4);
        out->nodes.resize(out->nodes.size() + 4);
        const uint32_t m = split(lo, hi, kBvhMaxDepth - (depth + 1));
        const Box l = range_box(lo, m), r = range_box(m, hi);
        if (!nrm.empty()) store_cones(node, range_cone(lo, m), range_cone(m, hi));
        const int32_t lr = build(lo, m, depth + 1), rr = build(m, hi, depth + 1);
        float4* nd = &out->nodes[4 * (size_t)node];
        nd[0] = make_float4(l.mn[0], l.mn[1], l.mn[2], l.mx[0]);
        nd[1] = make_float4(l.mx[1], l.mx[2], r.mn[0], r.mn[1]);
        nd[2] = make_float4(r.mn[2], r.mx[0], r.mx[1], r.mx[2]);
        float4 refs = make_float4(0.f, 0.f, 0.f, 0.f);
        std::memcpy(&refs.x, &lr, 4); std::memcpy(&refs.y, &rr, 4);
        nd[3] = refs;
        return (int32_t)node;
    }
};

}  // namespace

// ---- triangle classes, the cone trees and the line table of spt_tribvh.h ---------------------------------------------------------------
namespace {

enum TriClass { kTriDead = 0, kTriRegular = 1, kTriThin = 2 };

struct TriGeom {
    double v0[3], e1[3], e2[3], n[3];     // n = e1 x e2 in double (exact products of floats, one rounding of 2^-53 in the difference)
    double l1, l2, nn, g;                 // |e1|, |e2|, |n|, g = |e1||e2| / |n| (inf for n = 0)
    TriClass cls;
};

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(const double* a, const double* b, double* c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }

TriGeom tri_geom(const float4* r)
{
    TriGeom t;
    t.v0[0] = r[0].x; t.v0[1] = r[0].y; t.v0[2] = r[0].z;
    t.e1[0] = r[1].x; t.e1[1] = r[1].y; t.e1[2] = r[1].z;
    t.e2[0] = r[2].x; t.e2[1] = r[2].y; t.e2[2] = r[2].z;
    cross3(t.e1, t.e2, t.n);
    t.l1 = std::sqrt(dot3(t.e1, t.e1)); t.l2 = std::sqrt(dot3(t.e2, t.e2)); t.nn = std::sqrt(dot3(t.n, t.n));
    const double E = t.l1 * t.l2;
    t.g = t.nn > 0.0 ? E / t.nn : std::numeric_limits<double>::infinity();
    t.cls = !(E > 0.0) ? kTriDead : (t.g <= kTriThinG ? kTriRegular : kTriThin);
    return t;
}

// One triangle as the cone trees see it: a unit direction (plane normal or long-edge direction; either sign), the point v0, its
// tolerance (tau of spt_tribvh.h (2) or a of (3)) and its longest edge.
struct ConeItem { double d[3], v0[3], tol, edge; uint32_t gid; };

ConeItem cone_item(const TriGeom& t, uint32_t gid)
{
    ConeItem c{};
    c.gid = gid;
    for (int a = 0; a < 3; ++a) c.v0[a] = t.v0[a];
    c.edge = std::max(t.l1, t.l2);
    if (t.cls == kTriRegular) {
        for (int a = 0; a < 3; ++a) c.d[a] = t.n[a] / t.nn;
        c.tol = kTriBand * t.g;
    } else {
        const bool first_longer = t.l1 >= t.l2;
        const double* eL = first_longer ? t.e1 : t.e2;
        const double lL = first_longer ? t.l1 : t.l2, lS = first_longer ? t.l2 : t.l1;
        for (int a = 0; a < 3; ++a) c.d[a] = eL[a] / lL;
        c.tol = lS * ((std::isfinite(t.g) ? 1.0 / t.g : 0.0) + 7.2 * 0x1p-24);
    }
    return c;
}

struct ConeChild { float a[3], kappa, p[3], s1, s2, s3; };   // planes: s1 = sigma, s2 = tau, s3 = te;  lines: s1 = lam

// solves the symmetric 3 x 3 system A x = b (A positive definite by construction: a regularisation sits on its diagonal)
inline bool solve3(const double A[3][3], const double b[3], double x[3])
{
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double inv[3][3] = {{c00, A[0][2] * A[2][1] - A[0][1] * A[2][2], A[0][1] * A[1][2] - A[0][2] * A[1][1]},
                              {c01, A[0][0] * A[2][2] - A[0][2] * A[2][0], A[0][2] * A[1][0] - A[0][0] * A[1][2]},
                              {c02, A[0][1] * A[2][0] - A[0][0] * A[2][1], A[0][0] * A[1][1] - A[0][1] * A[1][0]}};
    for (int i = 0; i < 3; ++i) x[i] = (inv[i][0] * b[0] + inv[i][1] * b[1] + inv[i][2] * b[2]) / det;
    return std::isfinite(x[0]) && std::isfinite(x[1]) && std::isfinite(x[2]);
}

// Tree over ConeItems, clustered by direction and position; PLANES = true: spt_tribvh.h (2), else (3).
struct ConeBuilder {
    bool planes = true;
    std::vector<ConeItem> items;
    std::vector<uint32_t> order;
    std::vector<float4>* nodes = nullptr;
    double dir_scale = 1.0;           // one unit of direction difference counts as this much distance when splitting (the scene's size)
    uint32_t depth = 0;
    uint32_t node_f4() const { return planes ? 6u : 5u; }

    ConeChild summary(uint32_t lo, uint32_t hi)
    {
        ConeChild c{};
        c.a[0] = 1.f; c.kappa = -1e30f;                                  // empty child: never entered
        if (lo == hi) return c;
        const uint32_t n = hi - lo;
        const double* ref = items[order[lo]].d;
        double mean[3] = {0, 0, 0}, cen[3] = {0, 0, 0};
        for (uint32_t i = lo; i < hi; ++i) {
            ConeItem& it = items[order[i]];
            if (dot3(it.d, ref) < 0.0) for (int k = 0; k < 3; ++k) it.d[k] = -it.d[k];      // same plane / line, the orientation next to the first one's
            for (int k = 0; k < 3; ++k) { mean[k] += it.d[k]; cen[k] += it.v0[k] / n; }
        }
        const double len = std::sqrt(dot3(mean, mean));
        if (len > 1e-9 * n) for (int k = 0; k < 3; ++k) c.a[k] = (float)(mean[k] / len);
        // reference point: where the cluster's planes (lines) come closest to meeting, pulled to the centroid of the v0 where they do not say
        double p[3] = {cen[0], cen[1], cen[2]};
        if (n > 1) {
            double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, b[3] = {0, 0, 0};
            for (uint32_t i = lo; i < hi; ++i) {
                const ConeItem& it = items[order[i]];
                const double rel[3] = {it.v0[0] - cen[0], it.v0[1] - cen[1], it.v0[2] - cen[2]};
                const double dr = dot3(it.d, rel);
                for (int r = 0; r < 3; ++r) {
                    for (int q = 0; q < 3; ++q) A[r][q] += planes ? it.d[r] * it.d[q] : ((r == q ? 1.0 : 0.0) - it.d[r] * it.d[q]);
                    b[r] += planes ? it.d[r] * dr : rel[r] - it.d[r] * dr;
                }
            }
            const double reg = 1e-4 * n;
            for (int r = 0; r < 3; ++r) A[r][r] += reg;
            double x[3];
            if (solve3(A, b, x)) for (int k = 0; k < 3; ++k) p[k] = cen[k] + x[k];
        }
        for (int k = 0; k < 3; ++k) c.p[k] = (float)p[k];
        const double al = std::sqrt((double)c.a[0] * c.a[0] + (double)c.a[1] * c.a[1] + (double)c.a[2] * c.a[2]);
        double kap = 0.0, s1 = 0.0, tmax = 0.0, rho = 0.0;
        for (uint32_t i = lo; i < hi; ++i) {
            const ConeItem& it = items[order[i]];
            double dm = 0.0, dp = 0.0;
            for (int k = 0; k < 3; ++k) { dm += (it.d[k] - c.a[k]) * (it.d[k] - c.a[k]); dp += (it.d[k] + c.a[k]) * (it.d[k] + c.a[k]); }
            kap = std::max(kap, std::sqrt(std::min(dm, dp)));
            const double rel[3] = {it.v0[0] - c.p[0], it.v0[1] - c.p[1], it.v0[2] - c.p[2]};
            const double r = std::sqrt(dot3(rel, rel));
            rho = std::max(rho, r);
            if (planes) {
                s1 = std::max(s1, std::fabs(dot3(it.d, rel)));
                tmax = std::max(tmax, it.tol);
            } else {
                double cr[3];
                cross3(it.d, rel, cr);
                s1 = std::max(s1, std::sqrt(dot3(cr, cr)) + it.tol + 8.1 * 0x1p-24 * r);
            }
        }
        c.kappa = round_up(kap + std::fabs(al - 1.0) + 1e-7);
        c.s1 = round_up(s1 * (1.0 + 1e-9) + 1e-30);
        if (planes) {
            double te = 0.0;
            for (uint32_t i = lo; i < hi; ++i) te = std::max(te, items[order[i]].tol * (rho + 2.0 * items[order[i]].edge) * 1.001);
            c.s2 = round_up(tmax);
            c.s3 = round_up(te);
        }
        return c;
    }

    void store(uint32_t node, const ConeChild& l, const ConeChild& r, int32_t lr, int32_t rr)
    {
        float4* nd = &(*nodes)[(size_t)node_f4() * node];
        float4 refs = make_float4(0.f, 0.f, 0.f, 0.f);
        std::memcpy(&refs.x, &lr, 4); std::memcpy(&refs.y, &rr, 4);
        nd[0] = make_float4(l.a[0], l.a[1], l.a[2], l.kappa);
        nd[1] = make_float4(l.p[0], l.p[1], l.p[2], l.s1);
        nd[2] = make_float4(r.a[0], r.a[1], r.a[2], r.kappa);
        nd[3] = make_float4(r.p[0], r.p[1], r.p[2], r.s1);
        if (planes) { nd[4] = make_float4(l.s2, l.s3, r.s2, r.s3); nd[5] = refs; }
        else nd[4] = refs;
    }

    // subtree over order[lo, hi), hi - lo >= 2; its reference sits at depth `d`
    int32_t build(uint32_t lo, uint32_t hi, uint32_t d)
    {
        if (d >= kBvhMaxDepth) throw std::runtime_error("spt_set_mesh_accel: cone tree depth bound violated");
        const uint32_t node = (uint32_t)(nodes->size() / node_f4());
        nodes->resize(nodes->size() + node_f4());
        (void)summary(lo, hi);                                            // aligns the orientations of the whole range to its first item
        double mn[6], mx[6];
        for (int k = 0; k < 6; ++k) { mn[k] = std::numeric_limits<double>::infinity(); mx[k] = -mn[k]; }
        auto coord = [&](uint32_t g, int k) { return k < 3 ? items[g].d[k] : items[g].v0[k - 3]; };
        for (uint32_t i = lo; i < hi; ++i)
            for (int k = 0; k < 6; ++k) { const double v = coord(order[i], k); mn[k] = std::min(mn[k], v); mx[k] = std::max(mx[k], v); }
        // two candidate median splits -- along the widest direction axis and along the widest position axis --; the one whose children
        // are the tighter pair wins: kappa (times the scene's size: what a unit of direction costs in (B) / the line test) + sigma / lam,
        // weighted by the children's sizes.  Splitting by direction alone would keep the same-facing patches of different objects
        // together, whose planes (lines) pass nowhere near each other.
        int ad = 0, ap = 3;
        for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ad] - mn[ad]) ad = k;
        for (int k = 4; k < 6; ++k) if (mx[k] - mn[k] > mx[ap] - mn[ap]) ap = k;
        const uint32_t m = lo + (hi - lo + 1) / 2;
        auto split_on = [&](int a) {
            std::nth_element(order.begin() + lo, order.begin() + m, order.begin() + hi, [&](uint32_t x, uint32_t y) {
                const double cx = coord(x, a), cy = coord(y, a);
                return cx < cy || (cx == cy && x < y);
            });
        };
        auto cost_of = [&]() {
            const ConeChild l = summary(lo, m), r = summary(m, hi);
            return (double)(m - lo) * ((double)l.kappa * dir_scale + l.s1) + (double)(hi - m) * ((double)r.kappa * dir_scale + r.s1);
        };
        int a = ad;
        if (mx[ap] - mn[ap] > 0.0 && mx[ad] - mn[ad] > 0.0) {
            split_on(ap);
            const double cp = cost_of();
            split_on(ad);
            const double cd = cost_of();
            if (cp < cd) { a = ap; split_on(ap); }
        } else {
            if (!(mx[ad] - mn[ad] > 0.0)) a = ap;
            split_on(a);
        }
        (void)summary(lo, hi);                                            // the candidates' summaries re-oriented the halves: align the range again
        depth = std::max(depth, d + 1);
        const int32_t lr = m - lo == 1 ? ~(int32_t)items[order[lo]].gid : build(lo, m, d + 1);
        const int32_t rr = hi - m == 1 ? ~(int32_t)items[order[m]].gid : build(m, hi, d + 1);
        store(node, summary(lo, m), summary(m, hi), lr, rr);
        return (int32_t)node;
    }

    void run()
    {
        const uint32_t n = (uint32_t)items.size();
        nodes->clear();
        if (n == 0) return;
        if (n >= 0x7FFFFFFFu) throw std::runtime_error("spt_set_mesh_accel: too many triangles for the leaf encoding");
        order.resize(n);
        std::iota(order.begin(), order.end(), 0u);
        if (n == 1) {
            nodes->resize(node_f4());
            depth = 1;
            store(0, summary(0, 1), summary(0, 0), ~(int32_t)items[0].gid, ~(int32_t)items[0].gid);
            return;
        }
        if (build(0, n, 0) != 0) throw std::runtime_error("cone tree: internal error (root is not node 0)");
    }
};

// the scene's size (largest extent of the vertices' bounding box), for the direction / position balance of the cone trees' splits
double scene_size(const float4* recs, uint32_t ntris)
{
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t g = 0; g < ntris; ++g) {
        const float4* r = recs + 3 * (size_t)g;
        const double v[3][3] = {{r[0].x, r[0].y, r[0].z}, {(double)r[0].x + r[1].x, (double)r[0].y + r[1].y, (double)r[0].z + r[1].z},
                                {(double)r[0].x + r[2].x, (double)r[0].y + r[2].y, (double)r[0].z + r[2].z}};
        for (auto& p : v) for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    const double s = ntris ? std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]}) : 1.0;
    return std::isfinite(s) && s > 0.0 ? s : 1.0;
}

}  // namespace

void build_bvh(const float4* recs, uint32_t ntris, Bvh& out, int form)
{
    out = Bvh{};
    Builder b;
    b.recs = recs; b.rec_f4 = 3; b.out = &out;
    std::vector<Box> boxes(ntris);
    for (uint32_t g = 0; g < ntris; ++g) boxes[g] = padded_box(recs + 3 * (size_t)g);      // throws on non-finite vertices
    ConeBuilder planes, lines;
    planes.planes = true; planes.nodes = &out.planes;
    lines.planes = false; lines.nodes = &out.lines;
    planes.dir_scale = lines.dir_scale = scene_size(recs, ntris);
    for (uint32_t g = 0; g < ntris; ++g) {
        const TriGeom t = tri_geom(recs + 3 * (size_t)g);
        if (t.cls == kTriDead) { ++out.dead_count; continue; }
        (t.cls == kTriRegular ? planes : lines).items.push_back(cone_item(t, g));
        if (t.cls != kTriRegular) continue;
        b.ids.push_back(g);
        for (int a = 0; a < 3; ++a) b.nrm.push_back(t.n[a] / t.nn);
        b.gq.push_back(t.g);
        b.el.push_back(std::max(t.l1, t.l2));
        b.box.push_back(boxes[g]);
        for (int a = 0; a < 3; ++a) b.cen.push_back(0.5f * boxes[g].mn[a] + 0.5f * boxes[g].mx[a]);
    }
    out.regular_count = (uint32_t)planes.items.size();
    out.thin_count = (uint32_t)lines.items.size();
    if (b.ids.empty()) { b.nrm.assign(3, 0.0); b.nrm[0] = 1.0; }             // an empty tree still carries (never read) cones
    b.run();
    if (out.cones.size() < 3 * (out.nodes.size() / 4)) out.cones.resize(3 * (out.nodes.size() / 4), make_float4(1.f, 0.f, 0.f, 2.f));
    planes.run();
    out.flat = form == 1 || (form == 0 && out.thin_count <= kTriFlatLines);     // the thin triangles as a table or as a tree (spt_tribvh.h (3))
    if (out.flat) {
        // groups of lines through a common point: the long edge's two endpoints of every thin triangle are hashed on a grid of 2^-18 of the
        // scene's size; the fullest cell takes every triangle that has an endpoint in it (a pole takes its needles), and so on; what
        // shares no endpoint with three others stands alone (p = v0)
        const double cell = std::max(planes.dir_scale, 1e-30) * 0x1p-18;
        struct End { long long k[3]; uint32_t item; double p[3]; };
        std::vector<End> ends;
        const uint32_t nt = (uint32_t)lines.items.size();
        std::vector<double> elen(nt);
        for (uint32_t i = 0; i < nt; ++i) {
            const TriGeom t = tri_geom(recs + 3 * (size_t)lines.items[i].gid);
            elen[i] = std::max(t.l1, t.l2);
            for (int side = 0; side < 2; ++side) {
                End e{};
                e.item = i;
                for (int a = 0; a < 3; ++a) { e.p[a] = lines.items[i].v0[a] + (side ? lines.items[i].d[a] * elen[i] : 0.0); e.k[a] = (long long)std::floor(e.p[a] / cell); }
                ends.push_back(e);
            }
        }
        // (the direction of an item may have been flipped nowhere yet: cone_item keeps eL's own orientation, so v0 + d |eL| is the far end)
        std::sort(ends.begin(), ends.end(), [](const End& x, const End& y) {
            for (int a = 0; a < 3; ++a) if (x.k[a] != y.k[a]) return x.k[a] < y.k[a];
            return x.item < y.item;
        });
        struct Cell { size_t lo, hi; };
        std::vector<Cell> cells;
        for (size_t i = 0; i < ends.size();) {
            size_t j = i + 1;
            while (j < ends.size() && ends[j].k[0] == ends[i].k[0] && ends[j].k[1] == ends[i].k[1] && ends[j].k[2] == ends[i].k[2]) ++j;
            cells.push_back({i, j});
            i = j;
        }
        std::sort(cells.begin(), cells.end(), [](const Cell& x, const Cell& y) { return x.hi - x.lo > y.hi - y.lo || (x.hi - x.lo == y.hi - y.lo && x.lo < y.lo); });
        std::vector<char> taken(nt, 0);
        auto emit_group = [&](const std::vector<uint32_t>& members, const double p[3]) {
            float4 h = make_float4((float)p[0], (float)p[1], (float)p[2], 0.f);
            const uint32_t cnt = (uint32_t)members.size();
            std::memcpy(&h.w, &cnt, 4);
            out.flat_lines.push_back(h);
            out.flat_line_index.push_back(0xFFFFFFFFu);
            for (uint32_t i : members) {
                const ConeItem& it = lines.items[i];
                const double rel[3] = {it.v0[0] - h.x, it.v0[1] - h.y, it.v0[2] - h.z};
                double cr[3];
                cross3(it.d, rel, cr);
                const double tol = std::sqrt(dot3(cr, cr)) + it.tol + 8.1 * 0x1p-24 * std::sqrt(dot3(rel, rel));
                out.flat_lines.push_back(make_float4((float)it.d[0], (float)it.d[1], (float)it.d[2], round_up(tol * (1.0 + 1e-6) + 1e-30)));
                out.flat_line_index.push_back(it.gid);
            }
        };
        for (const Cell& c : cells) {
            std::vector<uint32_t> members;
            double p[3] = {0, 0, 0};
            for (size_t i = c.lo; i < c.hi; ++i)
                if (!taken[ends[i].item] && (members.empty() || members.back() != ends[i].item)) {
                    members.push_back(ends[i].item);
                    for (int a = 0; a < 3; ++a) p[a] += ends[i].p[a];
                }
            if (members.size() < 4) continue;
            for (int a = 0; a < 3; ++a) p[a] /= (double)members.size();
            for (uint32_t i : members) taken[i] = 1;
            emit_group(members, p);
        }
        for (uint32_t i = 0; i < nt; ++i)
            if (!taken[i]) emit_group(std::vector<uint32_t>{i}, lines.items[i].v0);
    } else {
        lines.run();
    }
    out.ball_depth = std::max(planes.depth, lines.depth);
}

// The regular triangles in whose plane the point `o` lies to within condition (B) of spt_tribvh.h (2), for rays whose LINES pass through
// o and whose origins are at most `extra` away from it along the ray: the only triangles such a ray can be reported by through a
// determinant that is zero to rounding.  (B) holds at the ray's origin ro = o + s d, |s| <= extra: |nh.(ro - v0)| <= tau (R_ro + 2 e),
// R_ro <= R + extra; with (A), |nh.d| < tau, moving back to o costs tau extra more:  |nh.(o - v0)| <= tau (R + 2 e + 2 extra); the
// float evaluation of ro = o + d push and of the normalised direction moves the line by 4 u (|o| + extra) at most.  For the rays
// of depth 0 of a frame -- pinhole camera: extra = 0; smallpt camera: extra = 140 |d| -- this short list (empty, as a rule) replaces
// the walk of the plane tree.
void camera_planes(const float4* recs, uint32_t ntris, const float o[3], float extra, std::vector<uint32_t>& out)
{
    out.clear();
    const double x = std::fabs((double)extra);
    const double slack = 16.0 * 0x1p-24 * (std::max({std::fabs((double)o[0]), std::fabs((double)o[1]), std::fabs((double)o[2])}) + x);
    for (uint32_t g = 0; g < ntris; ++g) {
        const TriGeom t = tri_geom(recs + 3 * (size_t)g);
        if (t.cls != kTriRegular) continue;
        const double r[3] = {(double)o[0] - t.v0[0], (double)o[1] - t.v0[1], (double)o[2] - t.v0[2]};
        const double R = std::sqrt(dot3(r, r)), e = std::max(t.l1, t.l2), tau = kTriBand * t.g;
        if (!(std::fabs(dot3(t.n, r)) / t.nn > tau * (R + 2.0 * e + 2.0 * x) * (1.0 + 0x1p-9) + slack)) out.push_back(g);      // (NaN origins list everything)
    }
}

// Box of a sphere, rounded outward, NOT padded: the traversal inflates node boxes per ray (spt_mesh.hip, closest_sphere_bvh)
static Box sphere_box(const float4 g, float radius)
{
    const double r = std::fabs((double)radius);
    if (!(std::isfinite(g.x) && std::isfinite(g.y) && std::isfinite(g.z) && std::isfinite(r)))
        throw std::runtime_error("spt_set_sphere_accel: a sphere has non-finite centre or radius");
    const double c[3] = {g.x, g.y, g.z};
    Box b;
    for (int a = 0; a < 3; ++a) { b.mn[a] = round_down(c[a] - r); b.mx[a] = round_up(c[a] + r); }
    return b;
}

void build_sphere_bvh(const float4* geom, const float* radius, uint32_t n, Bvh& out)
{
    out = Bvh{};
    // spheres far larger than the rest (the walls and the light of a Cornell box) would make every node box scene-sized:
    // they are tested for every ray instead.  "Far larger" = more than 16 x the median radius; at most kBvhAlways of them.
    std::vector<float> rs(n);
    for (uint32_t i = 0; i < n; ++i) rs[i] = std::fabs(radius[i]);
    std::vector<uint32_t> huge;
    if (n > 0) {
        std::vector<float> sorted(rs);
        std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
        const float cut = 16.0f * sorted[n / 2];
        for (uint32_t i = 0; i < n; ++i) if (rs[i] > cut) huge.push_back(i);
        if (huge.size() > kBvhAlways) {                                   // keep the largest ones
            std::sort(huge.begin(), huge.end(), [&](uint32_t x, uint32_t y) { return rs[x] > rs[y] || (rs[x] == rs[y] && x < y); });
            huge.resize(kBvhAlways);
        }
        std::sort(huge.begin(), huge.end());
    }
    out.always = huge;
    Builder b;
    b.recs = geom; b.rec_f4 = 1; b.out = &out; b.leaf_max = kBvhLeafSpheres;
    size_t h = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (h < huge.size() && huge[h] == i) { ++h; continue; }
        b.ids.push_back(i);
        const Box bx = sphere_box(geom[i], radius[i]);
        b.box.push_back(bx);
        for (int a = 0; a < 3; ++a) b.cen.push_back(0.5f * bx.mn[a] + 0.5f * bx.mx[a]);
    }
    if (b.ids.empty()) b.ids.clear();
    b.run();
}

// Shared structural walk; `expect[g]` = how often primitive g must be referenced (0 for the always-tested spheres).
static bool validate_walk(const float4* recs, uint32_t rec_f4, uint32_t ntris, const std::vector<uint32_t>& expect,
                          const std::vector<Box>& prim_box, const Bvh& bvh, std::string& why)
{
    std::vector<uint32_t> seen(ntris, 0u);
    struct Item { int32_t ref; uint32_t depth; Box bound; };
    std::vector<Item> stack;
    Box all; all.mn[0] = all.mn[1] = all.mn[2] = -INFINITY; all.mx[0] = all.mx[1] = all.mx[2] = INFINITY;
    if (bvh.nodes.size() < 4 || bvh.nodes.size() % 4) { why = "node array size"; return false; }
    stack.push_back({0, 0u, all});
    size_t visited_nodes = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        if (it.ref < 0) {
            const uint32_t code = (uint32_t)~it.ref, first = code >> 4, cnt = code & 15u;
            if (cnt > 15u) { why = "leaf count"; return false; }
            if (it.depth > kBvhMaxDepth) { why = "leaf deeper than the bound"; return false; }
            for (uint32_t k = 0; k < cnt; ++k) {
                if (first + k >= bvh.index.size()) { why = "leaf range"; return false; }
                const uint32_t g = bvh.index[first + k];
                if (g >= ntris) { why = "global index"; return false; }
                ++seen[g];
                if (std::memcmp(&bvh.tris[rec_f4 * (size_t)(first + k)], &recs[rec_f4 * (size_t)g], 16 * rec_f4) != 0) { why = "leaf record differs from the source record"; return false; }
                const Box& pb = prim_box[g];
                for (int a = 0; a < 3; ++a)
                    if (!(pb.mn[a] >= it.bound.mn[a] && pb.mx[a] <= it.bound.mx[a])) { why = "primitive outside its leaf's box"; return false; }
            }
            continue;
        }
        if ((size_t)it.ref * 4 + 3 >= bvh.nodes.size()) { why = "node index"; return false; }
        if (++visited_nodes > bvh.nodes.size() / 4) { why = "cycle"; return false; }
        const float4* nd = &bvh.nodes[4 * (size_t)it.ref];
        Box l, r;
        l.mn[0] = nd[0].x; l.mn[1] = nd[0].y; l.mn[2] = nd[0].z; l.mx[0] = nd[0].w; l.mx[1] = nd[1].x; l.mx[2] = nd[1].y;
        r.mn[0] = nd[1].z; r.mn[1] = nd[1].w; r.mn[2] = nd[2].x; r.mx[0] = nd[2].y; r.mx[1] = nd[2].z; r.mx[2] = nd[2].w;
        int32_t lr, rr;
        std::memcpy(&lr, &nd[3].x, 4); std::memcpy(&rr, &nd[3].y, 4);
        for (const Box* b : {&l, &r})
            for (int a = 0; a < 3; ++a)
                if (b->mn[a] <= b->mx[a] && !(b->mn[a] >= it.bound.mn[a] && b->mx[a] <= it.bound.mx[a])) { why = "child box outside its parent's"; return false; }
        if (it.depth + 1 > kBvhMaxDepth) { why = "node deeper than the bound"; return false; }
        stack.push_back({lr, it.depth + 1, l});
        stack.push_back({rr, it.depth + 1, r});
    }
    for (uint32_t g = 0; g < ntris; ++g)
        if (seen[g] != expect[g]) { why = "primitive " + std::to_string(g) + " referenced " + std::to_string(seen[g]) + " times"; return false; }
    return true;
}

// A cone tree against the triangles it was built over: every triangle of the class in exactly one leaf and, for every ancestor child,
// inside its cone (either orientation) and within its sigma / tau / te (planes) or lam (lines); depth bound respected.
static bool validate_cone_tree(const std::vector<float4>& nodes, bool planes, const std::vector<ConeItem>& item_of, const std::vector<uint32_t>& expect, std::string& why)
{
    const uint32_t f4 = planes ? 6u : 5u, ntris = (uint32_t)expect.size();
    std::vector<uint32_t> seen(ntris, 0u);
    uint32_t want = 0;
    for (uint32_t e : expect) want += e;
    if (want == 0) { if (!nodes.empty()) { why = "nodes without triangles"; return false; } return true; }
    if (nodes.empty() || nodes.size() % f4) { why = "node array size"; return false; }
    struct Item { int32_t ref; uint32_t depth; int parent; };
    struct Link { ConeChild c; int parent; };
    std::vector<Link> chain;
    std::vector<Item> stack;
    stack.push_back({0, 0u, -1});
    size_t visited = 0;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        if (it.ref < 0) {
            const uint32_t g = (uint32_t)~it.ref;
            if (g >= ntris || !expect[g]) { why = "leaf names a triangle of another class"; return false; }
            if (it.depth > kBvhMaxDepth) { why = "leaf deeper than the bound"; return false; }
            ++seen[g];
            const ConeItem& t = item_of[g];
            for (int l = it.parent; l >= 0; l = chain[l].parent) {
                const ConeChild& c = chain[l].c;
                const double al = std::sqrt((double)c.a[0] * c.a[0] + (double)c.a[1] * c.a[1] + (double)c.a[2] * c.a[2]);
                double dm = 0.0, dp = 0.0;
                for (int k = 0; k < 3; ++k) { dm += (t.d[k] - c.a[k] / al) * (t.d[k] - c.a[k] / al); dp += (t.d[k] + c.a[k] / al) * (t.d[k] + c.a[k] / al); }
                if (!(std::sqrt(std::min(dm, dp)) <= (double)c.kappa)) { why = "triangle " + std::to_string(g) + " outside an ancestor's cone"; return false; }
                const double rel[3] = {t.v0[0] - c.p[0], t.v0[1] - c.p[1], t.v0[2] - c.p[2]};
                const double r = std::sqrt(dot3(rel, rel));
                if (planes) {
                    if (!(std::fabs(dot3(t.d, rel)) <= (double)c.s1 && t.tol <= (double)c.s2 && t.tol * (r + 2.0 * t.edge) <= (double)c.s3)) {
                        why = "triangle " + std::to_string(g) + ": sigma / tau / te of an ancestor"; return false;
                    }
                } else {
                    double cr[3];
                    cross3(t.d, rel, cr);
                    if (!(std::sqrt(dot3(cr, cr)) + t.tol + 8.1 * 0x1p-24 * r <= (double)c.s1)) { why = "triangle " + std::to_string(g) + ": lam of an ancestor"; return false; }
                }
            }
            continue;
        }
        if ((size_t)it.ref * f4 + f4 > nodes.size()) { why = "node index"; return false; }
        if (++visited > nodes.size() / f4) { why = "cycle"; return false; }
        if (it.depth + 1 > kBvhMaxDepth) { why = "node deeper than the bound"; return false; }
        const float4* nd = &nodes[(size_t)f4 * it.ref];
        int32_t lr, rr;
        std::memcpy(&lr, &nd[f4 - 1].x, 4); std::memcpy(&rr, &nd[f4 - 1].y, 4);
        for (int side = 0; side < 2; ++side) {
            const float4 ax = nd[side ? 2 : 0], pt = nd[side ? 3 : 1];
            ConeChild c{};
            c.a[0] = ax.x; c.a[1] = ax.y; c.a[2] = ax.z; c.kappa = ax.w; c.p[0] = pt.x; c.p[1] = pt.y; c.p[2] = pt.z; c.s1 = pt.w;
            if (planes) { c.s2 = side ? nd[4].z : nd[4].x; c.s3 = side ? nd[4].w : nd[4].y; }
            if (c.kappa < 0.f) continue;                                      // empty child (a tree over one triangle)
            chain.push_back({c, it.parent});
            stack.push_back({side ? rr : lr, it.depth + 1, (int)chain.size() - 1});
        }
    }
    for (uint32_t g = 0; g < ntris; ++g)
        if (seen[g] != expect[g]) { why = "triangle " + std::to_string(g) + " referenced " + std::to_string(seen[g]) + " times"; return false; }
    return true;
}

// The cones of the spatial tree (spt_tribvh.h (1)): every regular triangle's normal within kappa of each ancestor child's axis (either
// orientation), its 1 / g >= the stored one, its longest edge within the stored one.
static bool validate_cones(const Bvh& bvh, const std::vector<TriGeom>& geo, std::string& why)
{
    if (bvh.cones.size() != 3 * (bvh.nodes.size() / 4)) { why = "cone array size"; return false; }
    struct Link { float4 axis; float iq, ee; int parent; };
    struct Item { int32_t ref; int parent; };
    std::vector<Link> chain;
    std::vector<Item> stack;
    stack.push_back({0, -1});
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        if (it.ref < 0) {
            const uint32_t code = (uint32_t)~it.ref, first = code >> 4, cnt = code & 15u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const TriGeom& t = geo[bvh.index[first + k]];
                const double nh[3] = {t.n[0] / t.nn, t.n[1] / t.nn, t.n[2] / t.nn};
                for (int l = it.parent; l >= 0; l = chain[l].parent) {
                    const Link& c = chain[l];
                    const double ax[3] = {c.axis.x, c.axis.y, c.axis.z};
                    const double al = std::sqrt(dot3(ax, ax));
                    double dm = 0.0, dp = 0.0;
                    for (int a = 0; a < 3; ++a) { dm += (nh[a] - ax[a] / al) * (nh[a] - ax[a] / al); dp += (nh[a] + ax[a] / al) * (nh[a] + ax[a] / al); }
                    if (!(std::sqrt(std::min(dm, dp)) <= (double)c.axis.w)) { why = "a normal outside an ancestor's cone"; return false; }
                    if (!(1.0 / t.g >= (double)c.iq && 1.016 * std::max(t.l1, t.l2) <= (double)c.ee)) { why = "g / edge bound of an ancestor's cone"; return false; }
                }
            }
            continue;
        }
        const float4* nd = &bvh.nodes[4 * (size_t)it.ref];
        const float4* cn = &bvh.cones[3 * (size_t)it.ref];
        int32_t lr, rr;
        std::memcpy(&lr, &nd[3].x, 4); std::memcpy(&rr, &nd[3].y, 4);
        chain.push_back({cn[0], cn[2].x, cn[2].y, it.parent});
        stack.push_back({lr, (int)chain.size() - 1});
        chain.push_back({cn[1], cn[2].z, cn[2].w, it.parent});
        stack.push_back({rr, (int)chain.size() - 1});
    }
    return true;
}

bool validate_bvh(const float4* recs, uint32_t ntris, const Bvh& bvh, std::string& why)
{
    std::vector<Box> pb(ntris);
    std::vector<uint32_t> regular(ntris, 0u), thin(ntris, 0u);
    std::vector<TriGeom> geo(ntris);
    std::vector<ConeItem> item(ntris);
    uint32_t nreg = 0, nthin = 0, ndead = 0;
    for (uint32_t g = 0; g < ntris; ++g) {
        pb[g] = padded_box(recs + 3 * (size_t)g);
        const TriGeom t = geo[g] = tri_geom(recs + 3 * (size_t)g);
        if (t.cls == kTriDead) { ++ndead; continue; }
        item[g] = cone_item(t, g);
        if (t.cls == kTriRegular) { regular[g] = 1u; ++nreg; } else { thin[g] = 1u; ++nthin; }
    }
    if (bvh.regular_count != nreg || bvh.thin_count != nthin || bvh.dead_count != ndead) { why = "triangle class counts"; return false; }
    if (!validate_walk(recs, 3, ntris, regular, pb, bvh, why)) return false;
    if (bvh.flat) {                                                          // the thin triangles as a table of groups
        if (!bvh.lines.empty() || bvh.flat_lines.size() != bvh.flat_line_index.size()) { why = "line table size"; return false; }
        std::vector<uint32_t> seen(ntris, 0u);
        uint32_t listed = 0;
        for (size_t i = 0; i < bvh.flat_lines.size();) {
            const float4 h = bvh.flat_lines[i];
            uint32_t cnt;
            std::memcpy(&cnt, &h.w, 4);
            if (cnt == 0 || i + cnt > bvh.flat_lines.size() - 1) { why = "line table group header"; return false; }
            for (uint32_t k = 1; k <= cnt; ++k) {
                const float4 e = bvh.flat_lines[i + k];
                const uint32_t g = bvh.flat_line_index[i + k];
                if (g >= ntris || !thin[g] || seen[g]++) { why = "line table index"; return false; }
                double dd = 0.0;
                for (int a = 0; a < 3; ++a) { const double d = (&e.x)[a] - item[g].d[a]; dd += d * d; }
                const double rel[3] = {item[g].v0[0] - h.x, item[g].v0[1] - h.y, item[g].v0[2] - h.z};
                double cr[3];
                cross3(item[g].d, rel, cr);
                const double need = std::sqrt(dot3(cr, cr)) + item[g].tol + 8.1 * 0x1p-24 * std::sqrt(dot3(rel, rel));
                if (!(std::sqrt(dd) <= 3e-7 && (double)e.w >= need)) { why = "line table record " + std::to_string(g); return false; }
                ++listed;
            }
            i += (size_t)cnt + 1;
        }
        if (listed != nthin) { why = "line table: thin triangles listed " + std::to_string(listed) + " of " + std::to_string(nthin); return false; }
    } else if (!bvh.flat_lines.empty()) { why = "line table beside the line tree"; return false; }
    if (!validate_cones(bvh, geo, why)) return false;
    if (!validate_cone_tree(bvh.planes, true, item, regular, why)) { why = "plane tree: " + why; return false; }
    if (!bvh.flat && !validate_cone_tree(bvh.lines, false, item, thin, why)) { why = "line tree: " + why; return false; }
    return true;
}

bool validate_sphere_bvh(const float4* geom, const float* radius, uint32_t n, const Bvh& bvh, std::string& why)
{
    std::vector<Box> pb(n);
    std::vector<uint32_t> expect(n, 1u);
    for (uint32_t g = 0; g < n; ++g) pb[g] = sphere_box(geom[g], radius[g]);
    if (bvh.always.size() > kBvhAlways) { why = "always-list too long"; return false; }
    for (size_t k = 0; k < bvh.always.size(); ++k) {
        if (bvh.always[k] >= n || (k && bvh.always[k] <= bvh.always[k - 1])) { why = "always-list not ascending / out of range"; return false; }
        expect[bvh.always[k]] = 0u;
    }
    return validate_walk(geom, 1, n, expect, pb, bvh, why);
}

}  // namespace spt
