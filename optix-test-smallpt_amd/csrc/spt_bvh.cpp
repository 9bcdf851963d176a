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
    uint32_t leaf_max = kBvhLeafTris; // primitives per leaf (<= 7: three bits of the leaf reference)
    std::vector<uint32_t> ids;       // global id of every primitive handed to the builder (empty = identity)
    std::vector<Box> box;            // per primitive
    std::vector<float> cen;          // 3 per primitive
    std::vector<uint32_t> order;     // permutation being partitioned
    Bvh* out;

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
        return ~(int32_t)((first << 3) | (hi - lo));
    }

    // whole hierarchy over the primitives whose boxes / centroids are set; the root is always node 0
    void run()
    {
        const uint32_t n = (uint32_t)box.size();
        if (n >= (1u << 28)) throw std::runtime_error("hierarchy: too many primitives for the leaf encoding");
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

// A needle: the record's normal cross(e1, e2) is tiny against the longest edge squared (zero-area triangles included)
static bool thin_triangle(const float4* r)
{
    const double e1[3] = {r[1].x, r[1].y, r[1].z}, e2[3] = {r[2].x, r[2].y, r[2].z}, n[3] = {r[0].w, r[1].w, r[2].w};
    const double e3[3] = {e2[0] - e1[0], e2[1] - e1[1], e2[2] - e1[2]};
    auto sq = [](const double* v) { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; };
    const double longest2 = std::max({sq(e1), sq(e2), sq(e3)});
    return !(sq(n) > kBvhThinRatio * kBvhThinRatio * longest2 * longest2);
}

void build_bvh(const float4* recs, uint32_t ntris, Bvh& out)
{
    out = Bvh{};
    // two hierarchies of the same layout: the regular triangles (ray segment, distance cut) and the thin ones (whole line, no cut)
    Bvh thin;
    Builder b, t;
    b.recs = recs; b.rec_f4 = 3; b.out = &out;
    t.recs = recs; t.rec_f4 = 3; t.out = &thin;
    for (uint32_t g = 0; g < ntris; ++g) {
        const Box bx = padded_box(recs + 3 * (size_t)g);
        Builder& dst = thin_triangle(recs + 3 * (size_t)g) ? t : b;
        dst.ids.push_back(g);
        dst.box.push_back(bx);
        for (int a = 0; a < 3; ++a) dst.cen.push_back(0.5f * bx.mn[a] + 0.5f * bx.mx[a]);
    }
    const uint32_t nthin = (uint32_t)t.ids.size();
    b.run();
    if (nthin) {
        t.run();
        out.thin_nodes.swap(thin.nodes); out.thin_tris.swap(thin.tris); out.thin_index.swap(thin.index);
        out.thin_count = nthin;
        out.depth = std::max(out.depth, thin.depth);
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
    b.recs = geom; b.rec_f4 = 1; b.out = &out;
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
            const uint32_t code = (uint32_t)~it.ref, first = code >> 3, cnt = code & 7u;
            if (cnt > 7u) { why = "leaf count"; return false; }
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

bool validate_bvh(const float4* recs, uint32_t ntris, const Bvh& bvh, std::string& why)
{
    std::vector<Box> pb(ntris);
    std::vector<uint32_t> regular(ntris, 1u), thin(ntris, 0u);
    uint32_t nthin = 0;
    for (uint32_t g = 0; g < ntris; ++g) {
        pb[g] = padded_box(recs + 3 * (size_t)g);
        if (thin_triangle(recs + 3 * (size_t)g)) { regular[g] = 0u; thin[g] = 1u; ++nthin; }
    }
    if (bvh.thin_count != nthin) { why = "thin-triangle count"; return false; }
    if (!validate_walk(recs, 3, ntris, regular, pb, bvh, why)) return false;
    if (nthin == 0) return bvh.thin_nodes.empty();
    Bvh t;                                                       // the second hierarchy through the same structural walk
    t.nodes = bvh.thin_nodes; t.tris = bvh.thin_tris; t.index = bvh.thin_index;
    if (!validate_walk(recs, 3, ntris, thin, pb, t, why)) { why = "thin hierarchy: " + why; return false; }
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
