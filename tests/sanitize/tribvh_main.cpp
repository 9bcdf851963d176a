// CPU harness of the triangle hierarchy's exhaustive-equivalence (optix-test-smallpt_amd/csrc/spt_tribvh.h): the builder
// (spt_bvh.cpp) and the VERY walk / node-test functions the gfx950 kernel calls, against the exhaustive loop of the reference
// (scene.cpp:95-116 over triIntersect :52-70, smallest t > 0, lowest index among equal t), on random rays and on the rays built to
// break a hierarchy: in a triangle's plane (anywhere in it, also far from the triangle), tilted out of it by 2^-6 ... 2^-26,
// along edges, through vertices, along the supporting lines of needles far beyond their tips, axis-parallel, from far away.
// Compile with -ffp-contract=off (tri_test is the reference's arithmetic: one rounding per operation).  argv[1] = rays per family
// (default 3000).  Prints the walks' cost beside the result.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../optix-test-smallpt_amd/csrc/spt_bvh.h"

namespace {

struct V3 { float x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }

constexpr uint32_t kInfKey = 0x60AD78ECu - 1u;          // key of 1e20f; key(t) = bits(t) - 1 (spt_mesh.hip)

// triIntersect on a record {v0, n.x} {e1, n.y} {e2, n.z}: scene.cpp:56-68
inline float tri_test(const float4* r, V3 ro, V3 rd)
{
    const V3 v0{r[0].x, r[0].y, r[0].z}, e1{r[1].x, r[1].y, r[1].z}, e2{r[2].x, r[2].y, r[2].z}, n{r[0].w, r[1].w, r[2].w};
    const V3 rov0 = ro - v0;
    const V3 q = cross(rov0, rd);
    const float d = (float)(1.0 / (double)dot(rd, n));   // :62 (a double division rounded to float)
    const float u = d * dot(neg(q), e2);
    const float v = d * dot(q, e1);
    const float t = d * dot(neg(n), rov0);
    if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f) return 1e20f;
    return t;
}
inline uint32_t key_of(float t) { uint32_t b; std::memcpy(&b, &t, 4); return b - 1u; }

struct Scene {
    std::string name;
    std::vector<float4> recs;
    std::vector<V3> verts;                              // for aiming rays
    void add(V3 a, V3 b, V3 c)
    {
        const V3 e1 = b - a, e2 = c - a, n = cross(e1, e2);
        recs.push_back(make_float4(a.x, a.y, a.z, n.x));
        recs.push_back(make_float4(e1.x, e1.y, e1.z, n.y));
        recs.push_back(make_float4(e2.x, e2.y, e2.z, n.z));
        verts.push_back(a); verts.push_back(b); verts.push_back(c);
    }
    uint32_t ntris() const { return (uint32_t)(recs.size() / 3); }
};

// latitude / longitude sphere with duplicated pole rows (the layout of makeSphereTriMesh, scene.cpp:3-48): L + 1 rows of 2L + 1
// vertices, the pole rows collapse to points up to the rounding of cos(-pi/2), which is where the needles come from
void add_sphere(Scene& s, V3 c, float radius, uint32_t L)
{
    const uint32_t W = 2 * L;
    const float pi = 3.14159265358979323846f, half_pi = 0.5f * pi;
    const float dphi = pi * 2.f * (1.f / W), dtheta = pi * (1.f / L);
    std::vector<V3> p;
    for (uint32_t j = 0; j <= L; ++j) {
        const float ct = std::cos(-half_pi + j * dtheta), st = std::sin(-half_pi + j * dtheta);
        for (uint32_t i = 0; i <= W; ++i) p.push_back(c + V3{std::sin(i * dphi) * ct, st, std::cos(i * dphi) * ct} * radius);
    }
    for (uint32_t j = 0; j < L; ++j)
        for (uint32_t i = 0; i < W; ++i) {
            const uint32_t o = j * (W + 1);
            s.add(p[o + i], p[o + i + 1], p[o + W + 1 + i + 1]);
            s.add(p[o + i], p[o + W + 1 + i + 1], p[o + i + W + 1]);
        }
}

struct HostStack {
    uint32_t v[40];
    static inline double pushes = 0;
    void push(uint32_t sp, uint32_t x) { pushes += 1; if (sp >= 33) { std::printf("stack overflow\n"); std::exit(1); } v[sp] = x; }
    uint32_t pop(uint32_t sp) const { return v[sp]; }
};

struct Cost { double box_nodes = 0, plane_nodes = 0, line_nodes = 0, tests = 0, rays = 0; };

// the closest hit through the three structures, exactly as closest_triangle_bvh of spt_mesh.hip composes them (cam_list: the ray is one
// of a pinhole camera whose origin's planes are listed, spt_bvh.h camera_planes)
void closest_bvh(const Scene& s, const spt::Bvh& bvh, V3 ro, V3 rd, uint32_t& near_key, uint32_t& near_tri, Cost& cost, const std::vector<uint32_t>* cam_list = nullptr)
{
    near_key = kInfKey; near_tri = 0xFFFFFFFFu;
    float tcut = 1e20f;
    HostStack st;
    auto consider = [&](const float4* r, uint32_t g) {
        const float t = tri_test(r, ro, rd);
        const uint32_t key = key_of(t);
        cost.tests += 1;
        if (key < near_key || (key == near_key && g < near_tri)) {
            if (key < kInfKey) { near_key = key; near_tri = g; tcut = t * 1.0001f; }
        }
    };
    spt::TriQuery q;
    spt::tri_query(ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, q);
    const float ivx = 1.0f / rd.x, ivy = 1.0f / rd.y, ivz = 1.0f / rd.z;
    double visited = 0;
    auto leaf = [&](uint32_t first, uint32_t cnt) {
        visited += 1;
        for (uint32_t k = 0; k < cnt; ++k) consider(&bvh.tris[3 * (size_t)(first + k)], bvh.index[first + k]);
    };
    static const bool no_planes = std::getenv("TRIBVH_NO_PLANES") != nullptr, no_lines = std::getenv("TRIBVH_NO_LINES") != nullptr;   // to see what each structure is there for
    spt::tri_walk_boxes<true>(bvh.nodes.data(), bvh.cones.data(), ro.x, ro.y, ro.z, ivx, ivy, ivz, q.h[0], q.h[1], q.h[2], tcut, st, leaf);
    auto plane = [&](uint32_t g) { cost.plane_nodes += 1; consider(&s.recs[3 * (size_t)g], g); };
    auto line = [&](uint32_t g) { cost.line_nodes += 1; consider(&s.recs[3 * (size_t)g], g); };
    if (!no_planes) {
        if (cam_list) { for (uint32_t g : *cam_list) plane(g); }          // a pinhole camera's ray: the planes through its origin are listed
        else if (!bvh.planes.empty()) spt::tri_walk_planes(bvh.planes.data(), q, st, plane);
    }
    if (!no_lines) {
        if (bvh.flat) { if (bvh.thin_count) spt::tri_scan_lines(bvh.flat_lines.data(), bvh.flat_line_index.data(), (uint32_t)bvh.flat_lines.size(), q, st, line); }
        else if (!bvh.lines.empty()) spt::tri_walk_lines(bvh.lines.data(), q, st, line);
    }
    cost.box_nodes += visited;
    cost.rays += 1;
}

void closest_exhaustive(const Scene& s, V3 ro, V3 rd, uint32_t& near_key, uint32_t& near_tri)
{
    near_key = kInfKey; near_tri = 0xFFFFFFFFu;
    const uint32_t n = s.ntris();
    for (uint32_t g = 0; g < n; ++g) {
        const uint32_t key = key_of(tri_test(&s.recs[3 * (size_t)g], ro, rd));
        if (key < near_key) { near_key = key; near_tri = g; }
    }
}

struct Ray { V3 o, d; int family; };

V3 normalized(V3 v) { const float l = std::sqrt(dot(v, v)); return l > 0 ? v * (1.0f / l) : V3{1, 0, 0}; }

void make_rays(const Scene& s, std::mt19937& rng, size_t per_family, std::vector<Ray>& rays, std::vector<std::string>& names)
{
    std::uniform_real_distribution<float> U(-1.f, 1.f), U01(0.f, 1.f);
    V3 lo{1e30f, 1e30f, 1e30f}, hi{-1e30f, -1e30f, -1e30f};
    for (const V3& v : s.verts) { lo = {std::fmin(lo.x, v.x), std::fmin(lo.y, v.y), std::fmin(lo.z, v.z)}; hi = {std::fmax(hi.x, v.x), std::fmax(hi.y, v.y), std::fmax(hi.z, v.z)}; }
    const V3 ctr = (lo + hi) * 0.5f;
    const V3 ext{std::fmax(hi.x - lo.x, 1e-3f), std::fmax(hi.y - lo.y, 1e-3f), std::fmax(hi.z - lo.z, 1e-3f)};
    const float size = std::sqrt(dot(ext, ext));
    auto rnd_dir = [&]() { V3 d; do { d = {U(rng), U(rng), U(rng)}; } while (dot(d, d) > 1.f || dot(d, d) < 1e-4f); return normalized(d); };
    auto rnd_eye = [&](float reach) { return V3{ctr.x + reach * ext.x * U(rng), ctr.y + reach * ext.y * U(rng), ctr.z + reach * ext.z * U(rng)}; };
    auto pick = [&]() { return (uint32_t)(rng() % s.ntris()); };
    auto tri = [&](uint32_t g, V3& a, V3& e1, V3& e2) {
        const float4* r = &s.recs[3 * (size_t)g];
        a = {r[0].x, r[0].y, r[0].z}; e1 = {r[1].x, r[1].y, r[1].z}; e2 = {r[2].x, r[2].y, r[2].z};
    };
    int fam = 0;
    auto family = [&](const char* name, auto&& gen) {
        names.push_back(name);
        for (size_t k = 0; k < per_family; ++k) { Ray r = gen(); r.family = fam; rays.push_back(r); }
        ++fam;
    };
    family("random", [&]() { return Ray{rnd_eye(1.5f), rnd_dir(), 0}; });
    family("random, far origin", [&]() { const V3 d = rnd_dir(); return Ray{ctr - d * (size * (3.f + 300.f * U01(rng))) + rnd_dir() * size * 0.3f, d, 0}; });
    family("at a vertex", [&]() { const V3 eye = rnd_eye(1.5f); return Ray{eye, normalized(s.verts[rng() % s.verts.size()] - eye), 0}; });
    family("at a point of a triangle", [&]() {
        V3 a, e1, e2; tri(pick(), a, e1, e2);
        float u = U01(rng), v = U01(rng); if (u + v > 1.f) { u = 1.f - u; v = 1.f - v; }
        if (rng() % 4 == 0) v = 0.f;                                           // on an edge
        const V3 eye = rnd_eye(1.5f);
        return Ray{eye, normalized(a + e1 * u + e2 * v - eye), 0};
    });
    family("from a vertex", [&]() { return Ray{s.verts[rng() % s.verts.size()], rnd_dir(), 0}; });
    family("along an edge", [&]() {
        V3 a, e1, e2; tri(pick(), a, e1, e2);
        const V3 d = normalized(rng() % 2 ? e1 : e2);
        return Ray{a - d * (size * U01(rng) * (rng() % 2 ? 1.f : 0.f)), d, 0};
    });
    // in a triangle's plane: origin and direction both in it, anywhere (also far from the triangle), then tilted / lifted out of it
    for (int tilt = -1; tilt <= 26; tilt += (tilt < 6 ? 7 : 4)) {
        std::string nm = tilt < 0 ? "in a plane" : "in a plane, tilted 2^-" + std::to_string(tilt);
        names.push_back(nm);
        for (size_t k = 0; k < per_family; ++k) {
            V3 a, e1, e2; tri(pick(), a, e1, e2);
            const V3 nh = normalized(cross(e1, e2)), b1 = normalized(e1), b2 = normalized(cross(nh, b1));
            const float reach = size * (rng() % 3 ? 1.f : 30.f);
            V3 o = a + b1 * (reach * U(rng)) + b2 * (reach * U(rng));
            const float ang = 3.14159265f * U(rng);
            V3 d = b1 * std::cos(ang) + b2 * std::sin(ang);
            if (tilt >= 0) {
                const float eps = std::ldexp(1.f, -tilt) * (rng() % 2 ? 1.f : -1.f);
                if (rng() % 2) d = normalized(d + nh * eps); else o = o + nh * (eps * reach);
            }
            rays.push_back(Ray{o, d, fam});
        }
        ++fam;
    }
    // lines that cross the supporting line of a triangle's longer edge somewhere (needles: far beyond their tips)
    family("across an edge's line", [&]() {
        V3 a, e1, e2; tri(pick(), a, e1, e2);
        const V3 eL = dot(e1, e1) >= dot(e2, e2) ? e1 : e2;
        const V3 target = a + normalized(eL) * (size * 3.f * U(rng));
        const V3 eye = rnd_eye(2.f);
        return Ray{eye, normalized(target - eye), 0};
    });
    family("axis-parallel through a vertex", [&]() {
        const int ax = (int)(rng() % 3); const float sgn = rng() % 2 ? 1.f : -1.f;
        V3 d{0, 0, 0}; (&d.x)[ax] = sgn;
        const V3 v = s.verts[rng() % s.verts.size()];
        return Ray{v - d * (size * (rng() % 2 ? 2.f : 0.f)), d, 0};
    });
    family("from a surface along its normal", [&]() {
        V3 a, e1, e2; tri(pick(), a, e1, e2);
        return Ray{a + e1 * 0.3f + e2 * 0.3f, normalized(cross(e1, e2)) * (rng() % 2 ? 1.f : -1.f), 0};
    });
    family("unnormalised direction", [&]() { return Ray{rnd_eye(1.5f), rnd_dir() * std::pow(10.f, 3.f * U(rng)), 0}; });
}

}  // namespace

int main(int argc, char** argv)
{
    const size_t per_family = argc > 1 ? (size_t)std::atol(argv[1]) : 3000;
    std::mt19937 rng(2024);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<Scene> scenes;
    { Scene s; s.name = "two tessellated spheres (L = 24)"; add_sphere(s, {-1, 0, -4}, 1.f, 24); add_sphere(s, {1.5f, 0, -5}, 1.f, 24); scenes.push_back(s); }
    { Scene s; s.name = "one fine sphere (L = 48: thin polar triangles)"; add_sphere(s, {0.3f, -0.2f, 7.f}, 2.5f, 48); scenes.push_back(s); }
    {   // a room of large quads with a ball in it, far from the origin
        Scene s; s.name = "room + ball, 3e4 away from the origin";
        const V3 o{30000.f, -20000.f, 15000.f};
        const float h = 50.f;
        const V3 c[8] = {{-h, -h, -h}, {h, -h, -h}, {h, h, -h}, {-h, h, -h}, {-h, -h, h}, {h, -h, h}, {h, h, h}, {-h, h, h}};
        const int f[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 5, 4}, {2, 3, 7, 6}, {0, 3, 7, 4}, {1, 2, 6, 5}};
        for (auto& q : f) { s.add(o + c[q[0]], o + c[q[1]], o + c[q[2]]); s.add(o + c[q[0]], o + c[q[2]], o + c[q[3]]); }
        add_sphere(s, o + V3{10, -30, 5}, 16.5f, 12);
        scenes.push_back(s);
    }
    {
        Scene s; s.name = "triangle soup";
        for (int i = 0; i < 2500; ++i) {
            const V3 c{10.f * U(rng), 10.f * U(rng), 10.f * U(rng)};
            const float sc = std::pow(10.f, U(rng));
            s.add(c + V3{U(rng), U(rng), U(rng)} * sc, c + V3{U(rng), U(rng), U(rng)} * sc, c + V3{U(rng), U(rng), U(rng)} * sc);
        }
        scenes.push_back(s);
    }
    {
        Scene s; s.name = "coplanar soup (y = 3) with slivers and degenerate triangles";
        for (int i = 0; i < 1200; ++i) {
            const V3 c{10.f * U(rng), 3.f, 10.f * U(rng)};
            V3 a = c + V3{U(rng), 0, U(rng)}, b = c + V3{U(rng), 0, U(rng)}, d = c + V3{U(rng), 0, U(rng)};
            if (i % 7 == 0) d = a + (b - a) * 0.5f + V3{1e-5f * U(rng), 0, 1e-5f * U(rng)};      // sliver
            if (i % 31 == 0) b = a;                                                               // an edge of length zero
            if (i % 37 == 0) d = a + (b - a) * 2.f;                                               // collinear
            s.add(a, b, d);
        }
        scenes.push_back(s);
    }
    { Scene s; s.name = "one triangle"; s.add({-1, -1, -3}, {1, -1, -3}, {0, 1, -3}); scenes.push_back(s); }
    {
        Scene s; s.name = "tiny ball 1e-3 (L = 8) beside a 1e3 ball (L = 16)";
        add_sphere(s, {0, 0, 0}, 1e-3f, 8); add_sphere(s, {0, -1e3f - 2.f, -6.f}, 1e3f, 16);
        scenes.push_back(s);
    }
    size_t total = 0, mismatches = 0;
    for (int form = 1; form <= 2; ++form)
    for (const Scene& s0 : scenes) {
        Scene s = s0;
        s.name = std::string(form == 1 ? "[line table] " : "[line tree]  ") + s0.name;
        spt::Bvh bvh;
        spt::build_bvh(s.recs.data(), s.ntris(), bvh, form);
        std::string why;
        if (!spt::validate_bvh(s.recs.data(), s.ntris(), bvh, why)) { std::printf("invalid hierarchy (%s): %s\n", s.name.c_str(), why.c_str()); return 1; }
        std::vector<Ray> rays;
        std::vector<std::string> names;
        make_rays(s, rng, per_family, rays, names);
        std::vector<size_t> bad(names.size(), 0), hits(names.size(), 0);
        Cost cost;
        std::vector<Cost> fcost(names.size());
        const bool verbose = std::getenv("TRIBVH_VERBOSE") != nullptr;
        for (const Ray& r : rays) {
            uint32_t k0, t0, k1, t1;
            closest_exhaustive(s, r.o, r.d, k0, t0);
            closest_bvh(s, bvh, r.o, r.d, k1, t1, cost);
            if (verbose) { uint32_t k2, t2; closest_bvh(s, bvh, r.o, r.d, k2, t2, fcost[r.family]); }
            if (k0 < kInfKey) ++hits[r.family];
            if (k0 != k1 || t0 != t1) {
                if (bad[r.family]++ < 2)
                    std::printf("  MISMATCH %s / %s: ray (%.9g %.9g %.9g) (%.9g %.9g %.9g): exhaustive key %08x tri %u, hierarchy key %08x tri %u\n", s.name.c_str(),
                                names[r.family].c_str(), r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, k0, t0, k1, t1);
            }
        }
        // cameras: one point per batch that every ray's line passes through -- most of them placed IN a triangle's plane --, rays starting
        // there (pinhole) or pushed along their direction (smallpt camera), the plane tree replaced by the point's list
        {
            names.push_back("camera rays (the planes through the camera point listed)");
            bad.push_back(0); hits.push_back(0); fcost.emplace_back();
            const int fam = (int)names.size() - 1;
            std::uniform_real_distribution<float> U(-1.f, 1.f);
            const size_t cams = 12, per_cam = (per_family + cams - 1) / cams;
            for (size_t c = 0; c < cams; ++c) {
                const float4* r = &s.recs[3 * (size_t)(rng() % s.ntris())];
                const V3 a{r[0].x, r[0].y, r[0].z}, e1{r[1].x, r[1].y, r[1].z}, e2{r[2].x, r[2].y, r[2].z};
                V3 o = a + e1 * (3.f * U(rng)) + e2 * (3.f * U(rng));                     // in the plane of a triangle, beside it
                if (c % 3 == 0) o = o + V3{U(rng), U(rng), U(rng)} * (0.5f * std::sqrt(dot(e1, e1)));   // ... or not
                const float of[3] = {o.x, o.y, o.z};
                const float push = c % 2 ? 0.f : 1.4f * std::sqrt(dot(e1, e1));           // pinhole, or the smallpt camera's ro = o + d push
                std::vector<uint32_t> list;
                spt::camera_planes(s.recs.data(), s.ntris(), of, push * 1.3f, list);
                const V3 o_cam = o;
                for (size_t k = 0; k < per_cam; ++k) {
                    V3 dd;
                    if (k % 2) { const float u = U(rng), v = U(rng); dd = normalized(e1 * u + e2 * v); }     // in that plane
                    else dd = normalized(V3{U(rng), U(rng), U(rng)});
                    dd = dd * (1.f + 0.25f * U(rng));                                        // |d| <= 1.25 < 1.3
                    const V3 d = normalized(dd);
                    o = o_cam + dd * push;
                    uint32_t k0, t0, k1, t1;
                    closest_exhaustive(s, o, d, k0, t0);
                    closest_bvh(s, bvh, o, d, k1, t1, cost, &list);
                    if (k0 < kInfKey) ++hits[fam];
                    if (k0 != k1 || t0 != t1) {
                        if (bad[fam]++ < 2) std::printf("  MISMATCH %s / pinhole: origin (%.9g %.9g %.9g) dir (%.9g %.9g %.9g): exhaustive key %08x tri %u, hierarchy key %08x tri %u (list of %zu)\n",
                                                        s.name.c_str(), o.x, o.y, o.z, d.x, d.y, d.z, k0, t0, k1, t1, list.size());
                    }
                    rays.push_back(Ray{o, d, fam});
                }
            }
        }
        size_t b = 0, h = 0;
        for (size_t f = 0; f < names.size(); ++f) { b += bad[f]; h += hits[f]; }
        std::printf("%-70s %6u triangles (%u regular, %u thin, %u dead) %7zu rays %7zu hits  mismatches %zu | per ray: %.1f leaves of the box tree, "
                    "%.2f plane candidates, %.2f line candidates, %.1f tests\n", s.name.c_str(), s.ntris(), bvh.regular_count, bvh.thin_count,
                    bvh.dead_count, rays.size(), h, b, cost.box_nodes / cost.rays, cost.plane_nodes / cost.rays, cost.line_nodes / cost.rays, cost.tests / cost.rays);
        if (verbose)
            for (size_t f = 0; f < names.size(); ++f)
                std::printf("    %-40s box leaves %.1f, plane candidates %.2f, line candidates %.2f, tests %.1f\n", names[f].c_str(), fcost[f].box_nodes / fcost[f].rays,
                            fcost[f].plane_nodes / fcost[f].rays, fcost[f].line_nodes / fcost[f].rays, fcost[f].tests / fcost[f].rays);
        for (size_t f = 0; f < names.size(); ++f) if (bad[f]) std::printf("    %-40s %zu of %zu differ\n", names[f].c_str(), bad[f], per_family);
        total += rays.size(); mismatches += b;
    }
    std::printf("stack pushes per ray (all walks) %.1f\n", HostStack::pushes / (double)total);
    std::printf("rays %zu, mismatches %zu, %s\n", total, mismatches, mismatches ? "tribvh harness FAILED" : "tribvh harness ok");
    return mismatches ? 1 : 0;
}
