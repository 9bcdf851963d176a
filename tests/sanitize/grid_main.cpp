// CPU harness (ASan/UBSan) of the sphere grid: host builder (optix-test-smallpt_amd/csrc/spt_grid.cpp) + the traversal arithmetic the
// gfx950 kernel uses (spt_grid.h: grid_ray_ok / grid_walk_*).  For random and adversarial rays over several kinds of sphere tables
// the closest hit found through the grid must equal -- key AND index -- that of the exhaustive loop of smallpt.cpp:54-70 over
// scene.cpp:129-140, evaluated with the same binary32 arithmetic (compile with -ffp-contract=off).  A deliberately under-registered
// grid is the negative control: the comparison must be able to fail.
//   grid_main            run the checks, print "grid harness ok <rays>"
//   grid_main stats R    walk statistics of a config-5-like scene at R cells per sphere (resolution tuning)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../optix-test-smallpt_amd/csrc/spt_grid.h"

namespace {

constexpr uint32_t kEpsBias = 0x38D1B717u + 1u;                  // bits(1e-4f) + 1
constexpr uint32_t kInfKey = 0x60AD78ECu - kEpsBias;             // key of 1e20f

uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// intersectAnalytic on integer keys, as the kernels evaluate it (spt_grid.hip sphere_key)
uint32_t sphere_key(const float4 g, const float o[3], const float d[3])
{
    const float opx = g.x - o[0], opy = g.y - o[1], opz = g.z - o[2];
    const float bb = opx * d[0] + opy * d[1] + opz * d[2];
    const float det = bb * bb - (opx * opx + opy * opy + opz * opz) + g.w;
    const float sd = std::sqrt(det);
    const uint32_t k1 = f2u(bb - sd) - kEpsBias, k2 = f2u(bb + sd) - kEpsBias;
    return k1 < k2 ? k1 : k2;
}

struct Hit { uint32_t key, index; };

Hit exhaustive(const std::vector<float4>& geom, const float o[3], const float d[3])
{
    Hit h{kInfKey, 0xFFFFFFFFu};
    for (uint32_t i = 0; i < geom.size(); ++i) {
        const uint32_t k = sphere_key(geom[i], o, d);
        if (k < h.key) { h.key = k; h.index = i; }               // strict '<': lowest index wins ties (smallpt.cpp:61)
    }
    return h;
}

struct WalkStats { unsigned long long rays = 0, grid_rays = 0, steps = 0, tests = 0, always = 0, late = 0; };

Hit through_grid(const std::vector<float4>& geom, const spt::SphereGrid& g, const float o[3], const float d[3], WalkStats& st)
{
    ++st.rays;
    float t_ok;
    if (!spt::grid_ray_ok(g.P, o[0], o[1], o[2], d[0], d[1], d[2], t_ok)) return exhaustive(geom, o, d);
    ++st.grid_rays;
    Hit h{kInfKey, 0xFFFFFFFFu};
    auto consider = [&](uint32_t i) {
        const uint32_t k = sphere_key(geom[i], o, d);
        if ((k < h.key || (k == h.key && i < h.index)) && k < kInfKey) { h.key = k; h.index = i; }
    };
    for (uint32_t i : g.always) { consider(i); ++st.always; }
    spt::GridWalk w;
    spt::grid_walk_begin(g.P, o[0], o[1], o[2], d[0], d[1], d[2], w);
    for (int guard = 0;; ++guard) {
        if (guard > 3 * spt::kGridMaxDim + 8) { std::printf("walk does not terminate\n"); std::exit(1); }
        if (w.ci >= g.cells.size()) { std::printf("walk left the table\n"); std::exit(1); }
        const uint32_t hd = g.cells[w.ci];
        if (hd == spt::kGridBorder) break;
        const uint32_t f = hd >> spt::kGridCountBits, c = hd & ((1u << spt::kGridCountBits) - 1u);
        for (uint32_t k = 0; k < c; ++k) { consider(g.refs[f + k]); ++st.tests; }
        const float m = spt::grid_walk_exit(w);
        const float near_t = h.key == kInfKey ? 1e20f : u2f(h.key + kEpsBias);
        if (!(m < near_t)) break;
        spt::grid_walk_step(w.tx, w.ty, w.tz, w.dtx, w.dty, w.dtz, w.sx, w.sy, w.sz, w.ci, m);
        ++st.steps;
    }
    // the walk has stopped (hit before the cell's exit, or border): its answer stands only if it lies within the ray's valid range
    if ((h.key == kInfKey ? 1e20f : u2f(h.key + kEpsBias)) > t_ok) { --st.grid_rays; ++st.late; return exhaustive(geom, o, d); }
    return h;
}

struct Scene { std::vector<float4> geom; std::vector<float> radius; };

void add(Scene& s, float x, float y, float z, float r) { s.geom.push_back(make_float4(x, y, z, r * r)); s.radius.push_back(r); }

void cornell_walls(Scene& s)
{
    add(s, 1e5f + 1, 40.8f, 81.6f, 1e5f); add(s, -1e5f + 99, 40.8f, 81.6f, 1e5f); add(s, 50, 40.8f, 1e5f, 1e5f);
    add(s, 50, 40.8f, -1e5f + 170, 1e5f); add(s, 50, 1e5f, 81.6f, 1e5f); add(s, 50, -1e5f + 81.6f, 81.6f, 1e5f);
    add(s, 50, 681.6f - .27f, 81.6f, 600);
}

Scene make_scene(int kind, uint32_t n, std::mt19937& rng)
{
    std::uniform_real_distribution<float> u(0.f, 1.f);
    Scene s;
    if (kind == 0 || kind == 1) cornell_walls(s);                                        // 0: config-5-like, 1: clustered sizes inside the box
    while (s.geom.size() < n) {
        float r = kind == 0 ? 0.5f + 2 * u(rng) : std::pow(10.f, -1.5f + 2.3f * u(rng));
        float c[3] = {5 + 90 * u(rng), 3 + 70 * u(rng), 10 + 140 * u(rng)};
        if (kind == 3) { c[0] = 50 + 3 * u(rng); c[1] = 40 + 3 * u(rng); c[2] = 80 + 3 * u(rng); }   // everything in a few cells
        if (kind == 4) { c[0] = 50; c[1] = 40; c[2] = 80; r = 0.1f + 0.01f * (float)s.geom.size(); }   // concentric shells
        if (kind == 5) { c[1] = 40; c[2] = 80; r = 0.3f; }                                             // a line of spheres: degenerate extent in y, z
        if (kind == 6) { c[0] += 4e4f; c[1] -= 3e4f; c[2] += 6e4f; }                                   // the table 300 ... 600 extents away from the origin: the face coordinates' own rounding (spt_grid.h (3)(ii))
        if (kind == 7 || kind == 8) { c[0] = 2000 * u(rng); c[1] = 40 + u(rng); c[2] = 80 + u(rng); r = 0.3f; }   // 128 cells along x, one or two along y and z
        if (kind == 8) { c[0] += 9e5f; c[1] += 9e5f; c[2] -= 9e5f; }                                   // ... 450 extents away
        if (kind >= 2 && !s.geom.empty() && u(rng) < 0.05f) { const float4 g = s.geom[rng() % s.geom.size()]; c[0] = g.x; c[1] = g.y; c[2] = g.z; }
        add(s, c[0], c[1], c[2], r);
    }
    return s;
}

void unit(float d[3]) { const float l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); for (int a = 0; a < 3; ++a) d[a] = d[a] * (1.0f / l); }

// rays of the kinds a path tracer produces plus adversarial ones
void make_ray(const Scene& s, const spt::SphereGrid& g, int kind, std::mt19937& rng, float o[3], float d[3])
{
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::normal_distribution<float> nrm(0.f, 1.f);
    const spt::GridParams& P = g.P;
    auto inside = [&] { for (int a = 0; a < 3; ++a) o[a] = P.gmin[a] + (P.gmax[a] - P.gmin[a]) * u(rng); };
    auto anydir = [&] { do { for (int a = 0; a < 3; ++a) d[a] = nrm(rng); } while (d[0] == 0 && d[1] == 0 && d[2] == 0); unit(d); };
    inside(); anydir();
    const float4 sp = s.geom[rng() % s.geom.size()];
    const float r = std::sqrt(sp.w);
    switch (kind) {
    case 0: break;                                                                       // inside, any direction
    case 1: for (int a = 0; a < 3; ++a) o[a] = P.gmin[a] + (P.gmax[a] - P.gmin[a]) * (3 * u(rng) - 1); break;   // around the box
    case 2: {                                                                            // leaves a sphere's surface like a bounce (D3 offset)
        float nn[3] = {nrm(rng), nrm(rng), nrm(rng)}; unit(nn);
        const float sgn = u(rng) < 0.5f ? 1.f : -1.f;
        o[0] = sp.x + nn[0] * (r + sgn * 0.02f); o[1] = sp.y + nn[1] * (r + sgn * 0.02f); o[2] = sp.z + nn[2] * (r + sgn * 0.02f);
        break;
    }
    case 3: {                                                                            // aimed to graze a sphere
        float nn[3] = {nrm(rng), nrm(rng), nrm(rng)}; unit(nn);
        const float k = r * (1.0f + (u(rng) - 0.5f) * 1e-3f);
        const float tgt[3] = {sp.x + nn[0] * k, sp.y + nn[1] * k, sp.z + nn[2] * k};
        // direction perpendicular to nn through the target point
        float t2[3] = {nrm(rng), nrm(rng), nrm(rng)};
        const float dp = t2[0] * nn[0] + t2[1] * nn[1] + t2[2] * nn[2];
        for (int a = 0; a < 3; ++a) t2[a] -= dp * nn[a];
        if (t2[0] == 0 && t2[1] == 0 && t2[2] == 0) t2[0] = 1;
        unit(t2);
        const float back = 100 * u(rng);
        for (int a = 0; a < 3; ++a) { o[a] = tgt[a] - back * t2[a]; d[a] = t2[a]; }
        break;
    }
    case 4: { const int a = rng() % 3; d[0] = d[1] = d[2] = 0; d[a] = u(rng) < 0.5f ? 1.f : -1.f; break; }     // axis-parallel
    case 5: {                                                                            // axis-parallel, origin exactly on cell faces
        const int a = rng() % 3; d[0] = d[1] = d[2] = 0; d[a] = u(rng) < 0.5f ? 1.f : -1.f;
        for (int b = 0; b < 3; ++b) o[b] = P.gmin[b] + (float)(rng() % (uint32_t)(P.dim[b] + 1)) * P.cell[b];
        break;
    }
    case 6: { const int a = rng() % 3; d[a] = d[a] * 1e-30f; break; }                                          // one component almost zero
    case 7: { const int a = rng() % 3; d[a] = 0.f; if (d[0] == 0 && d[1] == 0 && d[2] == 0) d[(a + 1) % 3] = 1; unit(d); break; }
    case 8: { const float k = 1.0f + (u(rng) - 0.5f) * 6e-6f; for (int a = 0; a < 3; ++a) d[a] *= k; break; }  // |d|^2 - 1 around the limit of the ray test
    case 11: {                                                                           // drifted direction length (a long mirror chain): valid up to t_ok only
        const float k = 1.0f + (u(rng) - 0.5f) * std::pow(10.f, -5.f + 3.5f * u(rng));
        for (int a = 0; a < 3; ++a) d[a] *= k;
        if (u(rng) < 0.5f) {                                                             // ... starting inside a sphere, like the chains trapped in a mirror ball
            float nn[3] = {nrm(rng), nrm(rng), nrm(rng)}; unit(nn);
            const float in = r * 0.9f * u(rng);
            o[0] = sp.x + nn[0] * in; o[1] = sp.y + nn[1] * in; o[2] = sp.z + nn[2] * in;
        }
        break;
    }
    case 9: {                                                                            // through a sphere centre from far away
        for (int a = 0; a < 3; ++a) o[a] = P.gmin[a] + (P.gmax[a] - P.gmin[a]) * (1.6f * u(rng) - 0.3f);
        d[0] = sp.x - o[0]; d[1] = sp.y - o[1]; d[2] = sp.z - o[2];
        if (d[0] == 0 && d[1] == 0 && d[2] == 0) d[0] = 1;
        unit(d);
        break;
    }
    default: {                                                                           // along a cell edge / diagonal through cell corners
        for (int b = 0; b < 3; ++b) o[b] = P.gmin[b] + (float)(rng() % (uint32_t)(P.dim[b] + 1)) * P.cell[b];
        d[0] = P.cell[0] * (float)((int)(rng() % 3) - 1); d[1] = P.cell[1] * (float)((int)(rng() % 3) - 1); d[2] = P.cell[2] * (float)((int)(rng() % 3) - 1);
        if (d[0] == 0 && d[1] == 0 && d[2] == 0) d[2] = 1;
        unit(d);
    }
    }
}

}  // namespace

int main(int argc, char** argv)
{
    std::mt19937 rng(2024);
    if (argc > 2 && std::strcmp(argv[1], "stats") == 0) {
        const double density = std::atof(argv[2]);
        Scene s = make_scene(0, 1024, rng);
        spt::SphereGrid g;
        spt::build_sphere_grid(s.geom.data(), s.radius.data(), 1024, density, 150 * 1024, g);
        if (!g.usable) { std::printf("not usable: %s\n", g.why.c_str()); return 1; }
        WalkStats st;
        float o[3], d[3];
        for (int i = 0; i < 200000; ++i) { make_ray(s, g, 2, rng, o, d); (void)through_grid(s.geom, g, o, d, st); }
        std::printf("density %.1f: dim %d x %d x %d, cell %.2f, %u refs, %zu B; per ray: %.2f steps, %.2f grid tests, %.1f always tests; grid rays %.3f\n", density,
                    g.P.dim[0], g.P.dim[1], g.P.dim[2], g.P.cell[0], g.P.nrefs, g.lds_bytes(), (double)st.steps / st.grid_rays, (double)st.tests / st.grid_rays,
                    (double)st.always / st.grid_rays, (double)st.grid_rays / st.rays);
        return 0;
    }
    unsigned long long rays = 0, mismatches = 0, control_mismatches = 0;
    WalkStats st;
    const struct { int kind; uint32_t n; double density; size_t budget; } cases[] = {
        {0, 1024, 12, 150 * 1024}, {0, 1024, 3, 150 * 1024}, {0, 300, 40, 150 * 1024}, {1, 600, 12, 150 * 1024}, {2, 257, 12, 150 * 1024},
        {3, 200, 12, 150 * 1024}, {4, 120, 12, 150 * 1024}, {5, 64, 12, 150 * 1024}, {2, 25, 12, 150 * 1024}, {1, 4096, 12, 150 * 1024}, {0, 1024, 12, 12 * 1024},
        {2, 1, 12, 150 * 1024}, {6, 1024, 4, 150 * 1024}, {6, 300, 40, 150 * 1024}, {7, 2000, 4, 150 * 1024}, {8, 2000, 4, 150 * 1024}};
    for (const auto& cs : cases) {
        Scene s = make_scene(cs.kind, cs.n, rng);
        spt::SphereGrid g;
        spt::build_sphere_grid(s.geom.data(), s.radius.data(), (uint32_t)s.geom.size(), cs.density, cs.budget, g);
        if (!g.usable) { std::printf("case kind %d n %u: grid not usable (%s)\n", cs.kind, cs.n, g.why.c_str()); return 1; }
        std::string why;
        if (!spt::validate_sphere_grid(s.geom.data(), s.radius.data(), (uint32_t)s.geom.size(), g, why)) { std::printf("case kind %d n %u: invalid grid: %s\n", cs.kind, cs.n, why.c_str()); return 1; }
        if (g.lds_bytes() > cs.budget) { std::printf("case kind %d n %u: %zu bytes exceed the budget\n", cs.kind, cs.n, g.lds_bytes()); return 1; }
        const int per_kind = cs.n > 2000 ? 1500 : 6000;
        float o[3], d[3];
        for (int rk = 0; rk <= 11; ++rk)
            for (int i = 0; i < per_kind; ++i) {
                make_ray(s, g, rk, rng, o, d);
                const Hit a = exhaustive(s.geom, o, d), b = through_grid(s.geom, g, o, d, st);
                ++rays;
                if (a.key != b.key || a.index != b.index) {
                    if (++mismatches <= 5) std::printf("MISMATCH scene kind %d n %u ray kind %d: exhaustive (%08x, %u) grid (%08x, %u) o=(%g %g %g) d=(%g %g %g)\n", cs.kind, cs.n, rk,
                                                       a.key, a.index, b.key, b.index, o[0], o[1], o[2], d[0], d[1], d[2]);
                }
            }
        if (cs.kind == 0 && cs.density == 12 && cs.budget > 100 * 1024) {
            // negative control: drop every reference of a sphere outside the cell that holds its centre
            spt::SphereGrid bad = g;
            for (size_t k = 0; k < bad.cells.size(); ++k) {
                if (bad.cells[k] == spt::kGridBorder) continue;
                const uint32_t f = bad.cells[k] >> spt::kGridCountBits, c = bad.cells[k] & ((1u << spt::kGridCountBits) - 1u);
                const int32_t x = (int32_t)(k % (size_t)bad.P.stride_y) - 1, y = (int32_t)((k / (size_t)bad.P.stride_y) % (size_t)(bad.P.dim[1] + 2)) - 1, z = (int32_t)(k / (size_t)bad.P.stride_z) - 1;
                uint32_t kept = 0;
                for (uint32_t j = 0; j < c; ++j) {
                    const float4 q = s.geom[bad.refs[f + j]];
                    const int32_t cx = (int32_t)std::floor((q.x - bad.P.gmin[0]) / bad.P.cell[0]), cy = (int32_t)std::floor((q.y - bad.P.gmin[1]) / bad.P.cell[1]), cz = (int32_t)std::floor((q.z - bad.P.gmin[2]) / bad.P.cell[2]);
                    if (cx == x && cy == y && cz == z) bad.refs[f + kept++] = bad.refs[f + j];
                }
                bad.cells[k] = (f << spt::kGridCountBits) | kept;
            }
            WalkStats dummy;
            for (int i = 0; i < 20000; ++i) {
                make_ray(s, bad, 2, rng, o, d);
                const Hit a = exhaustive(s.geom, o, d), b = through_grid(s.geom, bad, o, d, dummy);
                if (a.key != b.key || a.index != b.index) ++control_mismatches;
            }
        }
    }
    // non-finite input is refused
    try {
        Scene s; add(s, NAN, 0, 0, 1);
        spt::SphereGrid g;
        spt::build_sphere_grid(s.geom.data(), s.radius.data(), 1, 12, 150 * 1024, g);
        std::printf("non-finite input accepted\n");
        return 1;
    } catch (const std::runtime_error&) {
    }
    std::printf("rays %llu (through the grid %llu, %llu handed to the exhaustive loop at t_ok), mismatches %llu, negative control mismatches %llu\n", rays, st.grid_rays, st.late, mismatches, control_mismatches);
    if (mismatches != 0) return 1;
    if (control_mismatches == 0) { std::printf("the negative control did not fail: the comparison proves nothing\n"); return 1; }
    if (st.grid_rays * 2 < rays) { std::printf("fewer than half of the rays took the grid\n"); return 1; }
    std::printf("grid harness ok %llu\n", rays);
    return 0;
}
