// grid_sched_sim.cpp -- CPU model of how a wave of the grid kernel (csrc/spt_grid.hip) spends its vector-instruction issue slots
// under different scheduling designs, on config-5-like path traces (1017 random spheres inside the Cornell walls).  The traversal is
// the product's (csrc/spt_grid.{h,cpp}: builder + grid_walk_*), the shading is a plain float restatement (scheduling statistics only:
// material classes, path lengths, glass splits), the cost of a loop body is its VALU instruction count in the gfx950 ISA of the
// kernel (tools/isa_blocks.py).  Output per design: VALU wave-instructions per 64 lane-bounces and the lane utilisation they imply.
//   build: hipcc -x c++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/grid_sched_sim.cpp optix-test-smallpt_amd/csrc/spt_grid.cpp -o /tmp/grid_sched_sim
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../optix-test-smallpt_amd/csrc/spt_grid.h"

namespace {

constexpr uint32_t kEpsBias = 0x38D1B717u + 1u, kInfKey = 0x60AD78ECu - kEpsBias;
uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct V { float x, y, z; };
V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V operator*(V a, float s) { return {a.x * s, a.y * s, a.z * s}; }
float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V norm(V a) { return a * (1.0f / std::sqrt(dot(a, a))); }

uint32_t sphere_key(const float4 g, V o, V d)
{
    const V op{g.x - o.x, g.y - o.y, g.z - o.z};
    const float bb = dot(op, d), det = bb * bb - dot(op, op) + g.w, sd = std::sqrt(det);
    const uint32_t k1 = f2u(bb - sd) - kEpsBias, k2 = f2u(bb + sd) - kEpsBias;
    return k1 < k2 ? k1 : k2;
}

struct Scene { std::vector<float4> geom; std::vector<float> radius; std::vector<int> refl; std::vector<V> color, emis; };

void add(Scene& s, float x, float y, float z, float r, V e, V c, int refl)
{
    s.geom.push_back(make_float4(x, y, z, r * r)); s.radius.push_back(r); s.refl.push_back(refl); s.color.push_back(c); s.emis.push_back(e);
}

Scene config5_like(uint32_t n, std::mt19937& rng)
{
    std::uniform_real_distribution<float> u(0.f, 1.f);
    Scene s;
    add(s, 1e5f + 1, 40.8f, 81.6f, 1e5f, {0, 0, 0}, {.75f, .25f, .25f}, 0); add(s, -1e5f + 99, 40.8f, 81.6f, 1e5f, {0, 0, 0}, {.25f, .25f, .75f}, 0);
    add(s, 50, 40.8f, 1e5f, 1e5f, {0, 0, 0}, {.75f, .75f, .75f}, 0); add(s, 50, 40.8f, -1e5f + 170, 1e5f, {0, 0, 0}, {0, 0, 0}, 0);
    add(s, 50, 1e5f, 81.6f, 1e5f, {0, 0, 0}, {.75f, .75f, .75f}, 0); add(s, 50, -1e5f + 81.6f, 81.6f, 1e5f, {0, 0, 0}, {.75f, .75f, .75f}, 0);
    add(s, 50, 681.6f - .27f, 81.6f, 600, {1, 1, 1}, {0, 0, 0}, 0);
    while (s.geom.size() < n) {
        const float r = 0.5f + 2 * u(rng);
        const float cx = 5 + 90 * u(rng), cy = 3 + 70 * u(rng), cz = 10 + 140 * u(rng);
        const V col{.25f + .7f * u(rng), .25f + .7f * u(rng), .25f + .7f * u(rng)};
        const float t = u(rng);
        add(s, cx, cy, cz, r, {0, 0, 0}, col, t < .70f ? 0 : (t < .85f ? 1 : 2));
    }
    return s;
}

// one closest-hit query as the kernel's walk sees it: tests per visited cell (the last cell's STEP body decides to stop)
struct Query { uint32_t first; uint16_t ncells; uint8_t cls; uint8_t ends; };   // cls: material class of the hit (0 DIFF 1 SPEC 2 REFR 3 none); ends: the segment ends after this bounce
struct Traces {
    std::vector<uint8_t> cells;        // tests per visited cell, all queries back to back
    std::vector<Query> queries;        // all bounces, segment by segment
    std::vector<uint32_t> seg_first;   // first query of every segment (+ one past the end)
    std::vector<uint8_t> seg_camera;   // 1: the segment starts with a camera ray, 0: a popped glass child
};

struct HitRec { uint32_t key, idx; };

HitRec walk(const Scene& s, const spt::SphereGrid& g, V o, V d, Traces& T, uint32_t& ncells, unsigned long long st[3])
{
    HitRec h{kInfKey, 0};
    auto consider = [&](uint32_t i) {
        const uint32_t k = sphere_key(s.geom[i], o, d);
        if (k < h.key || (k == h.key && i < h.idx)) { h.key = k; h.idx = i; }
    };
    for (uint32_t i : g.always) consider(i);
    spt::GridWalk w;
    spt::grid_walk_begin(g.P, o.x, o.y, o.z, d.x, d.y, d.z, w);
    ncells = 0;
    for (;;) {
        const uint32_t hd = g.cells[w.ci];
        if (hd == spt::kGridBorder) break;                           // (the kernel learns this inside the previous cell's STEP body)
        const uint32_t f = hd >> spt::kGridCountBits, c = hd & ((1u << spt::kGridCountBits) - 1u);
        for (uint32_t k = 0; k < c; ++k) consider(g.refs[f + k]);
        T.cells.push_back((uint8_t)std::min(c, 255u)); ++ncells; st[1] += c;
        const float m = spt::grid_walk_exit(w);
        const float near_t = u2f(h.key + kEpsBias);
        if (!(m < near_t)) break;
        spt::grid_walk_step(w.tx, w.ty, w.tz, w.dtx, w.dty, w.dtz, w.sx, w.sy, w.sz, w.ci, m);
        ++st[0];
    }
    ++st[2];
    return h;
}

void make_traces(const Scene& s, const spt::SphereGrid& g, uint32_t nsamples, std::mt19937& rng, Traces& T)
{
    std::uniform_real_distribution<float> u(0.f, 1.f);
    unsigned long long st[3] = {0, 0, 0};
    const int W = 1024, H = 768;
    const V cam_o{50, 52, 295.6f}, cam_d = norm(V{0, -0.042612f, -1});
    const V cx{W * .5135f / H, 0, 0}, cy = norm(cross(cx, cam_d)) * .5135f;
    struct Pend { V o, d, w; int depth; };
    for (uint32_t sidx = 0; sidx < nsamples; ++sidx) {
        // samples of one pixel block are consecutive in a task: keep some coherence by drawing pixels in runs of 32 samples
        static int px = 0, py = 0;
        if (sidx % 32 == 0) { px = (int)(u(rng) * W); py = (int)(u(rng) * H); }
        std::vector<Pend> stack;
        const V dd = cx * (((float)px + u(rng)) / W - .5f) + cy * (((float)py + u(rng)) / H - .5f) + cam_d;
        stack.push_back({cam_o + dd * 140.f, norm(dd), {1, 1, 1}, 0});
        bool camera = true;
        while (!stack.empty()) {
            Pend p = stack.back(); stack.pop_back();
            T.seg_first.push_back((uint32_t)T.queries.size());
            T.seg_camera.push_back(camera ? 1 : 0);
            camera = false;
            for (;;) {
                uint32_t nc = 0;
                const uint32_t first = (uint32_t)T.cells.size();
                const HitRec h = walk(s, g, p.o, p.d, T, nc, st);
                Query q{first, (uint16_t)nc, 3, 1};
                if (h.key == kInfKey) { T.queries.push_back(q); break; }
                const float t = u2f(h.key + kEpsBias);
                const int refl = s.refl[h.idx];
                q.cls = (uint8_t)refl;
                const V x = p.o + p.d * t, n = norm(V{x.x - s.geom[h.idx].x, x.y - s.geom[h.idx].y, x.z - s.geom[h.idx].z});
                const V nl = dot(n, p.d) < 0 ? n : n * -1.f;
                V f = s.color[h.idx];
                const float pm = std::max(f.x, std::max(f.y, f.z));
                bool cont = true;
                if (p.depth > 5) { if (u(rng) < pm) f = f * (1.f / pm); else cont = false; }
                if (cont && p.depth + 1 >= 4096) cont = false;
                V no = x + nl * 0.02f, nd{0, 0, 1}, nf = f;
                if (cont) {
                    if (refl == 0) {
                        const float r1 = 6.2831853f * u(rng), r2 = u(rng), r2s = std::sqrt(r2);
                        const V w = nl, uu = norm(cross(std::fabs(w.x) > .1f ? V{0, 1, 0} : V{1, 0, 0}, w)), vv = cross(w, uu);
                        nd = norm(uu * (std::cos(r1) * r2s) + vv * (std::sin(r1) * r2s) + w * std::sqrt(1 - r2));
                    } else {
                        nd = p.d - n * (2.f * dot(n, p.d));
                        if (refl == 2) {
                            const bool into = dot(n, nl) > 0;
                            const float nnt = into ? 1.f / 1.5f : 1.5f, ddn = dot(p.d, nl), cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
                            if (!(cos2t < 0)) {
                                const V tdir = norm(p.d * nnt - n * ((into ? 1.f : -1.f) * (ddn * nnt + std::sqrt(cos2t))));
                                const float R0 = 0.04f, c = 1 - (into ? -ddn : dot(tdir, n)), Re = R0 + (1 - R0) * c * c * c * c * c, Tr = 1 - Re;
                                const V xin = x - nl * 0.02f;
                                if (p.depth <= 2) {
                                    stack.push_back({xin, tdir, V{p.w.x * f.x * Tr, p.w.y * f.y * Tr, p.w.z * f.z * Tr}, p.depth + 1});
                                    nf = f * Re;
                                } else {
                                    const float P = .25f + .5f * Re;
                                    if (u(rng) < P) nf = f * (Re / P); else { nf = f * (Tr / (1 - P)); no = xin; nd = tdir; }
                                }
                            }
                        }
                    }
                    p.w = V{p.w.x * nf.x, p.w.y * nf.y, p.w.z * nf.z};
                    if (p.w.x == 0.f && p.w.y == 0.f && p.w.z == 0.f) cont = false;
                }
                q.ends = cont ? 0 : 1;
                T.queries.push_back(q);
                if (!cont) break;
                p.o = no; p.d = nd; ++p.depth;
            }
        }
    }
    T.seg_first.push_back((uint32_t)T.queries.size());
    std::printf("traces: %u samples, %zu segments, %zu bounces (%.3f per sample), steps/ray %.2f, tests/ray %.2f, cells/ray %.2f\n", nsamples,
                T.seg_camera.size(), T.queries.size(), (double)T.queries.size() / nsamples, (double)st[0] / st[2], (double)st[1] / st[2], (double)T.cells.size() / st[2]);
}

// ---- cost model: VALU instructions per execution of a body by a wave (whatever the number of active lanes) ----
struct Cost {
    double gen = 95, pop = 12, begin = 85, always_test = 31, test = 33, step = 24, vote = 5, shade_common = 75, diff = 115, spec = 12, refr = 110,
           fetch = 25;
    double exchange = 70; double pool_pop = 8, pool_push = 14, st_load = 6, st_store = 6;   // pool designs: list pop, push (3 ballots + ranks), path state load / store
};

struct Result { double instr = 0, useful = 0; unsigned long long bounces = 0; double walk_instr = 0, walk_useful = 0; };

// a source of segments shared by the slots of the simulated wave
struct Feed {
    const Traces& T; size_t next = 0;
    explicit Feed(const Traces& t) : T(t) {}
    bool take(uint32_t& seg) { if (next >= T.seg_camera.size()) return false; seg = (uint32_t)next++; return true; }
};

// ---- design A: the round-3 kernel.  A lane owns its path; phases regenerate -> begin -> walk (majority vote TEST / STEP,
// left when fewer lanes walk than wait) -> shade. ----
Result sim_current(const Traces& T, const Cost& C, uint32_t nalways, uint32_t leave_q, bool fused)
{
    Result R;
    Feed F(T);
    struct Lane { int mode = 0; uint32_t seg = 0, q = 0, qend = 0, cell = 0, left = 0; bool has = false; };
    Lane L[64];
    bool dry = false;
    for (;;) {
        // A: regeneration
        int ngen = 0, npop = 0;
        for (auto& l : L) if (l.mode == 0) {
            if (l.has && l.q < l.qend) { l.mode = 1; continue; }     // (not used: a lane continues its segment in D)
            uint32_t seg;
            if (!dry && F.take(seg)) { l.seg = seg; l.q = T.seg_first[seg]; l.qend = T.seg_first[seg + 1]; l.has = true; l.mode = 1; if (T.seg_camera[seg]) ++ngen; else ++npop; }
            else dry = true;
        }
        if (ngen) { R.instr += C.gen + C.fetch * 0.05; R.useful += ngen * C.gen / 64; }
        if (npop) { R.instr += C.pop; R.useful += npop * C.pop / 64; }
        int alive = 0;
        for (auto& l : L) alive += l.mode != 0;
        if (!alive) break;
        // B: begin
        int nfresh = 0;
        for (auto& l : L) nfresh += l.mode == 1;
        if (nfresh) {
            const double c = C.begin + nalways * C.always_test;
            R.instr += c; R.useful += nfresh * c / 64;
            for (auto& l : L) if (l.mode == 1) { l.mode = 2; l.cell = 0; l.left = T.cells[T.queries[l.q].first]; R.bounces++; }
        }
        // C: walk
        int nwalk = 0, nidle = 0;
        for (auto& l : L) { nwalk += l.mode == 2; nidle += l.mode == 3; }
        while (nwalk != 0 && (uint32_t)nwalk * 16u >= (uint32_t)nidle * leave_q) {
            int nt = 0;
            for (auto& l : L) nt += l.mode == 2 && l.left > 0;
            auto do_test = [&]() { for (auto& l : L) if (l.mode == 2 && l.left > 0) --l.left; };
            auto do_step = [&](bool only_ready) {
                for (auto& l : L) if (l.mode == 2 && l.left == 0 && only_ready) {
                    const Query& q = T.queries[l.q];
                    if (l.cell + 1 >= q.ncells) { l.mode = 3; }
                    else { ++l.cell; l.left = T.cells[q.first + l.cell]; }
                }
            };
            if (fused) {
                // every iteration: lanes at the end of their cell step, then every lane with a reference tests it
                int ns = nwalk - nt;
                R.instr += C.step + C.test + 2; R.useful += (ns * C.step) / 64; R.walk_instr += C.step + C.test + 2; R.walk_useful += ns * C.step / 64;
                do_step(true);
                int nt2 = 0;
                for (auto& l : L) nt2 += l.mode == 2 && l.left > 0;
                R.useful += nt2 * C.test / 64; R.walk_useful += nt2 * C.test / 64;
                do_test();
            } else if (2 * nt >= nwalk) { R.instr += C.test + C.vote; R.useful += nt * C.test / 64; R.walk_instr += C.test + C.vote; R.walk_useful += nt * C.test / 64; do_test(); }
            else { R.instr += C.step + C.vote; R.useful += (nwalk - nt) * C.step / 64; R.walk_instr += C.step + C.vote; R.walk_useful += (nwalk - nt) * C.step / 64; do_step(true); }
            nwalk = 0; nidle = 0;
            for (auto& l : L) { nwalk += l.mode == 2; nidle += l.mode == 3; }
        }
        // D: shade
        int nc[4] = {0, 0, 0, 0}, nh = 0;
        for (auto& l : L) if (l.mode == 3) { ++nc[T.queries[l.q].cls]; ++nh; }
        if (nh) {
            double c = C.shade_common;
            R.useful += (nh - nc[3]) * C.shade_common / 64;
            if (nc[0]) { c += C.diff; R.useful += nc[0] * C.diff / 64; }
            if (nc[1] || nc[2]) { c += C.spec; R.useful += (nc[1] + nc[2]) * C.spec / 64; }
            if (nc[2]) { c += C.refr; R.useful += nc[2] * C.refr / 64; }
            R.instr += c;
            for (auto& l : L) if (l.mode == 3) {
                const bool ends = T.queries[l.q].ends;
                ++l.q;
                if (ends || l.q >= l.qend) { l.mode = 0; l.has = false; } else l.mode = 1;
            }
        }
    }
    return R;
}

// ---- design P: wave-private pool of S path slots in LDS.  Classes GEN and HIT are run in batches of up to 64 with every lane active and
// followed, in the same lanes, by the begin of the new ray (ray test, always-tested spheres, walk set-up); the begun walk waits in READY.
// The 64 lanes of the wave are walkers: a lane holds one walk in registers, runs the fused STEP+TEST body, and when `drain` lanes have
// finished (or none is walking) the finished lanes hand their hits to HIT and take new walks from READY. ----
Result sim_pool(const Traces& T, const Cost& C, uint32_t nalways, int S, int drain, bool fused, bool split_refr, int min_batch)
{
    Result R;
    Feed F(T);
    struct Slot { uint32_t q = 0, qend = 0; bool has = false; };
    std::vector<Slot> slots(S);
    std::vector<int> gen, hit, ready;            // lists of slot ids
    for (int i = 0; i < S; ++i) gen.push_back(i);
    struct Walker { int slot = -1; uint32_t cell = 0, left = 0; bool done = false; };
    Walker Wk[64];
    bool dry = false;
    auto begin_cost = [&](int lanes) { const double c = C.begin + nalways * C.always_test + C.st_store + C.pool_push; R.instr += c; R.useful += lanes * (C.begin + nalways * C.always_test) / 64; R.bounces += lanes; };
    for (;;) {
        // ---- walkers ----
        int active = 0, done = 0;
        for (auto& w : Wk) { active += w.slot >= 0 && !w.done; done += w.slot >= 0 && w.done; }
        int freel = 0;
        for (auto& w : Wk) freel += w.slot < 0;
        // refill / drain
        if (done >= drain || (active == 0 && done > 0) || (freel > 0 && !ready.empty() && (freel >= drain || active == 0))) {
            R.instr += C.pool_push + C.pool_pop + C.st_load + C.st_store;
            for (auto& w : Wk) if (w.slot >= 0 && w.done) { hit.push_back(w.slot); w.slot = -1; w.done = false; }
            for (auto& w : Wk) if (w.slot < 0 && !ready.empty()) {
                w.slot = ready.back(); ready.pop_back();
                const Query& q = T.queries[slots[w.slot].q];
                w.cell = 0; w.left = T.cells[q.first]; w.done = false;
            }
            continue;
        }
        // is there production work worth doing before walking on?  (a full batch, or the walkers are starving)
        const bool starving = (int)ready.size() < drain && active < 64 - drain + 1;
        const int need = starving ? min_batch : 64;
        auto run_hit = [&]() {
            const int b = std::min<int>(64, (int)hit.size());
            int nc[4] = {0, 0, 0, 0};
            std::vector<int> batch(hit.end() - b, hit.end()); hit.resize(hit.size() - b);
            for (int s : batch) ++nc[T.queries[slots[s].q].cls];
            double c = C.shade_common + C.pool_pop + C.st_load;
            R.useful += (b - nc[3]) * C.shade_common / 64;
            if (nc[0]) { c += C.diff; R.useful += nc[0] * C.diff / 64; }
            if (nc[1] || nc[2]) { c += C.spec; R.useful += (nc[1] + nc[2]) * C.spec / 64; }
            if (nc[2]) { c += C.refr; R.useful += nc[2] * C.refr / 64; }
            R.instr += c;
            int cont = 0;
            for (int s : batch) {
                const bool ends = T.queries[slots[s].q].ends;
                ++slots[s].q;
                if (ends || slots[s].q >= slots[s].qend) { slots[s].has = false; gen.push_back(s); } else { ready.push_back(s); ++cont; }
            }
            if (cont) begin_cost(cont); else R.instr += C.pool_push;
        };
        auto run_gen = [&]() {
            const int b = std::min<int>(64, (int)gen.size());
            std::vector<int> batch(gen.end() - b, gen.end()); gen.resize(gen.size() - b);
            int ngen = 0, npop = 0;
            for (int s : batch) {
                uint32_t seg;
                if (!dry && F.take(seg)) { slots[s].q = T.seg_first[seg]; slots[s].qend = T.seg_first[seg + 1]; slots[s].has = true; ready.push_back(s); if (T.seg_camera[seg]) ++ngen; else ++npop; }
                else dry = true;                 // the slot retires
            }
            double c = C.pool_pop + C.st_load + C.fetch * 0.05;
            if (ngen) { c += C.gen; R.useful += ngen * C.gen / 64; }
            if (npop) { c += C.pop; R.useful += npop * C.pop / 64; }
            R.instr += c;
            if (ngen + npop) begin_cost(ngen + npop);
        };
        (void)split_refr;
        if ((int)hit.size() >= need && hit.size() >= gen.size()) { run_hit(); continue; }
        if ((int)gen.size() >= need && !dry) { run_gen(); continue; }
        if ((int)hit.size() >= need) { run_hit(); continue; }
        if (active > 0) {
            // one walk iteration
            int nt = 0, ns = 0;
            for (auto& w : Wk) if (w.slot >= 0 && !w.done) { if (w.left > 0) ++nt; else ++ns; }
            auto step_lane = [&](Walker& w) {
                const Query& q = T.queries[slots[w.slot].q];
                if (w.cell + 1 >= q.ncells) w.done = true; else { ++w.cell; w.left = T.cells[q.first + w.cell]; }
            };
            if (fused) {
                R.instr += C.step + C.test + 3; R.useful += ns * C.step / 64; R.walk_instr += C.step + C.test + 3; R.walk_useful += ns * C.step / 64;
                for (auto& w : Wk) if (w.slot >= 0 && !w.done && w.left == 0) step_lane(w);
                int nt2 = 0;
                for (auto& w : Wk) if (w.slot >= 0 && !w.done && w.left > 0) { --w.left; ++nt2; }
                R.useful += nt2 * C.test / 64; R.walk_useful += nt2 * C.test / 64;
            } else if (2 * nt >= nt + ns) {
                R.instr += C.test + C.vote; R.useful += nt * C.test / 64; R.walk_instr += C.test + C.vote; R.walk_useful += nt * C.test / 64;
                for (auto& w : Wk) if (w.slot >= 0 && !w.done && w.left > 0) --w.left;
            } else {
                R.instr += C.step + C.vote; R.useful += ns * C.step / 64; R.walk_instr += C.step + C.vote; R.walk_useful += ns * C.step / 64;
                for (auto& w : Wk) if (w.slot >= 0 && !w.done && w.left == 0) step_lane(w);
            }
            continue;
        }
        // nothing walks and nothing is ready: run whatever is there, however small
        if (!hit.empty()) { run_hit(); continue; }
        if (!gen.empty() && !dry) { run_gen(); continue; }
        break;
    }
    return R;
}

// ---- design Q: design P with (1) ray buffers separate from slots -- a walker holds its walk in registers and no buffer, so B buffers
// serve B + 64 paths --, (2) glass hits in a class of their own, (3) the roulette inside the shading batch (its losers idle through the
// class code and the begin). ----
Result sim_pool2(const Traces& T, const Cost& C, uint32_t nalways, int S, int B, int drain, int min_batch, bool verbose = false, int ksteps = 1, bool hit_global = false)
{
    Result R;
    Feed F(T);
    struct Slot { uint32_t q = 0, qend = 0; };
    std::vector<Slot> slots(S);
    std::vector<int> gen, hit, hitr, ready;      // gen: slot ids; the others: slot ids too (each stands for one buffer)
    int nfree = B;
    for (int i = 0; i < S; ++i) gen.push_back(i);
    struct Walker { int slot = -1; uint32_t cell = 0, left = 0; bool done = false; };
    Walker Wk[64];
    bool dry = false;
    unsigned long long n_batches[3] = {0, 0, 0}, n_lanes[3] = {0, 0, 0}, n_iter = 0, n_act = 0, n_exch = 0;
    auto begin_cost = [&](int lanes) { const double c = C.begin + nalways * C.always_test + C.st_store + C.pool_push; R.instr += c; R.useful += lanes * (C.begin + nalways * C.always_test) / 64; R.bounces += lanes; };
    auto run_hit = [&](std::vector<int>& L, bool refr) {
        const int b = hit_global ? std::min<int>(std::min<int>(64, (int)L.size()), nfree) : std::min<int>(64, (int)L.size());
        std::vector<int> batch(L.end() - b, L.end()); L.resize(L.size() - b);
        n_batches[refr ? 2 : 1]++; n_lanes[refr ? 2 : 1] += b;
        int nc[4] = {0, 0, 0, 0}, cont = 0;
        std::vector<int> alive;
        for (int s : batch) { const Query& q = T.queries[slots[s].q]; if (!(q.ends || slots[s].q + 1 >= slots[s].qend)) { ++nc[q.cls]; alive.push_back(s); } else { ++nc[3]; } }
        double c = C.shade_common + C.pool_pop + C.st_load + 8;            // + cold state from global memory
        R.useful += b * C.shade_common / 64;
        if (nc[0]) { c += C.diff; R.useful += nc[0] * C.diff / 64; }
        if (nc[1] || nc[2]) { c += C.spec; R.useful += (nc[1] + nc[2]) * C.spec / 64; }
        if (nc[2]) { c += C.refr; R.useful += nc[2] * C.refr / 64; }
        R.instr += c;
        for (int s : batch) {
            const bool ends = T.queries[slots[s].q].ends;
            ++slots[s].q;
            if (ends || slots[s].q >= slots[s].qend) { gen.push_back(s); if (!hit_global) ++nfree; } else { ready.push_back(s); ++cont; if (hit_global) --nfree; }
        }
        if (cont) begin_cost(cont); else R.instr += C.pool_push;
    };
    auto run_gen = [&]() {
        const int b = std::min<int>(std::min<int>(64, (int)gen.size()), nfree);
        std::vector<int> batch(gen.end() - b, gen.end()); gen.resize(gen.size() - b);
        n_batches[0]++; n_lanes[0] += b;
        int ngen = 0, npop = 0;
        for (int s : batch) {
            uint32_t seg;
            if (!dry && F.take(seg)) { slots[s].q = T.seg_first[seg]; slots[s].qend = T.seg_first[seg + 1]; ready.push_back(s); --nfree; if (T.seg_camera[seg]) ++ngen; else ++npop; }
            else dry = true;
        }
        double c = C.pool_pop + C.st_load + C.fetch * 0.05 + 8;
        if (ngen) { c += C.gen; R.useful += ngen * C.gen / 64; }
        if (npop) { c += C.pop; R.useful += npop * C.pop / 64; }
        R.instr += c;
        if (ngen + npop) begin_cost(ngen + npop);
    };
    for (;;) {
        int active = 0, done = 0, freel = 0;
        for (auto& w : Wk) { active += w.slot >= 0 && !w.done; done += w.slot >= 0 && w.done; freel += w.slot < 0; }
        // exchange: finished walkers hand their hit over (into the buffer of the READY walk they take, else into a free buffer); empty lanes take READY walks
        const int can_flush = hit_global ? done : std::min<int>(done, (int)ready.size() + nfree);
        const int can_fill = std::min<int>(freel + done, (int)ready.size());
        if ((can_flush + can_fill > 0) && (done + freel >= drain || active == 0) && (can_flush >= std::min(done, drain) || can_fill >= std::min(drain, freel + done) || active == 0)) {
            R.instr += C.exchange;
            ++n_exch;
            for (auto& w : Wk) if (w.slot >= 0 && w.done) {
                if (hit_global) {
                    const int cls = T.queries[slots[w.slot].q].cls;
                    (cls == 2 ? hitr : hit).push_back(w.slot);
                    w.slot = -1; w.done = false;
                } else if (!ready.empty()) {
                    const int cls = T.queries[slots[w.slot].q].cls;
                    (cls == 2 ? hitr : hit).push_back(w.slot);
                    w.slot = ready.back(); ready.pop_back();
                    const Query& q = T.queries[slots[w.slot].q];
                    w.cell = 0; w.left = T.cells[q.first]; w.done = false;
                } else if (nfree > 0) {
                    --nfree;
                    const int cls = T.queries[slots[w.slot].q].cls;
                    (cls == 2 ? hitr : hit).push_back(w.slot);
                    w.slot = -1; w.done = false;
                }
            }
            for (auto& w : Wk) if (w.slot < 0 && !ready.empty()) {
                w.slot = ready.back(); ready.pop_back(); ++nfree;
                const Query& q = T.queries[slots[w.slot].q];
                w.cell = 0; w.left = T.cells[q.first]; w.done = false;
            }
            continue;
        }
        const bool starving = (int)ready.size() < drain && active <= 64 - drain;
        const int need = starving ? min_batch : 64;
        const int gen_avail = dry ? 0 : std::min<int>((int)gen.size(), nfree);
        const int hit_avail = hit_global ? std::min<int>((int)hit.size(), nfree) : (int)hit.size();
        const int hitr_avail = hit_global ? std::min<int>((int)hitr.size(), nfree) : (int)hitr.size();
        if (hitr_avail >= need && hitr_avail >= hit_avail && hitr_avail >= gen_avail) { run_hit(hitr, true); continue; }
        if (hit_avail >= need && hit_avail >= gen_avail) { run_hit(hit, false); continue; }
        if (gen_avail >= need) { run_gen(); continue; }
        if (hit_avail >= need) { run_hit(hit, false); continue; }
        if (hitr_avail >= need) { run_hit(hitr, true); continue; }
        if (active > 0) {
            int ns = 0;
            for (auto& w : Wk) if (w.slot >= 0 && !w.done && w.left == 0) ++ns;
            ++n_iter; n_act += active;
            R.instr += ksteps * C.step + C.test + 3; R.walk_instr += ksteps * C.step + C.test + 3;
            for (int k = 0; k < ksteps; ++k)
                for (auto& w : Wk) if (w.slot >= 0 && !w.done && w.left == 0) {
                    const Query& q = T.queries[slots[w.slot].q];
                    R.useful += C.step / 64; R.walk_useful += C.step / 64;
                    if (w.cell + 1 >= q.ncells) w.done = true; else { ++w.cell; w.left = T.cells[q.first + w.cell]; }
                }
            (void)ns;
            int nt2 = 0;
            for (auto& w : Wk) if (w.slot >= 0 && !w.done && w.left > 0) { --w.left; ++nt2; }
            R.useful += nt2 * C.test / 64; R.walk_useful += nt2 * C.test / 64;
            continue;
        }
        // nothing walks: run whatever is there, however small
        if (!hit.empty() && hit_avail > 0) { run_hit(hit, false); continue; }
        if (!hitr.empty() && hitr_avail > 0) { run_hit(hitr, true); continue; }
        if (gen_avail > 0) { run_gen(); continue; }
        if (done > 0) { std::printf("stuck: %d done walkers, no buffer\n", done); break; }
        break;
    }
    if (verbose)
        std::printf("      batches: GEN %llu x %.1f  HIT %llu x %.1f  REFR %llu x %.1f ; walk iterations %llu x %.1f active lanes; exchanges %llu\n", n_batches[0], (double)n_lanes[0] / std::max(1ull, n_batches[0]),
                    n_batches[1], (double)n_lanes[1] / std::max(1ull, n_batches[1]), n_batches[2], (double)n_lanes[2] / std::max(1ull, n_batches[2]), n_iter, (double)n_act / std::max(1ull, n_iter), n_exch);
    return R;
}

void report(const char* name, const Result& r)
{
    const double wb = (double)r.bounces / 64;
    std::printf("%-58s %6.0f VALU instr / 64 lane-bounces  util %.3f | walk %5.0f util %.3f | rest %5.0f util %.3f\n", name, r.instr / wb, r.useful / r.instr,
                r.walk_instr / wb, r.walk_useful / r.walk_instr, (r.instr - r.walk_instr) / wb, (r.useful - r.walk_useful) / (r.instr - r.walk_instr));
}

}  // namespace

int main(int argc, char** argv)
{
    const uint32_t nsamples = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 60000;
    std::mt19937 rng(1024);
    const Scene s = config5_like(1024, rng);
    spt::SphereGrid g;
    spt::build_sphere_grid(s.geom.data(), s.radius.data(), (uint32_t)s.geom.size(), argc > 2 ? std::atof(argv[2]) : 4.0, (size_t)150 * 1024 - s.geom.size() * 16, g);
    if (!g.usable) { std::printf("grid unusable: %s\n", g.why.c_str()); return 1; }
    std::printf("grid %d x %d x %d, %zu references, %zu always-tested\n", g.P.dim[0], g.P.dim[1], g.P.dim[2], g.refs.size(), g.always.size());
    Traces T;
    make_traces(s, g, nsamples, rng, T);
    const Cost C;
    const uint32_t na = (uint32_t)g.always.size();
    report("A  round 3 (lane owns path, vote, leave when nwalk < nidle)", sim_current(T, C, na, 16, false));
    report("A' round 3 with every walk run to its end", sim_current(T, C, na, 0, false));
    report("A2 round 3 phases with a fused STEP+TEST body", sim_current(T, C, na, 16, true));
    for (int S : {96, 128, 160, 192, 256})
        for (int drain : {8, 16, 24}) {
            char nm[128];
            std::snprintf(nm, sizeof nm, "P  pool S=%d drain=%d fused", S, drain);
            report(nm, sim_pool(T, C, na, S, drain, true, false, 32));
            std::snprintf(nm, sizeof nm, "P  pool S=%d drain=%d vote", S, drain);
            report(nm, sim_pool(T, C, na, S, drain, false, false, 32));
        }
    for (double ex : {40.0, 70.0, 100.0})
        for (int drain : {8, 16, 24, 32, 40}) {
            Cost C2 = C; C2.exchange = ex;
            char nm[128];
            std::snprintf(nm, sizeof nm, "Q2 S=256 READY=80 drain=%d exchange cost %.0f", drain, ex);
            report(nm, sim_pool2(T, C2, na, 256, 80, drain, 32, false, 1, true));
        }
    return 0;
}
