// ASan/UBSan driver of the hierarchy builder (optix-test-smallpt_amd/csrc/spt_bvh.cpp): random triangle soups, coincident and
// collinear triangles, sizes around the leaf capacity; every build is validated structurally.
#include <cmath>
#include <cstdio>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../optix-test-smallpt_amd/csrc/spt_bvh.h"

static void record(std::vector<float4>& recs, const float v[3][3])
{
    const float e1[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]}, e2[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
    const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    recs.push_back(make_float4(v[0][0], v[0][1], v[0][2], n[0]));
    recs.push_back(make_float4(e1[0], e1[1], e1[2], n[1]));
    recs.push_back(make_float4(e2[0], e2[1], e2[2], n[2]));
}

int main()
{
    std::mt19937 rng(17);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    int builds = 0;
    for (int n : {0, 1, 2, 4, 5, 8, 9, 33, 257, 4097, 30000}) {
        for (int kind = 0; kind < 4; ++kind) {
            std::vector<float4> recs;
            for (int i = 0; i < n; ++i) {
                float v[3][3];
                const float scale = kind == 1 ? 0.f : std::pow(10.f, 2.f * u(rng));
                const float c[3] = {kind == 2 ? 0.f : 100.f * u(rng), kind == 3 ? 5.f : 100.f * u(rng), 100.f * u(rng)};
                for (auto& p : v) for (int a = 0; a < 3; ++a) p[a] = c[a] + scale * u(rng);   // kind 1: every triangle a point at c
                record(recs, v);
            }
            spt::Bvh bvh;
            spt::build_bvh(recs.data(), (uint32_t)n, bvh);
            std::string why;
            if (!spt::validate_bvh(recs.data(), (uint32_t)n, bvh, why)) { std::printf("invalid hierarchy n=%d kind=%d: %s\n", n, kind, why.c_str()); return 1; }
            ++builds;
        }
    }
    std::vector<float4> bad(3, make_float4(NAN, 0.f, 0.f, 0.f));
    try {
        spt::Bvh bvh;
        spt::build_bvh(bad.data(), 1, bvh);
        std::printf("non-finite input accepted\n");
        return 1;
    } catch (const std::runtime_error&) {
    }
    std::printf("bvh sanitizer run ok %d\n", builds);
    return 0;
}
