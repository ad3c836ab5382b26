// ASAN/UBSAN run of the host builders on random and degenerate triangle soups
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "host/sbvh.hpp"
#include "host/cwbvh.hpp"
static uint32_t rs = 4242;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
static float rf() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
int main(int argc, char** argv) {
    const int iters = atoi(argv[1]);
    size_t built = 0;
    for (int it = 0; it < iters; ++it) {
        const int kind = rnd() % 6;
        const size_t nt = kind == 5 ? 1 + rnd() % 3 : 1 + rnd() % 400;
        std::vector<crt::float3> v;
        std::vector<crt_triangle> t(nt);
        for (size_t i = 0; i < nt; ++i) {
            crt::float3 c{rf() * 10, rf() * 10, rf() * 10};
            for (int k = 0; k < 3; ++k) {
                crt::float3 p = c;
                const float s = kind == 1 ? 0.f : kind == 2 ? 5.f : 0.3f;           // degenerate points / huge overlapping / small
                p.x += s * (rf() - 0.5f); p.y += s * (rf() - 0.5f); p.z += (kind == 3 ? 0.f : s * (rf() - 0.5f));   // kind 3: coplanar
                if (kind == 4) p = crt::float3{1.f, 2.f, 3.f};                        // all coincident
                v.push_back(p);
            }
            std::memset(&t[i], 0, sizeof t[i]);
            int32_t* w = reinterpret_cast<int32_t*>(&t[i]);
            w[0] = (int32_t)(3 * i); w[1] = (int32_t)(3 * i + 1); w[2] = (int32_t)(3 * i + 2); w[3] = 0;
        }
        for (uint32_t flags : {0u, (uint32_t)crt::SBVH::NO_SPATIAL_SPLITS}) {
            crt::SBVH b(t, v, flags);
            if (b.flat_nodes.empty()) { printf("empty tree\n"); return 1; }
            crt::CWBVH c;
            if (!c.convert(b)) { printf("convert refused: %s (kind %d nt %zu depth %d)\n", c.error.c_str(), kind, nt, b.depth); continue; }
            ++built;
        }
    }
    printf("built %zu trees\n", built);
}
