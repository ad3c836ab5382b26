#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "host/obj_loader.hpp"
#ifndef FUZZ_DIR
#define FUZZ_DIR "/tmp/fuzz_obj"
#endif
static uint32_t rs = 777;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
static std::vector<char> slurp(const char* p) { FILE* f = fopen(p, "rb"); std::vector<char> v; char b[65536]; for (size_t k; (k = fread(b, 1, sizeof b, f)) > 0;) v.insert(v.end(), b, b + k); fclose(f); return v; }
static void spit(const char* p, const std::vector<char>& v) { FILE* f = fopen(p, "wb"); fwrite(v.data(), 1, v.size(), f); fclose(f); }
int main(int argc, char** argv) {
    const int iters = atoi(argv[1]);
    std::vector<char> obj = slurp(FUZZ_DIR "/scene.obj"), mtl = slurp(FUZZ_DIR "/scene.mtl");
    size_t ok = 0, bad = 0;
    const char pool[] = "0123456789 -./\nfvnmtlusKdea#\\eE+";
    for (int it = 0; it < iters; ++it) {
        std::vector<char> o = obj, m = mtl;
        std::vector<char>& t = (rnd() % 3 == 0) ? m : o;
        const int muts = 1 + rnd() % 6;
        for (int k = 0; k < muts; ++k) {
            const size_t pos = rnd() % t.size();
            const uint32_t kind = rnd() % 5;
            if (kind == 0) t[pos] = pool[rnd() % (sizeof pool - 1)];
            else if (kind == 1) t.erase(t.begin() + pos, t.begin() + std::min(t.size(), pos + 1 + rnd() % 20));
            else if (kind == 2) t.insert(t.begin() + pos, pool[rnd() % (sizeof pool - 1)]);
            else if (kind == 3) t[pos] = (char)rnd();
            else { const char* big = "99999999999"; t.insert(t.begin() + pos, big, big + 11); }
            if (t.empty()) t.push_back('\n');
        }
        spit(FUZZ_DIR "/w/scene.obj", o); spit(FUZZ_DIR "/w/scene.mtl", m);
        crt::Mesh mesh;
        if (mesh.read_object(FUZZ_DIR "/w/scene.obj")) ++ok; else ++bad;
    }
    printf("loaded %zu, refused %zu\n", ok, bad);
}
