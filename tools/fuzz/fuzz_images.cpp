#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <vector>
#include "host/image.hpp"
static uint32_t rs = 12345;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; return rs; }
int main(int argc, char** argv) {
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    DIR* d = opendir(argv[1]);
    std::vector<std::string> names;
    for (dirent* e; (e = readdir(d));) if (e->d_name[0] != '.') names.push_back(e->d_name);
    closedir(d);
    size_t ok = 0, bad = 0;
    for (const std::string& n : names) {
        std::string path = std::string(argv[1]) + "/" + n;
        FILE* f = fopen(path.c_str(), "rb");
        std::vector<uint8_t> base;
        uint8_t buf[65536];
        for (size_t k; (k = fread(buf, 1, sizeof buf, f)) > 0;) base.insert(base.end(), buf, buf + k);
        fclose(f);
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = base;
            const int muts = 1 + rnd() % 4;
            for (int k = 0; k < muts; ++k) {
                const size_t pos = (rnd() % 3 == 0) ? rnd() % m.size() : rnd() % (m.size() < 700 ? m.size() : 700);
                const uint32_t kind = rnd() % 4;
                m[pos] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? 0xff : kind == 2 ? 0 : (uint8_t)(m[pos] + 1);
            }
            if (rnd() % 10 == 0) m.resize(1 + rnd() % m.size());
            int w = 0, h = 0;
            std::vector<uint8_t> rgb;
            std::string err;
            // keep sizes sane for the fuzzer's memory: dimensions are validated by the decoders against the file size anyway
            if (crt::decode_image_rgb8(m.data(), m.size(), w, h, rgb, err)) { ++ok; if (rgb.size() != (size_t)w * h * 3) { printf("size mismatch %s\n", n.c_str()); return 1; } }
            else ++bad;
        }
    }
    printf("decoded %zu, refused %zu\n", ok, bad);
    return 0;
}
