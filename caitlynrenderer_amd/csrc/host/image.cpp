// See image.hpp.  Decoders written from the formats' specifications; output contract = stbi_load(..., 3):
// 8-bit RGB, top row first, alpha dropped, grey replicated, 16-bit samples reduced to their high byte.
// (JPEG: jpeg.cpp.)  Pinned byte for byte against the reference's own stb_image: tests/golden/stb_decodes.npz.
#include "image.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace crt {
namespace {

struct Reader {
    const uint8_t* p; size_t n, at = 0;
    bool ok = true;
    uint8_t u8() { if (at >= n) { ok = false; return 0; } return p[at++]; }
    uint32_t le16() { uint32_t a = u8(); return a | (u8() << 8); }
    uint32_t le32() { uint32_t a = le16(); return a | (le16() << 16); }
    uint32_t be32() { uint32_t a = u8() << 24; a |= u8() << 16; a |= u8() << 8; return a | u8(); }
    void skip(size_t k) { if (at + k > n) { ok = false; at = n; } else at += k; }
};

bool fail(std::string& error, const char* msg) { error = msg; return false; }
bool sane(int w, int h) { return w > 0 && h > 0 && w <= (1 << 15) && h <= (1 << 15); }

// ---------------------------------------------------------------- PNM (binary P5 / P6)
bool pnm_token(Reader& r, int& v) {
    for (;;) {                                          // white space and '#' comments
        if (r.at >= r.n) return false;
        const uint8_t c = r.p[r.at];
        if (c == '#') { while (r.at < r.n && r.p[r.at] != '\n' && r.p[r.at] != '\r') ++r.at; }
        else if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') ++r.at;
        else break;
    }
    if (r.p[r.at] < '0' || r.p[r.at] > '9') return false;
    long long x = 0;
    while (r.at < r.n && r.p[r.at] >= '0' && r.p[r.at] <= '9') { x = x * 10 + (r.p[r.at++] - '0'); if (x > (1 << 30)) return false; }
    v = (int)x;
    return true;
}
bool decode_pnm(const uint8_t* b, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error) {
    Reader r{b, n};
    r.at = 2;
    const int comp = b[1] == '5' ? 1 : 3;
    int maxv = 0;
    if (!pnm_token(r, w) || !pnm_token(r, h) || !pnm_token(r, maxv)) return fail(error, "pnm: bad header");
    if (!sane(w, h) || maxv < 1) return fail(error, "pnm: bad size or maxval");
    if (maxv > 255) return fail(error, "pnm: more than 8 bits per sample (the reference's stb_image refuses these too)");
    if (r.at >= n) return fail(error, "pnm: truncated");
    ++r.at;                                             // the single white-space byte after maxval
    const size_t need = (size_t)w * h * comp;
    if (r.at + need > n) return fail(error, "pnm: truncated");
    rgb.resize((size_t)w * h * 3);
    const uint8_t* s = b + r.at;
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int c = 0; c < 3; ++c) rgb[3 * i + c] = s[i * comp + (comp == 1 ? 0 : c)];
    return true;
}

// ---------------------------------------------------------------- BMP (uncompressed 8 / 24 / 32 bpp)
bool decode_bmp(const uint8_t* b, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error) {
    Reader r{b, n};
    r.skip(10);
    const uint32_t data_off = r.le32(), hdr = r.le32();
    if (hdr != 40 && hdr != 52 && hdr != 56 && hdr != 108 && hdr != 124) return fail(error, "bmp: unsupported header");
    const int32_t bw = (int32_t)r.le32(), bh = (int32_t)r.le32();
    const uint32_t planes = r.le16();
    const uint32_t bpp = r.le16(), comp = r.le32();
    r.skip(12);
    uint32_t n_colors = r.le32();
    r.le32();
    uint32_t mr = 0x00ff0000u, mg = 0x0000ff00u, mb = 0x000000ffu;
    if (comp == 3) { mr = r.le32(); mg = r.le32(); mb = r.le32(); }      // BI_BITFIELDS: the masks follow the 40 info bytes
    if (!r.ok) return fail(error, "bmp: truncated header");
    if (planes != 1) return fail(error, "bmp: bad plane count");             // as stb_image
    if (comp != 0 && !(comp == 3 && bpp == 32)) return fail(error, "bmp: compressed files are not supported");
    if (comp == 3 && !(mr == 0x00ff0000u && mg == 0x0000ff00u && mb == 0x000000ffu)) return fail(error, "bmp: non-standard bit masks");
    if (bpp != 8 && bpp != 24 && bpp != 32) return fail(error, "bmp: only 8, 24 and 32 bits per pixel");
    w = bw; h = bh < 0 ? -bh : bh;
    if (!sane(w, h)) return fail(error, "bmp: bad size");
    std::vector<uint8_t> pal;
    if (data_off < 14 + (size_t)hdr + (comp == 3 && hdr == 40 ? 12 : 0)) return fail(error, "bmp: pixel data inside the header");
    if (bpp == 8) {
        // stb_image sizes the palette by the room before the pixel data, and refuses a file that leaves none
        const size_t room = ((size_t)data_off - 14 - hdr) / 4;
        if (room == 0 || room > 256) return fail(error, "bmp: bad palette size");
        if (n_colors == 0 || n_colors > room) n_colors = (uint32_t)room;
        const size_t pal_at = 14 + (size_t)hdr;
        if (pal_at + 4 * (size_t)n_colors > n) return fail(error, "bmp: truncated palette");
        pal.assign(b + pal_at, b + pal_at + 4 * (size_t)n_colors);
    }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if (data_off + stride * h > n) return fail(error, "bmp: truncated pixel data");
    rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = b + data_off + stride * (size_t)(bh < 0 ? y : h - 1 - y);    // bottom-up unless height < 0
        uint8_t* d = rgb.data() + (size_t)y * w * 3;
        for (int x = 0; x < w; ++x) {
            if (bpp == 8) {
                const uint32_t k = s[x];
                if (k >= n_colors) return fail(error, "bmp: palette index out of range");
                d[3 * x] = pal[4 * k + 2]; d[3 * x + 1] = pal[4 * k + 1]; d[3 * x + 2] = pal[4 * k];
            } else {
                const uint8_t* px = s + (size_t)x * (bpp / 8);
                d[3 * x] = px[2]; d[3 * x + 1] = px[1]; d[3 * x + 2] = px[0];
            }
        }
    }
    return true;
}

// ---------------------------------------------------------------- TGA (types 1/2/3 and RLE 9/10/11; 8/24/32 bpp)
bool decode_tga(const uint8_t* b, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error) {
    Reader r{b, n};
    const uint32_t id_len = r.u8(), cmap_type = r.u8(), type = r.u8();
    const uint32_t cmap_first = r.le16(), cmap_len = r.le16(), cmap_bits = r.u8();
    r.le16(); r.le16();
    w = (int)r.le16(); h = (int)r.le16();
    const uint32_t bpp = r.u8(), desc = r.u8();
    if (!r.ok || !sane(w, h)) return fail(error, "tga: bad header");
    const bool rle = type >= 8;
    const uint32_t base = type & 7u;
    if (base < 1 || base > 3) return fail(error, "tga: unsupported image type");
    if (cmap_type > 1 || (cmap_type == 1) != (base == 1)) return fail(error, "tga: colour map does not fit the image type");   // as stb_image
    if (base == 1) { if (cmap_type != 1 || bpp != 8 || (cmap_bits != 24 && cmap_bits != 32)) return fail(error, "tga: unsupported colour map"); }
    else if (base == 2) { if (bpp != 24 && bpp != 32) return fail(error, "tga: only 24 and 32 bits per pixel"); }
    else if (bpp != 8) return fail(error, "tga: only 8-bit grey");
    r.skip(id_len);
    std::vector<uint8_t> pal;
    if (cmap_type == 1) {
        const size_t bytes = (size_t)cmap_len * ((cmap_bits + 7) / 8);
        if (r.at + bytes > n) return fail(error, "tga: truncated colour map");
        if (base == 1) pal.assign(b + r.at, b + r.at + bytes);
        r.skip(bytes);
    }
    const uint32_t px_bytes = bpp / 8;
    // bound the allocation by what the file can hold: raw data needs w*h*px_bytes input bytes, and a run-length
    // packet of 1 + px_bytes bytes expands to at most 128 pixels
    {
        const size_t left = n - std::min(n, r.at), want = (size_t)w * h * px_bytes;
        if (rle ? want / 128 > left : want > left) return fail(error, "tga: truncated pixel data");
    }
    std::vector<uint8_t> raw((size_t)w * h * px_bytes);
    if (!rle) {
        if (r.at + raw.size() > n) return fail(error, "tga: truncated pixel data");
        std::memcpy(raw.data(), b + r.at, raw.size());
    } else {
        size_t out = 0;
        while (out < raw.size()) {
            const uint32_t c = r.u8(), cnt = (c & 127u) + 1u;
            if (!r.ok || out + (size_t)cnt * px_bytes > raw.size()) return fail(error, "tga: bad run-length data");
            if (c & 128u) {
                uint8_t v[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < px_bytes; ++k) v[k] = r.u8();
                for (uint32_t i = 0; i < cnt; ++i) for (uint32_t k = 0; k < px_bytes; ++k) raw[out++] = v[k];
            } else {
                for (uint32_t i = 0; i < cnt * px_bytes; ++i) raw[out++] = r.u8();
            }
            if (!r.ok) return fail(error, "tga: truncated run-length data");
        }
    }
    rgb.resize((size_t)w * h * 3);
    const bool top_down = (desc >> 5) & 1u;              // the right-to-left bit (4) is ignored, as stb_image ignores it
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = raw.data() + (size_t)(top_down ? y : h - 1 - y) * w * px_bytes;
        uint8_t* d = rgb.data() + (size_t)y * w * 3;
        for (int x = 0; x < w; ++x) {
            const uint8_t* px = s + (size_t)x * px_bytes;
            if (base == 3) { d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = px[0]; }
            else if (base == 2) { d[3 * x] = px[2]; d[3 * x + 1] = px[1]; d[3 * x + 2] = px[0]; }
            else {
                const uint32_t k = px[0];
                if (k < cmap_first || k - cmap_first >= cmap_len) return fail(error, "tga: colour index out of range");
                const uint8_t* e = pal.data() + (size_t)(k - cmap_first) * (cmap_bits / 8);
                d[3 * x] = e[2]; d[3 * x + 1] = e[1]; d[3 * x + 2] = e[0];
            }
        }
    }
    return true;
}

// ---------------------------------------------------------------- PNG (plain and Adam7-interlaced; tRNS is dropped with the alpha)
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
bool decode_png(const uint8_t* b, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error) {
    Reader r{b, n};
    r.skip(8);
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, pal;
    bool have_ihdr = false, done = false;
    while (!done) {
        const uint32_t len = r.be32(), tag = r.be32();
        if (!r.ok || r.at + (size_t)len + 4 > n) return fail(error, "png: truncated chunk");
        const uint8_t* d = b + r.at;
        if (tag == 0x49484452u) {                        // IHDR
            if (len != 13) return fail(error, "png: bad IHDR");
            Reader q{d, len};
            w = (int)q.be32(); h = (int)q.be32();
            depth = q.u8(); ctype = q.u8();
            if (q.u8() != 0 || q.u8() != 0) return fail(error, "png: bad compression or filter method");
            interlace = q.u8();
            have_ihdr = true;
        } else if (tag == 0x504c5445u) pal.assign(d, d + len);            // PLTE
        else if (tag == 0x49444154u) idat.insert(idat.end(), d, d + len);  // IDAT
        else if (tag == 0x49454e44u) done = true;                          // IEND
        r.skip((size_t)len + 4);                                           // data + CRC (not verified, as in stb_image)
    }
    if (!have_ihdr || !sane(w, h)) return fail(error, "png: bad header");
    if (interlace > 1) return fail(error, "png: bad interlace method");
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels) return fail(error, "png: bad colour type");
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16))
        return fail(error, "png: bad bit depth");
    const size_t bits_pp = (size_t)channels * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    // the passes of the image: one, or Adam7's seven sub-images {x0, y0, dx, dy}, stored one after the other
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int whole[1][4] = {{0, 0, 1, 1}};
    const int (*pass)[4] = interlace ? adam7 : whole;
    const int n_pass = interlace ? 7 : 1;
    size_t raw_need = 0;
    for (int k = 0; k < n_pass; ++k) {
        const size_t pw = ((size_t)w - pass[k][0] + pass[k][2] - 1) / pass[k][2], ph = ((size_t)h - pass[k][1] + pass[k][3] - 1) / pass[k][3];
        if (w > pass[k][0] && h > pass[k][1]) raw_need += ((pw * bits_pp + 7) / 8 + 1) * ph;
    }
    // deflate expands by at most ~1032 : 1, so a header of a few bytes cannot ask for gigabytes
    if (raw_need / 1032 > idat.size() + 1) return fail(error, "png: bad compressed data");
    std::vector<uint8_t> raw(raw_need);
    uLongf raw_len = (uLongf)raw.size();
    const int zrc = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if ((zrc != Z_OK && zrc != Z_BUF_ERROR) || raw_len < raw.size()) return fail(error, "png: bad compressed data");
    rgb.resize((size_t)w * h * 3);
    const uint8_t* line = raw.data();
    for (int k = 0; k < n_pass; ++k) {
        if (w <= pass[k][0] || h <= pass[k][1]) continue;
        const int pw = (w - pass[k][0] + pass[k][2] - 1) / pass[k][2], ph = (h - pass[k][1] + pass[k][3] - 1) / pass[k][3];
        const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (int y = 0; y < ph; ++y, line += stride + 1) {
            const uint8_t ft = line[0];
            if (ft > 4) return fail(error, "png: bad filter type");
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0, up = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                const int x = line[1 + i];
                cur[i] = (uint8_t)(ft == 0 ? x : ft == 1 ? x + a : ft == 2 ? x + up : ft == 3 ? x + ((a + up) >> 1) : x + paeth(a, up, c));
            }
            uint8_t* row = rgb.data() + (size_t)(pass[k][1] + y * pass[k][3]) * w * 3;
            for (int x = 0; x < pw; ++x) {
                uint8_t* d = row + 3 * (size_t)(pass[k][0] + x * pass[k][2]);
                uint8_t s[4] = {0, 0, 0, 0};
                for (int c = 0; c < channels; ++c) {
                    if (depth == 8) s[c] = cur[(size_t)x * channels + c];
                    else if (depth == 16) s[c] = cur[((size_t)x * channels + c) * 2];                      // high byte
                    else {
                        const size_t bit = (size_t)x * depth;
                        const uint32_t v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                        s[c] = ctype == 3 ? (uint8_t)v : (uint8_t)(v * (depth == 1 ? 255u : depth == 2 ? 85u : 17u));   // grey scaled to 0..255
                    }
                }
                if (ctype == 3) {
                    if ((size_t)s[0] * 3 + 2 >= pal.size()) return fail(error, "png: palette index out of range");
                    d[0] = pal[3 * s[0]]; d[1] = pal[3 * s[0] + 1]; d[2] = pal[3 * s[0] + 2];
                } else if (channels <= 2) d[0] = d[1] = d[2] = s[0];
                else { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
            }
            prev.swap(cur);
        }
    }
    return true;
}

}  // namespace

bool decode_image_rgb8(const uint8_t* b, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error) {
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    w = h = 0;
    rgb.clear();
    if (n >= 8 && std::memcmp(b, png_sig, 8) == 0) return decode_png(b, n, w, h, rgb, error);
    if (n >= 3 && b[0] == 'P' && (b[1] == '5' || b[1] == '6')) return decode_pnm(b, n, w, h, rgb, error);
    if (n >= 26 && b[0] == 'B' && b[1] == 'M') return decode_bmp(b, n, w, h, rgb, error);
    if (n >= 3 && b[0] == 0xff && b[1] == 0xd8) return decode_jpeg_rgb8(b, n, w, h, rgb, error);
    if (n >= 18) return decode_tga(b, n, w, h, rgb, error);             // TGA has no signature: tried last, like stb_image
    return fail(error, "unknown image format");
}

bool decode_image_rgb8(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { error = "texture file not found: " + path; return false; }
    std::vector<uint8_t> bytes;
    uint8_t buf[65536];
    for (size_t k; (k = std::fread(buf, 1, sizeof buf, f)) > 0;) bytes.insert(bytes.end(), buf, buf + k);
    std::fclose(f);
    if (!decode_image_rgb8(bytes.data(), bytes.size(), w, h, rgb, error)) { error = path + ": " + error; return false; }
    return true;
}

void encode_png(const uint8_t* pixels, int width, int height, int channels, bool bottom_up, std::vector<uint8_t>& file) {
    const size_t stride = (size_t)width * channels;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height), cand(stride);
    const std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < height; ++y) {
        const uint8_t* cur = pixels + stride * (size_t)(bottom_up ? height - 1 - y : y);
        const uint8_t* prev = y == 0 ? zero.data() : pixels + stride * (size_t)(bottom_up ? height - y : y - 1);
        uint8_t* dst = raw.data() + (stride + 1) * (size_t)y;
        // the filter whose output has the smallest sum of absolute values (the heuristic of the PNG specification, 12.8)
        uint64_t best = ~0ull;
        for (int ft = 0; ft < 5; ++ft) {
            uint64_t sum = 0;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)channels ? cur[i - channels] : 0, up = prev[i], c = i >= (size_t)channels ? prev[i - channels] : 0;
                const int pred = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? up : ft == 3 ? (a + up) >> 1 : paeth(a, up, c);
                cand[i] = (uint8_t)(cur[i] - pred);
                sum += cand[i] < 128 ? cand[i] : 256 - cand[i];
            }
            if (sum < best) { best = sum; dst[0] = (uint8_t)ft; std::memcpy(dst + 1, cand.data(), stride); }
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("png: deflate failed");
    auto be32 = [&](std::vector<uint8_t>& v, uint32_t x) { for (int k = 3; k >= 0; --k) v.push_back((uint8_t)(x >> (8 * k))); };
    auto chunk = [&](const char* tag, const uint8_t* d, size_t n) {
        be32(file, (uint32_t)n);
        const size_t at = file.size();
        file.insert(file.end(), tag, tag + 4);
        file.insert(file.end(), d, d + n);
        be32(file, (uint32_t)crc32(0, file.data() + at, (uInt)(n + 4)));
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    file.assign(sig, sig + 8);
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    const uint8_t tail[5] = {8, (uint8_t)(channels == 4 ? 6 : 2), 0, 0, 0};
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk("IHDR", ihdr.data(), ihdr.size());
    chunk("IDAT", z.data(), zlen);
    chunk("IEND", nullptr, 0);
}

void texture_to_array_bytes(const uint8_t* rgb, int img_w, int img_h, int width, int height, uint8_t* out) {
    const float inv_255 = 1.0f / 255.0f;
    const size_t n_src = (size_t)img_w * img_h;
    std::vector<float> image(3 * n_src);                                  // Scene.h:326-333 (also :653-661)
    for (size_t i = 0; i < 3 * n_src; ++i) image[i] = 255 * (rgb[i] * inv_255);
    if (img_w == width && img_h == height) {                              // Scene.h:648-662: no resize
        for (size_t i = 0; i < 3 * n_src; ++i) out[i] = (uint8_t)image[i];
        return;
    }
    const float x_ratio = width > 1 ? float(img_w) / (width) : 1;         // Scene.h:337-338
    const float y_ratio = height > 1 ? float(img_h) / (height) : 1;
    // flat index as the reference forms it (a column one past the row end reads the next row's first pixel);
    // past the end of the whole image — undefined behaviour there — the last pixel is used
    auto at = [&](int yy, int xx, int c) { size_t k = (size_t)yy * img_w + xx; if (k >= n_src) k = n_src - 1; return image[3 * k + c]; };
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) {
            const float fx = x_ratio * j, fy = y_ratio * i;
            const int xl = (int)std::floor(fx), yl = (int)std::floor(fy), xh = (int)std::ceil(fx), yh = (int)std::ceil(fy);
            const float x_weight = fx - xl, y_weight = fy - yl;
            for (int c = 0; c < 3; ++c) {
                const float a = at(yl, xl, c), b = at(yl, xh, c), cc = at(yh, xl, c), d = at(yh, xh, c);
                const float pixel = a * (1 - x_weight) * (1 - y_weight) + b * x_weight * (1 - y_weight) +
                                    cc * y_weight * (1 - x_weight) + d * x_weight * y_weight;     // Scene.h:359-362
                out[3 * ((size_t)i * width + j) + c] = (uint8_t)pixel;    // Scene.h:701-703: float -> unsigned char
            }
        }
}

}  // namespace crt
