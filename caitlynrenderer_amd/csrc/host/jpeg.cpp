// JPEG (ITU-T T.81) decoder for the loader's texture path: baseline and progressive Huffman, 8-bit, 1 / 3 / 4 components.
//
// A JPEG's decoded BYTES are not fixed by the standard — they depend on the decoder's inverse DCT, chroma upsampling and
// colour conversion — so "what the reference sees" is what ITS decoder produces: stb_image v2.23, vendored by the reference
// and called as stbi_load(name, &w, &h, 0, 3) (Caitlyn/Scene.h:619).  This file is written from T.81 and implements the
// arithmetic choices that decoder makes, so that the bytes agree exactly (tests/golden/stb_decodes.npz holds the
// reference decoder's output for a set of files; tests/test_textures.py compares):
//   * inverse DCT: the 12-bit fixed-point "slow integer" factorisation (Loeffler-Ligtenberg-Moschytz, as in the IJG
//     library's jidctint), two extra bits kept between the column and the row pass, +128 level shift folded into the
//     final rounding, results clamped to 0..255; coefficients live in 16-bit words (products wrap to 16 bits);
//   * chroma upsampling by 2 (horizontally, vertically or both): the 3:1 "triangle" filter centred the JFIF way, edges
//     replicated over the component's real rows / ceil(width / factor) columns; any other factor: nearest sample;
//   * YCbCr -> RGB in 20-bit fixed point with the constants 1.402, 0.71414, 0.34414, 1.772 rounded to 12 bits, the
//     Cb contribution to green truncated to its upper 16 bits;
//   * colour model: three components are RGB when their ids are 'R','G','B' or an Adobe marker says transform 0 and there
//     is no JFIF marker, otherwise YCbCr; four components are CMYK (Adobe transform 0), YCCK (2) or YCbCr + ignored.
// Malformed files are refused (the reference's decoder also refuses most; where it would instead return partly decoded
// garbage this one refuses too: a refused texture is an error the caller sees).
#include "image.hpp"

#include <cstring>

namespace crt {
namespace {

// position in the 8x8 block (row-major) of the k-th coefficient of the zigzag sequence (T.81 figure A.6)
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Fail { const char* what; };

// Huffman table in the canonical form of T.81 annex C / F.2.2.3, with a 9-bit first-level lookup
struct Huffman {
    bool defined = false;
    uint8_t symbol[256];
    int32_t first_code[18], first_index[18], count[18];
    uint16_t look[512];                                   // (length << 8) | symbol for codes of up to 9 bits, 0 otherwise
    void build(const int counts[16], const uint8_t* symbols, int total) {
        std::memcpy(symbol, symbols, (size_t)total);
        std::memset(look, 0, sizeof look);
        int code = 0, index = 0;
        for (int len = 1; len <= 16; ++len) {
            first_code[len] = code; first_index[len] = index; count[len] = counts[len - 1];
            if (counts[len - 1] && code + counts[len - 1] - 1 >= (1 << len)) throw Fail{"jpeg: bad Huffman code lengths"};
            if (len <= 9)
                for (int k = 0; k < counts[len - 1]; ++k)
                    for (int fill = 0; fill < (1 << (9 - len)); ++fill)
                        look[((code + k) << (9 - len)) + fill] = (uint16_t)((len << 8) | symbols[index + k]);
            code = (code + counts[len - 1]) << 1;
            index += counts[len - 1];
        }
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int dc_pred = 0;
    int x = 0, y = 0;                                     // samples this component really has
    int w2 = 0, h2 = 0;                                   // plane size: whole MCUs
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;                           // progressive: all coefficients, block by block
    bool whole_mcus = false;                              // samples exist for the padded plane, not only ceil(x/8) blocks a row
};

struct Decoder {
    Decoder(const uint8_t* bytes, size_t size) : p(bytes), n(size) {}
    const uint8_t* p; size_t n, at = 0;
    // entropy-coded segment reader: `bits` valid bits at the top of `buf`; zero bits once a marker (or the end) is met
    uint32_t buf = 0; int bits = 0; int marker = -1; bool no_more = false;
    Huffman dc[4], ac[4];
    uint16_t quant[4][64]; bool quant_defined[4] = {false, false, false, false};
    Component comp[4];
    int n_comp = 0, width = 0, height = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    bool progressive = false, jfif = false; int adobe_transform = -1, rgb_ids = 0;
    int restart_interval = 0, todo = 0, eob_run = 0;
    int scan_n = 0, order[4], spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0;

    int get8() { return at < n ? p[at++] : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }
    bool at_end() const { return at >= n; }
    void skip(int k) { if (k < 0 || at + (size_t)k > n) at = n; else at += (size_t)k; }

    void fill() {
        do {
            const unsigned b = no_more ? 0u : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();             // fill bytes
                if (c != 0) { marker = c; no_more = true; return; }
            }
            buf |= b << (24 - bits);
            bits += 8;
        } while (bits <= 24);
    }
    int decode(const Huffman& h) {
        if (bits < 16) fill();
        const uint16_t e = h.look[buf >> 23];
        if (e) {
            const int len = e >> 8;
            if (len > bits) throw Fail{"jpeg: bad Huffman code"};
            buf <<= len; bits -= len;
            return e & 255;
        }
        for (int len = 10; len <= 16; ++len) {
            const int code = (int)(buf >> (32 - len)) - h.first_code[len];
            if (code >= 0 && code < h.count[len]) {
                if (len > bits) throw Fail{"jpeg: bad Huffman code"};
                buf <<= len; bits -= len;
                return h.symbol[h.first_index[len] + code];
            }
        }
        throw Fail{"jpeg: bad Huffman code"};
    }
    int get_bits(int k) {                                 // k in 1..16
        if (bits < k) fill();
        // fill() adds nothing once it has met a marker: entropy data that ends in mid-coefficient is refused here (stb_image
        // goes on with a negative bit count and returns partly decoded garbage); `bits` therefore never drops below zero
        if (bits < k) throw Fail{"jpeg: entropy-coded data ends inside a coefficient"};
        const int v = (int)(buf >> (32 - k));
        buf <<= k; bits -= k;
        return v;
    }
    int get_bit() {
        if (bits < 1) fill();
        if (bits < 1) throw Fail{"jpeg: entropy-coded data ends inside a coefficient"};
        const int v = (int)(buf >> 31);
        buf <<= 1; --bits;
        return v;
    }
    int receive_extend(int k) {                           // T.81 F.2.2.1: k-bit magnitude, sign from its top bit
        if (k > 15) throw Fail{"jpeg: bad coefficient size"};
        const int v = get_bits(k);
        return v < (1 << (k - 1)) ? v - (1 << k) + 1 : v;
    }
    void reset_entropy() {
        bits = 0; buf = 0; no_more = false; marker = -1;
        for (Component& c : comp) c.dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }
    // after a restart interval: true when decoding goes on (an RSTn marker is there), false when the scan ends here
    bool interval_done() {
        if (--todo > 0) return true;
        if (bits < 24) fill();
        if (marker < 0xd0 || marker > 0xd7) return false;
        reset_entropy();
        return true;
    }

    static int16_t wrap16(int64_t v) { return (int16_t)(uint16_t)(uint64_t)v; }

    void decode_block_baseline(int16_t* data, Component& c) {
        const Huffman &hd = dc[c.hd], &ha = ac[c.ha];
        const uint16_t* q = quant[c.tq];
        const int t = decode(hd);
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? receive_extend(t) : 0;
        c.dc_pred += diff;
        data[0] = wrap16((int64_t)c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = decode(ha), s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;                    // end of block
                k += 16;
            } else {
                k += r;
                if (k > 63) throw Fail{"jpeg: coefficient index out of range"};
                const int z = kZigzag[k++];
                data[z] = wrap16((int64_t)receive_extend(s) * q[z]);
            }
        }
    }
    void decode_block_prog_dc(int16_t* data, Component& c) {
        if (spec_end != 0) throw Fail{"jpeg: a scan cannot mix DC and AC coefficients"};
        if (bits < 16) fill();
        if (succ_high == 0) {
            std::memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode(dc[c.hd]);
            const int diff = t ? receive_extend(t) : 0;
            c.dc_pred += diff;
            data[0] = wrap16((int64_t)c.dc_pred * (1 << succ_low));
        } else if (get_bit()) {
            data[0] = wrap16((int64_t)data[0] + (int16_t)(1 << succ_low));
        }
    }
    void refine(int16_t* pc, int bit) {                   // one correction bit for an already non-zero coefficient
        if (get_bit() && (*pc & bit) == 0) *pc = wrap16(*pc > 0 ? (int64_t)*pc + bit : (int64_t)*pc - bit);
    }
    void decode_block_prog_ac(int16_t* data, Component& c) {
        if (spec_start == 0) throw Fail{"jpeg: a scan cannot mix DC and AC coefficients"};
        const Huffman& ha = ac[c.ha];
        if (succ_high == 0) {                             // first pass over this band (T.81 G.1.2.2)
            if (eob_run) { --eob_run; return; }
            int k = spec_start;
            do {
                const int rs = decode(ha), s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) throw Fail{"jpeg: coefficient index out of range"};
                    data[kZigzag[k++]] = wrap16((int64_t)receive_extend(s) * (1 << succ_low));
                }
            } while (k <= spec_end);
            return;
        }
        const int bit = 1 << succ_low;                    // refinement pass (T.81 G.1.2.3)
        if (eob_run) {
            --eob_run;
            for (int k = spec_start; k <= spec_end; ++k) {
                int16_t* pc = &data[kZigzag[k]];
                if (*pc != 0) refine(pc, bit);
            }
            return;
        }
        int k = spec_start;
        do {
            const int rs = decode(ha);
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += get_bits(r);
                    r = 64;                               // to the end of the band: only corrections remain
                }
            } else {
                if (s != 1) throw Fail{"jpeg: bad Huffman code"};
                s = get_bit() ? bit : -bit;
            }
            while (k <= spec_end) {                       // skip r zero coefficients, correcting the non-zero ones passed
                int16_t* pc = &data[kZigzag[k++]];
                if (*pc != 0) refine(pc, bit);
                else {
                    if (r == 0) { *pc = (int16_t)s; break; }
                    --r;
                }
            }
        } while (k <= spec_end);
    }

    // ---- inverse DCT: one 8-point pass of the LLM factorisation, constants scaled by 2^12.  Inputs s0..s7, results as
    // the even part e[0..3] (+ rounding bias added by the caller) and the odd part o[0..3]: out[i] = e[i] + o[3-i] ...
    struct Pass { int32_t e0, e1, e2, e3, o0, o1, o2, o3; };
    static Pass idct8(int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4, int64_t s5, int64_t s6, int64_t s7) {
        // even part: rotation of (s2, s6) by 6pi/16, butterfly with (s0 +- s4)
        const int64_t z = (s2 + s6) * 2217;               // 0.5411961 * 4096
        const int64_t c2 = z + s6 * -7567;                // -1.847759065
        const int64_t c3 = z + s2 * 3135;                 //  0.765366865
        const int64_t a0 = (s0 + s4) * 4096, a1 = (s0 - s4) * 4096;
        // odd part
        const int64_t z5 = (s7 + s3 + s5 + s1) * 4816;    //  1.175875602
        const int64_t z1 = z5 + (s7 + s1) * -3685;        // -0.899976223
        const int64_t z2 = z5 + (s5 + s3) * -10497;       // -2.562915447
        const int64_t z3 = (s7 + s3) * -8034;             // -1.961570560
        const int64_t z4 = (s5 + s1) * -1597;             // -0.390180644
        Pass r;
        r.e0 = (int32_t)(uint32_t)(uint64_t)(a0 + c3); r.e3 = (int32_t)(uint32_t)(uint64_t)(a0 - c3);
        r.e1 = (int32_t)(uint32_t)(uint64_t)(a1 + c2); r.e2 = (int32_t)(uint32_t)(uint64_t)(a1 - c2);
        r.o3 = (int32_t)(uint32_t)(uint64_t)(s1 * 6149 + z1 + z4);      // 1.501321110
        r.o2 = (int32_t)(uint32_t)(uint64_t)(s3 * 12586 + z2 + z3);     // 3.072711026
        r.o1 = (int32_t)(uint32_t)(uint64_t)(s5 * 8410 + z2 + z4);      // 2.053119869
        r.o0 = (int32_t)(uint32_t)(uint64_t)(s7 * 1223 + z1 + z3);      // 0.298631336
        return r;
    }
    static int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
    static int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
    static uint8_t clamp8(int32_t v) { return v < 0 ? 0 : v > 255 ? 255 : (uint8_t)v; }
    static void idct_block(uint8_t* out, int stride, const int16_t* d) {
        int32_t mid[64];
        for (int x = 0; x < 8; ++x) {                     // columns; 2^12 scale brought down to 2^2
            const Pass t = idct8(d[x], d[8 + x], d[16 + x], d[24 + x], d[32 + x], d[40 + x], d[48 + x], d[56 + x]);
            const int32_t e[4] = {add32(t.e0, 512), add32(t.e1, 512), add32(t.e2, 512), add32(t.e3, 512)};
            const int32_t o[4] = {t.o3, t.o2, t.o1, t.o0};
            for (int i = 0; i < 4; ++i) {
                mid[8 * i + x] = add32(e[i], o[i]) >> 10;
                mid[8 * (7 - i) + x] = sub32(e[i], o[i]) >> 10;
            }
        }
        const int32_t bias = 65536 + (128 << 17);         // rounding for the 17-bit shift + the level shift of 128
        for (int y = 0; y < 8; ++y) {
            const int32_t* m = mid + 8 * y;
            const Pass t = idct8(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
            const int32_t e[4] = {add32(t.e0, bias), add32(t.e1, bias), add32(t.e2, bias), add32(t.e3, bias)};
            const int32_t o[4] = {t.o3, t.o2, t.o1, t.o0};
            uint8_t* row = out + (size_t)stride * y;
            for (int i = 0; i < 4; ++i) {
                row[i] = clamp8(add32(e[i], o[i]) >> 17);
                row[7 - i] = clamp8(sub32(e[i], o[i]) >> 17);
            }
        }
    }

    // ---- segments
    void read_tables_or_skip(int m) {
        if (m == 0xdd) {                                  // DRI
            if (get16() != 4) throw Fail{"jpeg: bad DRI length"};
            restart_interval = get16();
        } else if (m == 0xdb) {                           // DQT
            int len = get16() - 2;
            while (len > 0) {
                const int q = get8(), wide = q >> 4, t = q & 15;
                if (wide > 1) throw Fail{"jpeg: bad DQT precision"};
                if (t > 3) throw Fail{"jpeg: bad DQT table"};
                for (int i = 0; i < 64; ++i) quant[t][kZigzag[i]] = (uint16_t)(wide ? get16() : get8());
                quant_defined[t] = true;
                len -= wide ? 129 : 65;
            }
            if (len != 0) throw Fail{"jpeg: bad DQT length"};
        } else if (m == 0xc4) {                           // DHT
            int len = get16() - 2;
            while (len > 0) {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) throw Fail{"jpeg: bad DHT header"};
                int counts[16], total = 0;
                for (int i = 0; i < 16; ++i) { counts[i] = get8(); total += counts[i]; }
                if (total > 256) throw Fail{"jpeg: bad DHT symbol count"};
                uint8_t symbols[256];
                for (int i = 0; i < total; ++i) symbols[i] = (uint8_t)get8();
                (tc ? ac[th] : dc[th]).build(counts, symbols, total);
                len -= 17 + total;
            }
            if (len != 0) throw Fail{"jpeg: bad DHT length"};
        } else if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {   // APPn / COM
            int len = get16();
            if (len < 2) throw Fail{"jpeg: bad segment length"};
            len -= 2;
            if (m == 0xe0 && len >= 5) {
                static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; ++i) ok &= get8() == tag[i];
                len -= 5;
                if (ok) jfif = true;
            } else if (m == 0xee && len >= 12) {
                static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; ++i) ok &= get8() == tag[i];
                len -= 6;
                if (ok) { get8(); get16(); get16(); adobe_transform = get8(); len -= 6; }
            }
            skip(len);
        } else {
            throw Fail{m < 0 ? "jpeg: marker expected" : "jpeg: unsupported marker"};
        }
    }
    int next_marker() {                                   // -1: the next byte is not a marker
        if (marker >= 0) { const int m = marker; marker = -1; return m; }
        int x = get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = get8();
        return x;
    }
    void read_frame_header() {
        const int len = get16();
        if (len < 11) throw Fail{"jpeg: bad SOF length"};
        if (get8() != 8) throw Fail{"jpeg: only 8-bit samples"};
        height = get16(); width = get16();
        if (height == 0 || width == 0) throw Fail{"jpeg: bad size"};
        n_comp = get8();
        if (n_comp != 1 && n_comp != 3 && n_comp != 4) throw Fail{"jpeg: bad component count"};
        if (len != 8 + 3 * n_comp) throw Fail{"jpeg: bad SOF length"};
        for (int i = 0; i < n_comp; ++i) {
            Component& c = comp[i];
            c.id = get8();
            if (n_comp == 3 && c.id == "RGB"[i]) ++rgb_ids;
            const int q = get8();
            c.h = q >> 4; c.v = q & 15; c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) throw Fail{"jpeg: bad sampling factor"};
            if (c.tq > 3) throw Fail{"jpeg: bad quantisation table index"};
            if (c.h > h_max) h_max = c.h;
            if (c.v > v_max) v_max = c.v;
        }
        if (width > (1 << 15) || height > (1 << 15)) throw Fail{"jpeg: image too large"};
        // a block costs at least a bit of entropy-coded data: a header of a few bytes cannot ask for gigabytes
        if ((size_t)width * height / 1024 > n) throw Fail{"jpeg: truncated"};
        mcu_x = (width + 8 * h_max - 1) / (8 * h_max);
        mcu_y = (height + 8 * v_max - 1) / (8 * v_max);
        for (int i = 0; i < n_comp; ++i) {
            Component& c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
        }
    }
    void read_scan_header() {
        const int len = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > 4 || scan_n > n_comp) throw Fail{"jpeg: bad SOS component count"};
        if (len != 6 + 2 * scan_n) throw Fail{"jpeg: bad SOS length"};
        for (int i = 0; i < scan_n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < n_comp && comp[which].id != id) ++which;
            if (which == n_comp) throw Fail{"jpeg: SOS names an unknown component"};
            comp[which].hd = q >> 4; comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3) throw Fail{"jpeg: bad Huffman table index"};
            order[i] = which;
        }
        spec_start = get8(); spec_end = get8();
        const int a = get8();
        succ_high = a >> 4; succ_low = a & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) throw Fail{"jpeg: bad SOS"};
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) throw Fail{"jpeg: bad SOS"};
            spec_end = 63;
        }
        for (int i = 0; i < scan_n; ++i) {                // tables the scan will use must exist
            const Component& c = comp[order[i]];
            const bool need_dc = !progressive || spec_start == 0, need_ac = !progressive || spec_start != 0;
            if (need_dc && !(progressive && succ_high != 0) && !dc[c.hd].defined) throw Fail{"jpeg: scan uses an undefined Huffman table"};
            if (need_ac && !ac[c.ha].defined) throw Fail{"jpeg: scan uses an undefined Huffman table"};
            if (!progressive && !quant_defined[c.tq]) throw Fail{"jpeg: scan uses an undefined quantisation table"};
        }
    }
    void read_scan_data() {
        reset_entropy();
        int16_t block[64];
        if (scan_n == 1) {                                // one component: its blocks in raster order, real size only
            Component& c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    if (!progressive) {
                        decode_block_baseline(block, c);
                        idct_block(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, block);
                    } else {
                        int16_t* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * (c.w2 / 8));
                        if (spec_start == 0) decode_block_prog_dc(data, c);
                        else decode_block_prog_ac(data, c);
                    }
                    if (!interval_done()) return;
                }
            return;
        }
        for (int j = 0; j < mcu_y; ++j)                   // interleaved: MCU by MCU
            for (int i = 0; i < mcu_x; ++i) {
                for (int k = 0; k < scan_n; ++k) {
                    Component& c = comp[order[k]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x) {
                            const int bx = i * c.h + x, by = j * c.v + y;
                            if (!progressive) {
                                decode_block_baseline(block, c);
                                idct_block(c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, block);
                                c.whole_mcus = true;
                            } else {
                                decode_block_prog_dc(c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8)), c);
                            }
                        }
                }
                if (!interval_done()) return;
            }
    }
    void finish_progressive() {
        for (int k = 0; k < n_comp; ++k) {
            Component& c = comp[k];
            if (!quant_defined[c.tq]) throw Fail{"jpeg: undefined quantisation table"};
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    int16_t* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * (c.w2 / 8));
                    for (int z = 0; z < 64; ++z) data[z] = wrap16((int64_t)data[z] * quant[c.tq][z]);
                    idct_block(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
        }
    }
    void decode_planes() {
        if (next_marker() != 0xd8) throw Fail{"jpeg: no SOI"};
        int m = next_marker();
        while (m != 0xc0 && m != 0xc1 && m != 0xc2) {
            read_tables_or_skip(m);
            m = next_marker();
            while (m < 0) {                               // padding after a segment
                if (at_end()) throw Fail{"jpeg: no SOF"};
                m = next_marker();
            }
        }
        progressive = m == 0xc2;
        read_frame_header();
        bool scanned = false;
        m = next_marker();
        while (m != 0xd9) {
            if (m == 0xda) {
                read_scan_header();
                read_scan_data();
                scanned = true;
                if (marker < 0)                           // stray bytes after the entropy-coded data: look for the next marker
                    while (!at_end())
                        if (get8() == 0xff) { marker = get8(); break; }
            } else if (m == 0xdc) {                       // DNL
                const int len = get16(), lines = get16();
                if (len != 4 || lines != height) throw Fail{"jpeg: bad DNL"};
            } else {
                read_tables_or_skip(m);
            }
            m = next_marker();
        }
        if (!scanned) throw Fail{"jpeg: no scan"};
        if (progressive) finish_progressive();
    }
};

// ---- chroma upsampling: one output row from the nearer and the farther source row
void upsample_row(uint8_t* out, const uint8_t* near_row, const uint8_t* far_row, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) { std::memcpy(out, near_row, (size_t)w); return; }
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near_row[i] + far_row[i] + 2) >> 2);
        return;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near_row[0]; return; }
        out[0] = near_row[0];
        for (int i = 0; i + 1 < w; ++i) {                 // the two samples between source samples i and i + 1
            out[2 * i + 1] = (uint8_t)((3 * near_row[i] + near_row[i + 1] + 2) >> 2);
            out[2 * i + 2] = (uint8_t)((3 * near_row[i + 1] + near_row[i] + 2) >> 2);
        }
        // the reference's decoder weights the last interpolated sample towards w - 2, not w - 1 (unlike its left edge)
        out[2 * w - 2] = (uint8_t)((3 * near_row[w - 2] + near_row[w - 1] + 2) >> 2);
        out[2 * w - 1] = near_row[w - 1];
        return;
    }
    if (hs == 2 && vs == 2) {
        int prev = 3 * near_row[0] + far_row[0];          // vertical blend, scaled by 4
        out[0] = (uint8_t)((prev + 2) >> 2);
        if (w == 1) { out[1] = out[0]; return; }
        for (int i = 1; i < w; ++i) {
            const int cur = 3 * near_row[i] + far_row[i];
            out[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
            prev = cur;
        }
        out[2 * w - 1] = (uint8_t)((prev + 2) >> 2);
        return;
    }
    for (int i = 0; i < w; ++i)                           // any other factor: nearest sample
        for (int k = 0; k < hs; ++k) out[i * hs + k] = near_row[i];
}

uint8_t mul255(unsigned a, unsigned b) {                  // a * b / 255, rounded
    const unsigned t = a * b + 128;
    return (uint8_t)((t + (t >> 8)) >> 8);
}

void ycc_to_rgb(uint8_t* out, const uint8_t* y, const uint8_t* cb_row, const uint8_t* cr_row, int count) {
    for (int i = 0; i < count; ++i, out += 3) {
        const int32_t base = ((int32_t)y[i] << 20) + (1 << 19);
        const int32_t cr = cr_row[i] - 128, cb = cb_row[i] - 128;
        int32_t r = base + cr * (5743 << 8);                                                   // 1.402
        int32_t g = base + cr * -(2925 << 8) + (int32_t)((uint32_t)(cb * -(1410 << 8)) & 0xffff0000u);   // 0.71414, 0.34414
        int32_t b = base + cb * (7258 << 8);                                                   // 1.772
        r >>= 20; g >>= 20; b >>= 20;
        out[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        out[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
        out[2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
    }
}

}  // namespace

bool decode_jpeg_rgb8(const uint8_t* bytes, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error) {
    Decoder d(bytes, n);
    try {
        d.decode_planes();
    } catch (const Fail& f) {
        error = f.what;
        return false;
    } catch (const std::bad_alloc&) {
        error = "jpeg: out of memory";
        return false;
    }
    w = d.width; h = d.height;
    rgb.resize((size_t)w * h * 3);
    const bool is_rgb = d.n_comp == 3 && (d.rgb_ids == 3 || (d.adobe_transform == 0 && !d.jfif));
    // per component: the two source rows the current output row lies between, stepped as the rows go by
    struct Rows { int hs, vs, w_lores, ystep, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
    Rows rows[4];
    for (int k = 0; k < d.n_comp; ++k) {
        Rows& r = rows[k];
        r.hs = d.h_max / d.comp[k].h; r.vs = d.v_max / d.comp[k].v;
        r.ystep = r.vs >> 1; r.ypos = 0;
        r.w_lores = (w + r.hs - 1) / r.hs;
        r.line0 = r.line1 = d.comp[k].plane.data();
        r.buf.assign((size_t)r.w_lores * r.hs + 8, 0);
        // sampling factors that do not divide the largest one make the upsampler read columns no block was decoded for
        // (the reference's decoder returns uninitialised memory there): refused
        const int have = d.comp[k].whole_mcus ? d.comp[k].w2 : 8 * ((d.comp[k].x + 7) >> 3);
        if (r.w_lores > have) { error = "jpeg: unsupported sampling factors"; return false; }
    }
    for (int j = 0; j < h; ++j) {
        uint8_t* out = rgb.data() + (size_t)j * w * 3;
        const uint8_t* c[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < d.n_comp; ++k) {
            Rows& r = rows[k];
            const bool lower_half = r.ystep >= (r.vs >> 1);      // this output row is nearer to line1 than to line0
            upsample_row(r.buf.data(), lower_half ? r.line1 : r.line0, lower_half ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            c[k] = r.buf.data();
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y) r.line1 += d.comp[k].w2;
            }
        }
        if (d.n_comp == 1) {
            for (int i = 0; i < w; ++i) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = c[0][i];
        } else if (d.n_comp == 3) {
            if (is_rgb) for (int i = 0; i < w; ++i) { out[3 * i] = c[0][i]; out[3 * i + 1] = c[1][i]; out[3 * i + 2] = c[2][i]; }
            else ycc_to_rgb(out, c[0], c[1], c[2], w);
        } else if (d.adobe_transform == 0) {              // CMYK
            for (int i = 0; i < w; ++i) for (int k = 0; k < 3; ++k) out[3 * i + k] = mul255(c[k][i], c[3][i]);
        } else {
            ycc_to_rgb(out, c[0], c[1], c[2], w);
            if (d.adobe_transform == 2)                   // YCCK
                for (int i = 0; i < w; ++i) for (int k = 0; k < 3; ++k) out[3 * i + k] = mul255(255u - out[3 * i + k], c[3][i]);
        }
    }
    return true;
}

}  // namespace crt
