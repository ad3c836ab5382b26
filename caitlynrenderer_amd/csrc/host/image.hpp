// Texture files for the loader's map_Kd path (Caitlyn/Scene.h:597-710): decode to 8-bit RGB, top row first —
// what `stbi_load(name, &w, &h, 0, 3)` hands the reference (Scene.h:619) — and the reference's own bilinear resize
// to the common texture-array size (Scene.h:321-371).
//
// stb_image is a third-party header the reference vendors; it is not copied here.  image.cpp decodes the lossless
// formats: PNM (binary P5/P6, 8-bit), BMP (uncompressed 8/24/32 bpp), TGA (types 1/2/3 and their RLE forms 9/10/11)
// and PNG (all colour types and depths, Adam7 included, via zlib's inflate).  jpeg.cpp decodes JPEG (baseline and
// progressive) with the arithmetic of the reference's decoder, whose output defines the bytes for a lossy format.
// Every decoder is held to the reference's own stb_image byte for byte (tests/golden/stb_decodes.npz, made by
// tests/golden/make_stb_fixtures.py with oracle/_ref/libstbref.so).  GIF/PSD/PIC/HDR are refused.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace crt {

// rgb: h rows of w pixels, 3 bytes each, top row first.  Returns false and sets `error` on failure.
bool decode_image_rgb8(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error);
bool decode_image_rgb8(const uint8_t* bytes, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error);

// Scene.h:321-371 + :688-710: bilinear resize in fp32 on the 0..255 scale, result truncated to bytes the way
// `vector<unsigned char>::emplace_back(float)` does.  When the source already has the requested size the reference
// skips the resize and passes every byte through `255 * (b * (1/255.f))` (Scene.h:650-662) — the identity for all
// 256 values in fp32, as the tests check; `texture_to_array_bytes` follows both paths.
bool decode_jpeg_rgb8(const uint8_t* bytes, size_t n, int& w, int& h, std::vector<uint8_t>& rgb, std::string& error);   // jpeg.cpp

// PNG file (8-bit RGB or RGBA, rows filtered adaptively, zlib deflate) of `height` rows of `width` pixels with `channels`
// (3 or 4) bytes each; `bottom_up`: the first row in memory is the bottom row of the picture (the GL orientation of
// crt_resolve's output) and the file is written top row first.
void encode_png(const uint8_t* pixels, int width, int height, int channels, bool bottom_up, std::vector<uint8_t>& file);

void texture_to_array_bytes(const uint8_t* rgb, int w, int h, int out_w, int out_h, uint8_t* out);

}  // namespace crt
