// image_io.h — image file decoders of the loader (row f2).  The reference goes through the vendored stb_image (`stbi_load(path, &w, &h, &comp, 4)`,
// `stbi_load_16`, src/nerf_loader.cu:581, 589, 606, 636) and tinyexr (src/tinyexr_wrapper.cu); this build owns its decoders: PNG (png_reader.*, zlib),
// JPEG baseline + progressive (jpeg_reader.cpp), Radiance RGBE as stb_image's 8-bit view of it (hdr_reader.cpp), OpenEXR scanline + tiled NONE / RLE / ZIPS / ZIP / PIZ (exr_reader.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ngp {
// OpenEXR -> interleaved RGBA float32, top row first (missing alpha = 1).  Throws std::runtime_error.
void read_exr_rgba_f32(const std::string& path, int& w, int& h, std::vector<float>& rgba);
void decode_exr_rgba_f32(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<float>& rgba);
}

namespace ngp {
// JPEG (baseline / progressive Huffman, 8 bit, grey or YCbCr) -> RGBA8, alpha 255.  Throws std::runtime_error.
void decode_jpeg_rgba8(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<uint8_t>& pixels);
// Radiance .hdr -> RGBA8 the way stbi_load(.., 4) delivers it (float -> pow(v, 1 / 2.2) * 255 + 0.5, alpha 255).  Throws std::runtime_error.
void decode_hdr_rgba8(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<uint8_t>& pixels);
bool is_hdr_signature(const uint8_t* data, size_t n_bytes);
// `stbi_load(path, &w, &h, &comp, 4)` for the formats this build decodes: dispatch on the file signature (PNG, JPEG, HDR), not on the extension
void read_image_rgba8(const std::string& path, int& w, int& h, std::vector<uint8_t>& pixels);
}
