// png_reader.h — PNG -> RGBA8 for the transforms.json loader.  The reference decodes images with the vendored stb_image
// (`stbi_load(path, &w, &h, &comp, 4)`, src/nerf_loader.cu:581); this build links zlib and implements the PNG container, the five
// scanline filters and the conversion to 4 x 8 bit the same way stb_image's 4-channel request does (grey -> rgb replicate, opaque alpha,
// 16 bit -> high byte, palette + tRNS).  Adam7-interlaced files are de-interlaced (the seven passes are unfiltered one by one).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ngp {
// throws std::runtime_error; pixels = h rows of w RGBA8 texels, top row first
void read_png_rgba8(const std::string& path, int& w, int& h, std::vector<uint8_t>& pixels);
void decode_png_rgba8(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<uint8_t>& pixels);
// one 16-bit channel, what `stbi_load_16(path, &w, &h, &comp, 1)` returns for a PNG (depth images of the loader)
void read_png_gray16(const std::string& path, int& w, int& h, std::vector<uint16_t>& pixels);
void decode_png_gray16(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<uint16_t>& pixels);
}
