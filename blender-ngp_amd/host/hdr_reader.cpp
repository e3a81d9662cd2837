// hdr_reader.cpp — Radiance .hdr / .pic (RGBE) -> RGBA8, the way the reference's loader sees such a file: it hands every non-EXR path to
// `stbi_load(path, &w, &h, &comp, 4)` (src/nerf_loader.cu:581), and stb_image turns an HDR file into 8-bit there — RGBE -> float (mantissa * 2^(e - 136)),
// then v -> (float)pow(v, 1 / 2.2) * 255 + 0.5 clamped to [0, 255] and truncated, alpha 255.  This is a build-owned reader of the published format
// (Ward, "Real Pixels", Graphics Gems II): text header ("#?RADIANCE" or "#?RGBE", lines up to an empty one, FORMAT=32-bit_rle_rgbe required), the resolution
// line "-Y h +X w", then per scanline either the new run-length form (2 2 hi lo, then the four byte planes as runs (count > 128: count - 128 copies of the
// next byte) and dumps (count bytes)) or, for widths outside [8, 32767] or a first scanline without the 2 2 marker, flat RGBE quadruples.  Old-style
// repeat-pixel RLE is not read (stb_image does not read it either); other orientations than -Y +X are rejected like there.
#include "image_io.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ngp {

namespace {
struct Reader {
	const uint8_t* p; const uint8_t* end;
	int get() { if (p >= end) throw std::runtime_error{"HDR: truncated file"}; return *p++; }
	std::string line() {   // up to '\n' (not included); the end of the file ends a line too
		std::string s;
		while (p < end && *p != '\n') s.push_back((char)*p++);
		if (p < end) ++p;
		return s;
	}
};

inline uint8_t to_ldr(float v) {   // stb_image's HDR -> LDR rule with its default gamma 2.2, scale 1
	float z = (float)pow((double)v, (double)(1.0f / 2.2f)) * 255 + 0.5f;
	if (z < 0) z = 0;
	if (z > 255) z = 255;
	return (uint8_t)(int)z;
}

inline void put_pixel(uint8_t* out, const uint8_t* rgbe) {
	if (rgbe[3]) {
		const float f = (float)ldexp(1.0f, (int)rgbe[3] - (128 + 8));
		out[0] = to_ldr(rgbe[0] * f); out[1] = to_ldr(rgbe[1] * f); out[2] = to_ldr(rgbe[2] * f);
	} else {
		out[0] = out[1] = out[2] = to_ldr(0.0f);
	}
	out[3] = 255;   // alpha 1.0 -> (int)(255 + 0.5)
}
} // namespace

bool is_hdr_signature(const uint8_t* data, size_t n) {
	return (n >= 11 && !memcmp(data, "#?RADIANCE\n", 11)) || (n >= 7 && !memcmp(data, "#?RGBE\n", 7));
}

void decode_hdr_rgba8(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<uint8_t>& pixels) {
	Reader r{data, data + n_bytes};
	const std::string magic = r.line();
	if (magic != "#?RADIANCE" && magic != "#?RGBE") throw std::runtime_error{"HDR: not a Radiance file"};
	bool rle_rgbe = false;
	for (;;) {
		if (r.p >= r.end) throw std::runtime_error{"HDR: truncated header"};
		const std::string l = r.line();
		if (l.empty()) break;
		if (l == "FORMAT=32-bit_rle_rgbe") rle_rgbe = true;
	}
	if (!rle_rgbe) throw std::runtime_error{"HDR: unsupported format (FORMAT=32-bit_rle_rgbe is read)"};
	const std::string res = r.line();
	if (res.compare(0, 3, "-Y ") != 0) throw std::runtime_error{"HDR: unsupported data layout (-Y h +X w is read)"};
	char* q = nullptr;
	const long hh = strtol(res.c_str() + 3, &q, 10);
	while (*q == ' ') ++q;
	if (strncmp(q, "+X ", 3) != 0) throw std::runtime_error{"HDR: unsupported data layout (-Y h +X w is read)"};
	const long ww = strtol(q + 3, nullptr, 10);
	if (ww <= 0 || hh <= 0 || ww > (1 << 24) || hh > (1 << 24) || (int64_t)ww * hh > ((int64_t)1 << 28)) throw std::runtime_error{"HDR: bad image size"};
	w = (int)ww; h = (int)hh;
	// before the allocation: a run-length scanline costs at least 4 marker bytes + 2 bytes per 127-pixel run and channel, a flat one 4 bytes per pixel
	// (images narrower than 8 or wider than 32767 pixels are always flat — 4 w bytes per scanline, which for w = 1, 2 is LESS than a run-length line's 12)
	{
		const uint64_t flat_line = 4u * (uint64_t)w, rle_line = 4u + 8u * (((uint64_t)w + 126u) / 127u);
		const uint64_t per_line = (w < 8 || w >= 32768) ? flat_line : std::min(flat_line, rle_line);
		if ((uint64_t)h * per_line > (uint64_t)n_bytes + 64u) throw std::runtime_error{"HDR: the header promises more pixels than the file could encode"};
	}
	pixels.assign((size_t)w * h * 4, 0);
	bool flat = w < 8 || w >= 32768;
	std::vector<uint8_t> scan;
	for (int y = 0; y < h && !flat; ++y) {
		const int c1 = r.get(), c2 = r.get(), c3 = r.get();
		if (c1 != 2 || c2 != 2 || (c3 & 0x80)) {
			// no run-length marker: these bytes are a pixel (a valid RGBE pixel has a channel >= 128, so the marker cannot be one).  stb_image then reads the whole
			// image as flat data starting with this pixel; that is a sensible reading only on the first scanline
			if (y != 0) throw std::runtime_error{"HDR: scanline without a run-length marker after run-length scanlines"};
			r.p -= 3;
			flat = true;
			break;
		}
		const int len = (c3 << 8) | r.get();
		if (len != w) throw std::runtime_error{"HDR: invalid decoded scanline length"};
		scan.resize((size_t)w * 4);
		for (int k = 0; k < 4; ++k) {
			int i = 0;
			while (i < w) {
				int count = r.get();
				if (count > 128) {
					count -= 128;
					if (count > w - i) throw std::runtime_error{"HDR: bad RLE data"};
					const uint8_t v = (uint8_t)r.get();
					for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = v;
				} else {
					if (count > w - i) throw std::runtime_error{"HDR: bad RLE data"};
					if (count == 0) throw std::runtime_error{"HDR: bad RLE data (empty dump)"};   // stb_image would loop on it until the file ends
					for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = (uint8_t)r.get();
				}
			}
		}
		uint8_t* row = pixels.data() + (size_t)y * w * 4;
		for (int x = 0; x < w; ++x) put_pixel(row + 4 * x, scan.data() + 4 * (size_t)x);
	}
	if (flat) {
		if ((size_t)(r.end - r.p) < (size_t)w * h * 4) throw std::runtime_error{"HDR: truncated file"};
		for (size_t i = 0; i < (size_t)w * h; ++i) put_pixel(pixels.data() + 4 * i, r.p + 4 * i);
	}
}

} // namespace ngp
