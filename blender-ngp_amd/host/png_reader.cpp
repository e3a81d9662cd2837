#include "png_reader.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ngp {

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static int paeth(int a, int b, int c) {
	const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	if (pa <= pb && pa <= pc) return a;
	if (pb <= pc) return b;
	return c;
}

// container + zlib + the five scanline filters: the unfiltered rows (one filter byte, then row_bytes of samples, per scanline)
struct PngRaw { int w = 0, h = 0, depth = 0, ctype = 0, channels = 0; size_t row_bytes = 0; std::vector<uint8_t> raw, palette, trns; };
static void png_unfilter(const uint8_t* d, size_t n, PngRaw& R) {
	int w = 0, h = 0;
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
	if (n < 8 || memcmp(d, sig, 8) != 0) {
		if (n >= 2 && d[0] == 0xff && d[1] == 0xd8) throw std::runtime_error("JPEG decoding is not part of this build (PNG only)");
		throw std::runtime_error("not a PNG file");
	}
	size_t p = 8;
	int depth = 0, ctype = 0, interlace = 0;
	bool have_ihdr = false;
	std::vector<uint8_t> idat; std::vector<uint8_t>& palette = R.palette; std::vector<uint8_t>& trns = R.trns;
	while (p + 12 <= n) {
		const uint32_t len = be32(d + p);
		const uint8_t* type = d + p + 4;
		const uint8_t* body = d + p + 8;
		if (p + 12 + (size_t)len > n) throw std::runtime_error("PNG: truncated chunk");
		if (!memcmp(type, "IHDR", 4)) {
			if (len < 13) throw std::runtime_error("PNG: bad IHDR");
			w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
			have_ihdr = true;
		} else if (!memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
		else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
		else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
		else if (!memcmp(type, "IEND", 4)) break;
		p += 12 + (size_t)len;
	}
	if (!have_ihdr || w <= 0 || h <= 0) throw std::runtime_error("PNG: missing IHDR");
	if (interlace > 1) throw std::runtime_error("PNG: unknown interlace method");
	int channels;
	switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; default: throw std::runtime_error("PNG: bad colour type"); }
	if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) throw std::runtime_error("PNG: unsupported bit depth");
	if (ctype == 3 && palette.empty()) throw std::runtime_error("PNG: palette image without PLTE");
	if ((uint64_t)w * (uint64_t)h > ((uint64_t)1 << 28)) throw std::runtime_error("PNG: image larger than 2^28 pixels");
	const size_t bits_per_pixel = (size_t)channels * depth;
	const size_t row_bytes = ((size_t)w * bits_per_pixel + 7) / 8;
	// before any allocation sized by IHDR: DEFLATE expands at most 1032 : 1, so a header that promises more filtered bytes than the IDAT chunks could hold is hostile
	// (a 60-byte file must not force a multi-GiB buffer)
	if ((uint64_t)h * (uint64_t)(row_bytes + 1) / 1032u > (uint64_t)idat.size() + 64u) throw std::runtime_error("PNG: IHDR promises more pixels than the IDAT data could encode");
	const size_t bpp = std::max<size_t>(1, bits_per_pixel / 8);
	// unfilter `rows` scanlines of `rb` sample bytes each, in place (PNG spec 9.2); every scanline starts with its filter byte
	auto unfilter = [&](uint8_t* data, int rows, size_t rb) {
		std::vector<uint8_t> prev(rb, 0);
		for (int y = 0; y < rows; ++y) {
			uint8_t* row = data + (size_t)y * (rb + 1);
			const int f = row[0];
			uint8_t* cur = row + 1;
			for (size_t i = 0; i < rb; ++i) {
				const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
				int v = cur[i];
				switch (f) {
					case 0: break;
					case 1: v += a; break;
					case 2: v += b; break;
					case 3: v += (a + b) >> 1; break;
					case 4: v += paeth(a, b, c); break;
					default: throw std::runtime_error("PNG: bad filter type");
				}
				cur[i] = (uint8_t)v;
			}
			memcpy(prev.data(), cur, rb);
		}
	};
	std::vector<uint8_t>& raw = R.raw; raw.assign((row_bytes + 1) * (size_t)h, 0);
	if (!interlace) {
		uLongf raw_len = (uLongf)raw.size();
		const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
		if (zr != Z_OK || raw_len != raw.size()) throw std::runtime_error("PNG: zlib stream is corrupt or has the wrong size");
		unfilter(raw.data(), h, row_bytes);
	} else {
		// Adam7 (PNG spec 8.2): the stream holds seven reduced images — pass p takes the pixels (x0 + i dx, y0 + j dy) — each filtered on its own; empty passes
		// (narrow images) are absent.  They are unfiltered one by one and their pixels dropped into the rows of the full image, which the converters below read.
		static const int X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
		size_t total = 0;
		int pw[7], ph[7]; size_t prb[7];
		for (int k = 0; k < 7; ++k) {
			pw[k] = (w - X0[k] + DX[k] - 1) / DX[k]; ph[k] = (h - Y0[k] + DY[k] - 1) / DY[k];
			if (pw[k] <= 0 || ph[k] <= 0) { pw[k] = ph[k] = 0; prb[k] = 0; continue; }
			prb[k] = ((size_t)pw[k] * bits_per_pixel + 7) / 8;
			total += (prb[k] + 1) * (size_t)ph[k];
		}
		std::vector<uint8_t> passes(total);
		uLongf got = (uLongf)total;
		const int zr = uncompress(passes.data(), &got, idat.data(), (uLong)idat.size());
		if (zr != Z_OK || got != total) throw std::runtime_error("PNG: zlib stream is corrupt or has the wrong size");
		size_t at = 0;
		for (int k = 0; k < 7; ++k) {
			if (!ph[k]) continue;
			uint8_t* pd = passes.data() + at;
			unfilter(pd, ph[k], prb[k]);
			for (int j = 0; j < ph[k]; ++j) {
				const uint8_t* src = pd + (size_t)j * (prb[k] + 1) + 1;
				uint8_t* dst = raw.data() + (size_t)(Y0[k] + j * DY[k]) * (row_bytes + 1) + 1;
				for (int i = 0; i < pw[k]; ++i) {
					const size_t x = (size_t)X0[k] + (size_t)i * DX[k];
					if (bits_per_pixel >= 8) memcpy(dst + x * bpp, src + (size_t)i * bpp, bpp);
					else {
						const size_t sb = (size_t)i * depth, db = x * depth;
						const uint32_t v = (src[sb >> 3] >> (8 - depth - (sb & 7))) & ((1u << depth) - 1u);
						dst[db >> 3] |= (uint8_t)(v << (8 - depth - (db & 7)));
					}
				}
			}
			at += (prb[k] + 1) * (size_t)ph[k];
		}
	}
	R.w = w; R.h = h; R.depth = depth; R.ctype = ctype; R.channels = channels; R.row_bytes = row_bytes;
}

void decode_png_rgba8(const uint8_t* d, size_t n, int& w, int& h, std::vector<uint8_t>& out) {
	PngRaw R;
	png_unfilter(d, n, R);
	w = R.w; h = R.h;
	const int depth = R.depth, ctype = R.ctype, channels = R.channels;
	const size_t row_bytes = R.row_bytes;
	const std::vector<uint8_t>& raw = R.raw; const std::vector<uint8_t>& palette = R.palette; const std::vector<uint8_t>& trns = R.trns;
	// to RGBA8
	out.assign((size_t)w * h * 4, 255);
	for (int y = 0; y < h; ++y) {
		const uint8_t* cur = raw.data() + (size_t)y * (row_bytes + 1) + 1;
		uint8_t* o = out.data() + (size_t)y * w * 4;
		for (int x = 0; x < w; ++x, o += 4) {
			auto sample = [&](int c) -> uint32_t {   // raw sample value of channel c at pixel x (not scaled)
				if (depth == 8) return cur[(size_t)x * channels + c];
				if (depth == 16) return ((uint32_t)cur[((size_t)x * channels + c) * 2] << 8) | cur[((size_t)x * channels + c) * 2 + 1];
				const size_t bit = (size_t)x * depth;
				return (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
			};
			auto to8 = [&](uint32_t v) -> uint8_t { if (depth == 16) return (uint8_t)(v >> 8); if (depth == 8) return (uint8_t)v; return (uint8_t)(v * (255u / ((1u << depth) - 1u))); };
			switch (ctype) {
				case 0: {
					const uint32_t g = sample(0);
					o[0] = o[1] = o[2] = to8(g);
					if (trns.size() >= 2 && g == (((uint32_t)trns[0] << 8) | trns[1])) o[3] = 0;
					break;
				}
				case 2: {
					const uint32_t r = sample(0), g = sample(1), b = sample(2);
					o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
					if (trns.size() >= 6 && r == (((uint32_t)trns[0] << 8) | trns[1]) && g == (((uint32_t)trns[2] << 8) | trns[3]) && b == (((uint32_t)trns[4] << 8) | trns[5])) o[3] = 0;
					break;
				}
				case 3: {
					const uint32_t i = sample(0);
					if ((size_t)i * 3 + 2 >= palette.size()) throw std::runtime_error("PNG: palette index out of range");
					o[0] = palette[i * 3]; o[1] = palette[i * 3 + 1]; o[2] = palette[i * 3 + 2];
					if (i < trns.size()) o[3] = trns[i];
					break;
				}
				case 4: o[0] = o[1] = o[2] = to8(sample(0)); o[3] = to8(sample(1)); break;
				default: o[0] = to8(sample(0)); o[1] = to8(sample(1)); o[2] = to8(sample(2)); o[3] = to8(sample(3)); break;
			}
		}
	}
}

// stbi_load_16(path, &w, &h, &comp, 1) for PNG files (the reference's depth images, src/nerf_loader.cu:636): 16-bit samples as they are, 8-bit
// (and lower) ones widened as v * 257, palette entries likewise; more than one channel is reduced like stb_image does — grey + alpha keeps the
// grey, colour becomes (77 r + 150 g + 29 b) >> 8.
void decode_png_gray16(const uint8_t* d, size_t n, int& w, int& h, std::vector<uint16_t>& out) {
	PngRaw R;
	png_unfilter(d, n, R);
	w = R.w; h = R.h;
	out.assign((size_t)w * h, 0);
	for (int y = 0; y < h; ++y) {
		const uint8_t* cur = R.raw.data() + (size_t)y * (R.row_bytes + 1) + 1;
		for (int x = 0; x < w; ++x) {
			auto sample = [&](int c) -> uint32_t {
				if (R.depth == 8) return cur[(size_t)x * R.channels + c];
				if (R.depth == 16) return ((uint32_t)cur[((size_t)x * R.channels + c) * 2] << 8) | cur[((size_t)x * R.channels + c) * 2 + 1];
				const size_t bit = (size_t)x * R.depth;
				return (cur[bit >> 3] >> (8 - R.depth - (bit & 7))) & ((1u << R.depth) - 1u);
			};
			// stb_image reduces to one channel at the file's own depth first (8-bit: (77 r + 150 g + 29 b) >> 8 on bytes), then widens 8 -> 16 as v * 257
			auto to8 = [&](uint32_t v) -> uint32_t { if (R.depth == 8) return v; return v * (255u / ((1u << R.depth) - 1u)); };
			uint32_t lum;
			if (R.depth == 16) {
				if (R.ctype == 0 || R.ctype == 4) lum = sample(0);
				else lum = (sample(0) * 77u + sample(1) * 150u + sample(2) * 29u) >> 8;
			} else {
				uint32_t y8;
				if (R.ctype == 3) {
					const uint32_t i = sample(0);
					if ((size_t)i * 3 + 2 >= R.palette.size()) throw std::runtime_error("PNG: palette index out of range");
					y8 = (R.palette[i * 3] * 77u + R.palette[i * 3 + 1] * 150u + R.palette[i * 3 + 2] * 29u) >> 8;
				} else if (R.ctype == 0 || R.ctype == 4) y8 = to8(sample(0));
				else y8 = (to8(sample(0)) * 77u + to8(sample(1)) * 150u + to8(sample(2)) * 29u) >> 8;
				lum = y8 * 257u;
			}
			out[(size_t)y * w + x] = (uint16_t)lum;
		}
	}
}

void read_png_gray16(const std::string& path, int& w, int& h, std::vector<uint16_t>& pixels) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) throw std::runtime_error("Could not load depth image " + path);
	std::vector<uint8_t> buf;
	uint8_t tmp[1 << 16];
	size_t k;
	while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
	fclose(f);
	try { decode_png_gray16(buf.data(), buf.size(), w, h, pixels); }
	catch (const std::runtime_error& e) { throw std::runtime_error("Could not load depth image " + path + ": " + e.what()); }
}

void read_png_rgba8(const std::string& path, int& w, int& h, std::vector<uint8_t>& pixels) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) throw std::runtime_error("Could not open image file: " + path);
	std::vector<uint8_t> buf;
	uint8_t tmp[1 << 16];
	size_t k;
	while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
	fclose(f);
	try { decode_png_rgba8(buf.data(), buf.size(), w, h, pixels); }
	catch (const std::runtime_error& e) { throw std::runtime_error("Could not open image file " + path + ": " + e.what()); }
}

} // namespace ngp
