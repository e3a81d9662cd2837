// exr_reader.cpp — OpenEXR scanline files -> interleaved RGBA float32.  The reference reads .exr through the vendored tinyexr
// (src/tinyexr_wrapper.cu:62-112 `load_exr`, 114-192 `load_exr_to_gpu`; call sites src/nerf_loader.cu:541, 575 and src/testbed_image.cu:386);
// this is a build-owned reader of the published file layout, over zlib which the host already links for PNG:
//   magic 0x01312f76, version word, attribute list (name\0 type\0 size value) ending with an empty name, the block offset table, then per block
//   {int32 y, int32 bytes, data}.  A block holds 1 (NONE, RLE, ZIPS) or 16 (ZIP) scanlines; inside a block every scanline stores its channels one
//   after the other in the alphabetical order of the channel list (A, B, G, R), each as `width` little-endian HALF / FLOAT / UINT values.
//   RLE and ZIP data went through a byte predictor (d[i] += d[i-1] - 128) and an even / odd byte split before compression.
// Tiled, multi-part, deep and PIZ / PXR24 / B44 / DWA files are rejected with an error that says so.
#include "image_io.h"

#include <zlib.h>

#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>

#include "snapshot.h"   // half_bits_to_float

namespace ngp {

namespace {
struct Channel { std::string name; int type; int xs, ys; };   // type: 0 UINT, 1 HALF, 2 FLOAT

struct Cursor {
	const uint8_t* p; const uint8_t* end;
	void need(size_t n) const { if ((size_t)(end - p) < n) throw std::runtime_error{"EXR: truncated file"}; }
	uint32_t u32() { need(4); uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
	int32_t i32() { return (int32_t)u32(); }
	uint64_t u64() { need(8); uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
	std::string str() { const uint8_t* s = p; while (p < end && *p) ++p; if (p >= end) throw std::runtime_error{"EXR: unterminated string"}; std::string r((const char*)s, (size_t)(p - s)); ++p; return r; }
};

void undo_predictor_and_split(std::vector<uint8_t>& buf, std::vector<uint8_t>& tmp) {
	const size_t n = buf.size();
	for (size_t i = 1; i < n; ++i) buf[i] = (uint8_t)(buf[i - 1] + buf[i] - 128);
	tmp.resize(n);
	const size_t half = (n + 1) / 2;
	for (size_t i = 0, a = 0, b = half; i < n; ) { tmp[i++] = buf[a++]; if (i < n) tmp[i++] = buf[b++]; }
	buf.swap(tmp);
}

void inflate_block(const uint8_t* src, size_t n_src, std::vector<uint8_t>& dst, size_t n_expected) {
	dst.resize(n_expected);
	uLongf n = (uLongf)n_expected;
	if (uncompress(dst.data(), &n, src, (uLong)n_src) != Z_OK || n != n_expected) throw std::runtime_error{"EXR: zlib block does not inflate to the expected size"};
}

void unrle_block(const uint8_t* src, size_t n_src, std::vector<uint8_t>& dst, size_t n_expected) {
	dst.clear(); dst.reserve(n_expected);
	size_t i = 0;
	while (i < n_src) {
		const int8_t c = (int8_t)src[i++];
		if (c < 0) { const size_t n = (size_t)(-(int)c); if (i + n > n_src) throw std::runtime_error{"EXR: bad RLE run"}; dst.insert(dst.end(), src + i, src + i + n); i += n; }
		else { if (i >= n_src) throw std::runtime_error{"EXR: bad RLE run"}; dst.insert(dst.end(), (size_t)c + 1, src[i++]); }
		if (dst.size() > n_expected) throw std::runtime_error{"EXR: RLE block expands past the expected size"};   // a crafted block must not grow without bound
	}
	if (dst.size() != n_expected) throw std::runtime_error{"EXR: RLE block does not expand to the expected size"};
}
} // namespace

void decode_exr_rgba_f32(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<float>& rgba) {
	Cursor c{data, data + n_bytes};
	if (c.u32() != 20000630u) throw std::runtime_error{"EXR: bad magic number"};
	const uint32_t version = c.u32();
	if ((version & 0xff) != 2) throw std::runtime_error{"EXR: unsupported file version"};
	if (version & 0x200) throw std::runtime_error{"EXR: tiled files are not supported by this reader (scanline files are)"};
	if (version & 0x1800) throw std::runtime_error{"EXR: deep / multi-part files are not supported by this reader"};
	std::vector<Channel> channels;
	int compression = -1, dw[4] = {0, 0, -1, -1};
	while (true) {
		const std::string name = c.str();
		if (name.empty()) break;
		const std::string type = c.str();
		const uint32_t size = c.u32();
		c.need(size);
		Cursor v{c.p, c.p + size};
		c.p += size;
		if (name == "channels") {
			while (v.p < v.end && *v.p) {
				Channel ch; ch.name = v.str(); ch.type = v.i32(); v.need(4); v.p += 4; ch.xs = v.i32(); ch.ys = v.i32();
				if (ch.xs != 1 || ch.ys != 1) throw std::runtime_error{"EXR: subsampled channels are not supported"};
				if (ch.type < 0 || ch.type > 2) throw std::runtime_error{"EXR: unknown channel type"};
				channels.push_back(ch);
			}
		} else if (name == "compression") { v.need(1); compression = *v.p; }
		else if (name == "dataWindow") { for (int k = 0; k < 4; ++k) dw[k] = v.i32(); }
	}
	static const char* comp_names[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
	if (compression < 0 || compression > 3) {
		throw std::runtime_error{std::string{"EXR: compression "} + (compression >= 0 && compression < 10 ? comp_names[compression] : "?") + " is not supported by this reader (NONE, RLE, ZIPS, ZIP are)"};
	}
	// the window comes from the file: 64-bit arithmetic and a cap before anything is sized from it
	const int64_t w64 = (int64_t)dw[2] - dw[0] + 1, h64 = (int64_t)dw[3] - dw[1] + 1;
	if (w64 <= 0 || h64 <= 0 || channels.empty()) throw std::runtime_error{"EXR: empty data window or channel list"};
	if (w64 > 65536 || h64 > 65536 || w64 * h64 > ((int64_t)1 << 28)) throw std::runtime_error{"EXR: data window larger than 2^28 pixels"};
	w = (int)w64; h = (int)h64;
	const int lines_per_block = compression == 3 ? 16 : 1;
	const int n_blocks = (h + lines_per_block - 1) / lines_per_block;
	size_t bytes_per_line = 0;
	std::vector<size_t> ch_offset(channels.size());
	for (size_t k = 0; k < channels.size(); ++k) { ch_offset[k] = bytes_per_line; bytes_per_line += (size_t)w * (channels[k].type == 1 ? 2 : 4); }
	// RGBA by name (tinyexr's LoadEXR rule); a lone luminance channel is replicated; a missing alpha is 1
	int slot[4] = {-1, -1, -1, -1};
	for (size_t k = 0; k < channels.size(); ++k) {
		const std::string& n = channels[k].name;
		if (n == "R") slot[0] = (int)k; else if (n == "G") slot[1] = (int)k; else if (n == "B") slot[2] = (int)k; else if (n == "A") slot[3] = (int)k;
	}
	if (slot[0] < 0 && channels.size() == 1) slot[0] = slot[1] = slot[2] = 0;
	if (slot[0] < 0 || slot[1] < 0 || slot[2] < 0) throw std::runtime_error{"EXR: no R, G, B channels"};
	std::vector<uint64_t> offsets((size_t)n_blocks);
	for (auto& o : offsets) o = c.u64();
	rgba.assign((size_t)w * h * 4, 1.0f);
	std::vector<uint8_t> block, tmp;
	for (int b = 0; b < n_blocks; ++b) {
		if (n_bytes < 8 || offsets[(size_t)b] > n_bytes - 8) throw std::runtime_error{"EXR: block offset out of range"};
		Cursor bc{data + offsets[(size_t)b], data + n_bytes};
		const int y0 = bc.i32() - dw[1];
		const uint32_t n_src = bc.u32();
		bc.need(n_src);
		if (y0 < 0 || y0 >= h) throw std::runtime_error{"EXR: block outside the data window"};
		const int n_lines = std::min(lines_per_block, h - y0);
		const size_t n_raw = bytes_per_line * (size_t)n_lines;
		const uint8_t* raw = bc.p;
		if (compression != 0 && n_src < n_raw) {
			if (compression == 1) unrle_block(bc.p, n_src, block, n_raw); else inflate_block(bc.p, n_src, block, n_raw);
			undo_predictor_and_split(block, tmp);
			raw = block.data();
		} else if (n_src != n_raw) {
			throw std::runtime_error{"EXR: uncompressed block has the wrong size"};
		}
		for (int l = 0; l < n_lines; ++l) {
			const uint8_t* line = raw + bytes_per_line * (size_t)l;
			float* dst = rgba.data() + (size_t)(y0 + l) * w * 4;
			for (int s = 0; s < 4; ++s) {
				if (slot[s] < 0) continue;
				const Channel& ch = channels[(size_t)slot[s]];
				const uint8_t* src = line + ch_offset[(size_t)slot[s]];
				for (int x = 0; x < w; ++x) {
					float v;
					if (ch.type == 1) { uint16_t hb; memcpy(&hb, src + 2 * x, 2); v = half_bits_to_float(hb); }
					else if (ch.type == 2) memcpy(&v, src + 4 * x, 4);
					else { uint32_t u; memcpy(&u, src + 4 * x, 4); v = (float)u; }
					dst[4 * x + s] = v;
				}
			}
		}
	}
}

void read_exr_rgba_f32(const std::string& path, int& w, int& h, std::vector<float>& rgba) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error{"Could not open image file: " + path};
	std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	decode_exr_rgba_f32(bytes.data(), bytes.size(), w, h, rgba);
}

} // namespace ngp
