// exr_reader.cpp — OpenEXR scanline and tiled files -> interleaved RGBA float32.  The reference reads .exr through the vendored tinyexr
// (src/tinyexr_wrapper.cu:62-112 `load_exr`, 114-192 `load_exr_to_gpu`; call sites src/nerf_loader.cu:541, 575 and src/testbed_image.cu:386);
// this is a build-owned reader of the published file layout, over zlib which the host already links for PNG:
//   magic 0x01312f76, version word, attribute list (name\0 type\0 size value) ending with an empty name, the block offset table, then per block
//   {int32 y, int32 bytes, data}.  A block holds 1 (NONE, RLE, ZIPS) or 16 (ZIP) scanlines; inside a block every scanline stores its channels one
//   after the other in the alphabetical order of the channel list (A, B, G, R), each as `width` little-endian HALF / FLOAT / UINT values.
//   RLE and ZIP data went through a byte predictor (d[i] += d[i-1] - 128) and an even / odd byte split before compression.
//   PIZ (blocks of 32 scanlines): {u16 min, u16 max, bitmap bytes [min, max] of the 16-bit values in use, i32 n, n bytes of Huffman code}; the decoded 16-bit words
//   are, channel after channel, the block's samples (a FLOAT / UINT sample = two words), Haar-wavelet transformed in 2-D per word plane and mapped through the
//   bitmap's rank table — the published scheme of OpenEXR's ImfPizCompressor / ImfHuf / ImfWav, restated here (piz_*).
//   Tiled files (version bit 0x200, attribute `tiles`): the offset table lists the tiles of level (0, 0) first; a tile block is {i32 tile x, tile y, level x,
//   level y, i32 bytes, data} and holds the tile's scanlines (cropped at the image edge) in the scanline layout, compressed as ONE block.  Only level (0, 0) is read.
// Multi-part, deep and PXR24 / B44 / DWA files are rejected with an error that says so.
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

// ---- PIZ ----------------------------------------------------------------------------------------------------------------------------------------
constexpr int HUF_ENCSIZE = (1 << 16) + 1;   // symbols 0 .. 65535 and the run-length symbol
struct BitReader {
	const uint8_t* p; const uint8_t* end; uint64_t c = 0; int lc = 0; uint64_t used = 0;
	uint32_t get(int n) {   // MSB first
		while (lc < n) { if (p >= end) throw std::runtime_error{"EXR: PIZ Huffman data ends early"}; c = (c << 8) | *p++; lc += 8; }
		lc -= n; used += (uint64_t)n;
		return (uint32_t)((c >> lc) & ((1ull << n) - 1ull));
	}
};

// Huffman stage: 20-byte header {first symbol, last symbol (= the run-length symbol), table bytes, code bits, 0}, the code LENGTHS of the symbols first .. last in
// 6 bits each (59 .. 62: a run of 2 .. 5 zero lengths, 63: a run of 6 + next 8 bits), then the code.  The codes are canonical: within a length consecutive in symbol
// order, and the first code of length l - 1 is (first code of l + number of codes of l) >> 1 starting from 0 at length 58.
void piz_huf_uncompress(const uint8_t* src, size_t n_src, uint16_t* out, size_t n_out) {
	if (n_out == 0) return;
	if (n_src < 20) throw std::runtime_error{"EXR: PIZ Huffman header is truncated"};
	auto u32 = [&](size_t o) { uint32_t v; memcpy(&v, src + o, 4); return v; };
	const uint32_t im = u32(0), iM = u32(4), n_bits = u32(12);
	if (im >= (uint32_t)HUF_ENCSIZE || iM >= (uint32_t)HUF_ENCSIZE || im > iM) throw std::runtime_error{"EXR: PIZ Huffman symbol range is invalid"};
	std::vector<uint8_t> len((size_t)HUF_ENCSIZE, 0);
	BitReader tr{src + 20, src + n_src};
	for (uint32_t s = im; s <= iM; ++s) {
		const uint32_t l = tr.get(6);
		if (l >= 59) {
			const uint32_t run = l == 63 ? tr.get(8) + 6u : l - 59u + 2u;
			if (s + run > iM + 1) throw std::runtime_error{"EXR: PIZ Huffman table run passes the last symbol"};
			s += run - 1;   // (lengths are zero already)
		} else len[s] = (uint8_t)l;
	}
	const uint8_t* code_begin = tr.p;   // the table is read in whole bytes
	if ((uint64_t)(src + n_src - code_begin) * 8u < (uint64_t)n_bits) throw std::runtime_error{"EXR: PIZ Huffman code is shorter than its header says"};
	uint64_t count[59] = {0}, first[59] = {0};
	for (int s = 0; s < HUF_ENCSIZE; ++s) ++count[len[(size_t)s]];
	{
		uint64_t c = 0;
		for (int l = 58; l > 0; --l) { const uint64_t nc = (c + count[l]) >> 1; first[l] = c; c = nc; }
	}
	// symbols by (length, symbol): the k-th code of length l is first[l] + k
	uint32_t start[60]; start[1] = 0;
	for (int l = 1; l < 59; ++l) start[l + 1] = start[l] + (uint32_t)count[l];
	std::vector<uint32_t> sym(start[59]);
	{
		uint32_t fill[59]; for (int l = 1; l < 59; ++l) fill[l] = start[l];
		for (int s = 0; s < HUF_ENCSIZE; ++s) if (len[(size_t)s]) sym[fill[len[(size_t)s]]++] = (uint32_t)s;
	}
	constexpr int FAST = 12;   // direct table for codes up to 12 bits: (symbol << 4 | length), 0 = longer code
	std::vector<uint32_t> fast((size_t)1 << FAST, 0);
	for (int l = 1; l <= FAST; ++l)
		for (uint64_t k = 0; k < count[l]; ++k) {
			const uint64_t code = first[l] + k;
			if (code >> l) throw std::runtime_error{"EXR: PIZ Huffman code lengths are not a prefix code"};
			const uint32_t entry = (sym[start[l] + (uint32_t)k] << 6) | (uint32_t)l;
			for (uint64_t f = code << (FAST - l), e = (code + 1) << (FAST - l); f < e; ++f) fast[(size_t)f] = entry;
		}
	BitReader br{code_begin, src + n_src};
	size_t o = 0;
	while (o < n_out) {
		uint32_t s = 0; bool found = false;
		// peek FAST bits (zero padded at the end of the data)
		while (br.lc < FAST && br.p < br.end) { br.c = (br.c << 8) | *br.p++; br.lc += 8; }
		const uint32_t peek = br.lc >= FAST ? (uint32_t)((br.c >> (br.lc - FAST)) & ((1u << FAST) - 1u)) : (uint32_t)((br.c << (FAST - br.lc)) & ((1u << FAST) - 1u));
		const uint32_t e = fast[peek];
		if (e && (int)(e & 63u) <= br.lc) { br.lc -= (int)(e & 63u); br.used += e & 63u; s = e >> 6; found = true; }
		else {
			uint64_t v = 0;
			for (int l = 1; l < 59; ++l) {
				v = (v << 1) | br.get(1);
				if (count[l] && v >= first[l] && v - first[l] < count[l]) { s = sym[start[l] + (uint32_t)(v - first[l])]; found = true; break; }
			}
		}
		if (!found) throw std::runtime_error{"EXR: PIZ Huffman code is invalid"};
		if (s == iM) {   // run: repeat the previous word (next 8 bits) times
			const uint32_t run = br.get(8);
			if (o == 0 || o + run > n_out) throw std::runtime_error{"EXR: PIZ Huffman run is out of range"};
			const uint16_t v = out[o - 1];
			for (uint32_t k = 0; k < run; ++k) out[o++] = v;
		} else out[o++] = (uint16_t)s;
	}
	if (br.used > (uint64_t)n_bits) throw std::runtime_error{"EXR: PIZ Huffman code is longer than its header says"};
}

// inverse of the 2-D Haar-like wavelet ("wav2Decode"): plane of nx x ny words, element stride ox, row stride oy; values below 2^14 use the exact integer
// average / difference pair, larger ranges the modulo-2^16 variant
inline void piz_wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
	const int ls = (int16_t)l, hs = (int16_t)h;
	const int ai = ls + (hs & 1) + (hs >> 1);
	a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hs);
}
inline void piz_wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
	const int m = l, d = h;
	const int bb = (m - (d >> 1)) & 0xffff;
	const int aa = (d + bb - 0x8000) & 0xffff;
	b = (uint16_t)bb; a = (uint16_t)aa;
}
void piz_wav_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t max_value) {
	const bool w14 = max_value < (1 << 14);
	const int n = nx > ny ? ny : nx;
	int p = 1;
	while (p <= n) p <<= 1;
	p >>= 1;
	int p2 = p;
	p >>= 1;
	auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) piz_wdec14(l, h, a, b); else piz_wdec16(l, h, a, b); };
	while (p >= 1) {
		const int oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
		int y = 0;
		for (; y <= ny - p2; y += p2) {
			uint16_t* row = in + (size_t)y * oy;
			int x = 0;
			for (; x <= nx - p2; x += p2) {
				uint16_t* px = row + (size_t)x * ox; uint16_t* p01 = px + ox1; uint16_t* p10 = px + oy1; uint16_t* p11 = p10 + ox1;
				uint16_t i00, i01, i10, i11;
				dec(*px, *p10, i00, i10); dec(*p01, *p11, i01, i11);
				dec(i00, i01, *px, *p01); dec(i10, i11, *p10, *p11);
			}
			if (nx & p) { uint16_t* px = row + (size_t)x * ox; uint16_t* p10 = px + oy1; uint16_t i00; dec(*px, *p10, i00, *p10); *px = i00; }
		}
		if (ny & p) {
			uint16_t* row = in + (size_t)y * oy;
			for (int x = 0; x <= nx - p2; x += p2) { uint16_t* px = row + (size_t)x * ox; uint16_t* p01 = px + ox1; uint16_t i00; dec(*px, *p01, i00, *p01); *px = i00; }
		}
		(void)oy2; (void)ox2;
		p2 = p; p >>= 1;
	}
}

// one PIZ block of nx x ny pixels -> the scanline layout (per scanline: per channel nx samples)
void piz_decompress(const uint8_t* src, size_t n_src, std::vector<uint8_t>& dst, size_t n_expected, int nx, int ny, const std::vector<Channel>& channels) {
	if (n_src < 4) throw std::runtime_error{"EXR: PIZ block is truncated"};
	uint16_t min_nz, max_nz;
	memcpy(&min_nz, src, 2); memcpy(&max_nz, src + 2, 2);
	std::vector<uint8_t> bitmap(8192, 0);
	size_t at = 4;
	if (min_nz <= max_nz) {
		if (max_nz >= 8192) throw std::runtime_error{"EXR: PIZ bitmap range is invalid"};
		const size_t nb = (size_t)max_nz - min_nz + 1;
		if (n_src < at + nb) throw std::runtime_error{"EXR: PIZ block is truncated"};
		memcpy(bitmap.data() + min_nz, src + at, nb);
		at += nb;
	}
	std::vector<uint16_t> lut(65536, 0);
	uint32_t k = 0;
	for (uint32_t i = 0; i < 65536; ++i) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7u)))) lut[k++] = (uint16_t)i;
	const uint16_t max_value = (uint16_t)(k - 1);
	if (n_src < at + 4) throw std::runtime_error{"EXR: PIZ block is truncated"};
	int32_t n_huf; memcpy(&n_huf, src + at, 4); at += 4;
	if (n_huf < 0 || (size_t)n_huf > n_src - at) throw std::runtime_error{"EXR: PIZ Huffman data passes the end of the block"};
	size_t n_words = 0;
	for (const Channel& ch : channels) n_words += (size_t)nx * ny * (ch.type == 1 ? 1 : 2);
	if (n_words * 2 != n_expected) throw std::runtime_error{"EXR: PIZ block size mismatch"};
	std::vector<uint16_t> words(n_words);
	piz_huf_uncompress(src + at, (size_t)n_huf, words.data(), n_words);
	size_t off = 0;
	std::vector<size_t> ch_start(channels.size());
	for (size_t c = 0; c < channels.size(); ++c) {
		const int size = channels[c].type == 1 ? 1 : 2;
		ch_start[c] = off;
		for (int j = 0; j < size; ++j) piz_wav_decode(words.data() + off + j, nx, size, ny, nx * size, max_value);
		off += (size_t)nx * ny * size;
	}
	for (uint16_t& v : words) v = lut[v];
	dst.resize(n_expected);
	uint8_t* o = dst.data();
	for (int y = 0; y < ny; ++y)
		for (size_t c = 0; c < channels.size(); ++c) {
			const size_t n = (size_t)nx * (channels[c].type == 1 ? 1 : 2);
			memcpy(o, words.data() + ch_start[c] + (size_t)y * n, n * 2);
			o += n * 2;
		}
}
} // namespace

void decode_exr_rgba_f32(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<float>& rgba) {
	Cursor c{data, data + n_bytes};
	if (c.u32() != 20000630u) throw std::runtime_error{"EXR: bad magic number"};
	const uint32_t version = c.u32();
	if ((version & 0xff) != 2) throw std::runtime_error{"EXR: unsupported file version"};
	if (version & 0x1800) throw std::runtime_error{"EXR: deep / multi-part files are not supported by this reader"};
	const bool tiled = (version & 0x200) != 0;
	std::vector<Channel> channels;
	int compression = -1, dw[4] = {0, 0, -1, -1};
	uint32_t tile_w = 0, tile_h = 0;
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
		else if (name == "tiles") { tile_w = v.u32(); tile_h = v.u32(); }   // (the level mode byte follows: only level (0, 0), which comes first, is read)
	}
	static const char* comp_names[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
	if (compression < 0 || compression > 4) {
		throw std::runtime_error{std::string{"EXR: compression "} + (compression >= 0 && compression < 10 ? comp_names[compression] : "?") + " is not supported by this reader (NONE, RLE, ZIPS, ZIP, PIZ are)"};
	}
	// the window comes from the file: 64-bit arithmetic and a cap before anything is sized from it
	const int64_t w64 = (int64_t)dw[2] - dw[0] + 1, h64 = (int64_t)dw[3] - dw[1] + 1;
	if (w64 <= 0 || h64 <= 0 || channels.empty()) throw std::runtime_error{"EXR: empty data window or channel list"};
	if (w64 > 65536 || h64 > 65536 || w64 * h64 > ((int64_t)1 << 28)) throw std::runtime_error{"EXR: data window larger than 2^28 pixels"};
	w = (int)w64; h = (int)h64;
	if (tiled && (tile_w == 0 || tile_h == 0 || tile_w > 65536 || tile_h > 65536)) throw std::runtime_error{"EXR: tiled file without a valid `tiles` attribute"};
	size_t bytes_per_pixel = 0;
	for (const Channel& ch : channels) bytes_per_pixel += ch.type == 1 ? 2 : 4;
	// RGBA by name (tinyexr's LoadEXR rule); a lone luminance channel is replicated; a missing alpha is 1
	int slot[4] = {-1, -1, -1, -1};
	for (size_t k = 0; k < channels.size(); ++k) {
		const std::string& n = channels[k].name;
		if (n == "R") slot[0] = (int)k; else if (n == "G") slot[1] = (int)k; else if (n == "B") slot[2] = (int)k; else if (n == "A") slot[3] = (int)k;
	}
	if (slot[0] < 0 && channels.size() == 1) slot[0] = slot[1] = slot[2] = 0;
	if (slot[0] < 0 || slot[1] < 0 || slot[2] < 0) throw std::runtime_error{"EXR: no R, G, B channels"};
	{
		// before any allocation sized by the header: the file must at least hold the offset table of its blocks (8 bytes each) and one block header per entry — a
		// few-byte header cannot force a 4 GiB pixel buffer or a 2 GiB offset table (1 x 1 tiles)
		const uint64_t n_blk = tiled ? (uint64_t)((w64 + tile_w - 1) / tile_w) * (uint64_t)((h64 + tile_h - 1) / tile_h)
		                             : (uint64_t)((h64 + (compression == 3 ? 16 : compression == 4 ? 32 : 1) - 1) / (compression == 3 ? 16 : compression == 4 ? 32 : 1));
		const uint64_t left = (uint64_t)(c.end - c.p);
		if (n_blk > left / (tiled ? 28u : 16u)) throw std::runtime_error{"EXR: the file is too short for the offset table and block headers its data window implies"};
		// ... nor can its payload expand beyond what DEFLATE (1032 : 1 at best), RLE (128 : 1) or PIZ could have packed into the file
		if ((uint64_t)w64 * (uint64_t)h64 * bytes_per_pixel / 2048u > (uint64_t)n_bytes) throw std::runtime_error{"EXR: data window far larger than the file could encode"};
	}
	rgba.assign((size_t)w * h * 4, 1.0f);
	std::vector<uint8_t> block, tmp;
	// one compressed block of nx x ny pixels at (x0, y0): decompress to the scanline layout, then pick R, G, B, A
	auto take_block = [&](const uint8_t* src, size_t n_src, int x0, int y0, int nx, int ny) {
		const size_t line_bytes = bytes_per_pixel * (size_t)nx;
		const size_t n_raw = line_bytes * (size_t)ny;
		const uint8_t* raw = src;
		if (compression != 0 && n_src < n_raw) {
			if (compression == 4) piz_decompress(src, n_src, block, n_raw, nx, ny, channels);
			else {
				if (compression == 1) unrle_block(src, n_src, block, n_raw); else inflate_block(src, n_src, block, n_raw);
				undo_predictor_and_split(block, tmp);
			}
			raw = block.data();
		} else if (n_src != n_raw) {
			throw std::runtime_error{"EXR: uncompressed block has the wrong size"};
		}
		std::vector<size_t> ch_offset(channels.size());
		{ size_t o = 0; for (size_t k = 0; k < channels.size(); ++k) { ch_offset[k] = o; o += (size_t)nx * (channels[k].type == 1 ? 2 : 4); } }
		for (int l = 0; l < ny; ++l) {
			const uint8_t* line = raw + line_bytes * (size_t)l;
			float* dst = rgba.data() + ((size_t)(y0 + l) * w + x0) * 4;
			for (int s4 = 0; s4 < 4; ++s4) {
				if (slot[s4] < 0) continue;
				const Channel& ch = channels[(size_t)slot[s4]];
				const uint8_t* src_ch = line + ch_offset[(size_t)slot[s4]];
				for (int x = 0; x < nx; ++x) {
					float v;
					if (ch.type == 1) { uint16_t hb; memcpy(&hb, src_ch + 2 * x, 2); v = half_bits_to_float(hb); }
					else if (ch.type == 2) memcpy(&v, src_ch + 4 * x, 4);
					else { uint32_t u; memcpy(&u, src_ch + 4 * x, 4); v = (float)u; }
					dst[4 * x + s4] = v;
				}
			}
		}
	};
	if (tiled) {
		const int ntx = (int)((w + (int64_t)tile_w - 1) / tile_w), nty = (int)((h + (int64_t)tile_h - 1) / tile_h);
		std::vector<uint64_t> offsets((size_t)ntx * nty);
		for (auto& o : offsets) o = c.u64();
		for (size_t t = 0; t < offsets.size(); ++t) {
			if (n_bytes < 20 || offsets[t] > n_bytes - 20) throw std::runtime_error{"EXR: tile offset out of range"};
			Cursor bc{data + offsets[t], data + n_bytes};
			const int tx = bc.i32(), ty = bc.i32(), lx = bc.i32(), ly = bc.i32();
			const uint32_t n_src = bc.u32();
			bc.need(n_src);
			if (lx != 0 || ly != 0) throw std::runtime_error{"EXR: the first tiles of the file are not those of level (0, 0)"};
			if (tx < 0 || ty < 0 || tx >= ntx || ty >= nty) throw std::runtime_error{"EXR: tile outside the data window"};
			const int x0 = tx * (int)tile_w, y0 = ty * (int)tile_h;
			take_block(bc.p, n_src, x0, y0, std::min((int)tile_w, w - x0), std::min((int)tile_h, h - y0));
		}
		return;
	}
	const int lines_per_block = compression == 3 ? 16 : compression == 4 ? 32 : 1;
	const int n_blocks = (h + lines_per_block - 1) / lines_per_block;
	std::vector<uint64_t> offsets((size_t)n_blocks);
	for (auto& o : offsets) o = c.u64();
	for (int b = 0; b < n_blocks; ++b) {
		if (n_bytes < 8 || offsets[(size_t)b] > n_bytes - 8) throw std::runtime_error{"EXR: block offset out of range"};
		Cursor bc{data + offsets[(size_t)b], data + n_bytes};
		const int y0 = bc.i32() - dw[1];
		const uint32_t n_src = bc.u32();
		bc.need(n_src);
		if (y0 < 0 || y0 >= h) throw std::runtime_error{"EXR: block outside the data window"};
		take_block(bc.p, n_src, 0, y0, w, std::min(lines_per_block, h - y0));
	}
}

void read_exr_rgba_f32(const std::string& path, int& w, int& h, std::vector<float>& rgba) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error{"Could not open image file: " + path};
	std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	decode_exr_rgba_f32(bytes.data(), bytes.size(), w, h, rgba);
}

} // namespace ngp
