// jpeg_reader.cpp — JPEG (JFIF / Exif) -> RGBA8 for the transforms.json loader.  The reference decodes every non-EXR training image with the vendored
// stb_image (`stbi_load(path, &w, &h, &comp, 4)`, src/nerf_loader.cu:581) — data/nerf/fox, BASELINE config #2, is 50 .jpg frames.  This is a build-owned
// decoder of ITU-T T.81: baseline and extended sequential DCT (SOF0 / SOF1) and progressive DCT (SOF2: spectral selection + successive approximation),
// Huffman coding, 8-bit samples, restart intervals, 1 (grey) or 3 (YCbCr, or RGB with an Adobe transform-0 marker) components with sampling factors 1..2
// (4:4:4, 4:2:2, 4:4:0, 4:2:0), chroma brought to full resolution with the 3:1 "triangle" filter both stb_image and libjpeg use.  The inverse DCT is an exact
// separable float transform rounded once, so pixel values agree with libjpeg / stb_image to +-1..2 code values (their integer IDCTs round in between);
// tests/test_image_io_cpu.py holds it to PIL.  Arithmetic coding, 12-bit, lossless, hierarchical and CMYK files are rejected.
#include "image_io.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace ngp {

namespace {

const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
	bool present = false;
	uint8_t fast_len[512];       // 9-bit prefix -> code length (0: longer than 9 bits)
	uint8_t fast_sym[512];
	int32_t maxcode[18];         // largest code of each length, left-aligned to 16 bits, +1
	int32_t delta[17];           // symbol index = (code >> (16 - len)) + delta[len]
	uint8_t symbols[256];
	int n_symbols = 0;
	void build(const uint8_t counts[16], const uint8_t* syms, int n_syms) {
		// validate BEFORE anything is written: an over-subscribed table (e.g. three codes of length 1) would index past the 512-entry prefix
		// tables below — every count and symbol comes from the file
		int total = 0, next_code = 0;
		for (int len = 1; len <= 16; ++len) {
			total += counts[len - 1];
			next_code += counts[len - 1];
			if (next_code > (1 << len)) throw std::runtime_error{"JPEG: bad Huffman table (over-subscribed code lengths)"};
			next_code <<= 1;
		}
		if (total != n_syms || n_syms > 256) throw std::runtime_error{"JPEG: bad Huffman table"};
		memcpy(symbols, syms, (size_t)n_syms);
		n_symbols = n_syms;
		memset(fast_len, 0, sizeof(fast_len));
		int code = 0, k = 0;
		for (int len = 1; len <= 16; ++len) {
			delta[len] = k - code;
			for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
				if (len <= 9) {
					const int first = code << (9 - len), n = 1 << (9 - len);
					for (int j = 0; j < n; ++j) { fast_len[first + j] = (uint8_t)len; fast_sym[first + j] = syms[k]; }
				}
			}
			if (code > (1 << len)) throw std::runtime_error{"JPEG: bad Huffman table"};
			maxcode[len] = code << (16 - len);
			code <<= 1;
		}
		maxcode[17] = 0x7fffffff;
		present = true;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
	int blocks_w = 0, blocks_h = 0;     // allocated (padded to whole MCUs)
	int plane_w = 0, plane_h = 0;       // ceil(width * h / hmax), ceil(height * v / vmax): the blocks a non-interleaved scan covers
	int pred = 0;
	std::vector<int16_t> coef;          // blocks_w * blocks_h * 64, natural order
	std::vector<uint8_t> plane;         // blocks_w * 8 x blocks_h * 8 samples after the IDCT
};

class Decoder {
public:
	Decoder(const uint8_t* data, size_t n) : d(data), n(n) {}
	void decode(int& w, int& h, std::vector<uint8_t>& rgba);

private:
	const uint8_t* d; size_t n, pos = 0;
	int width = 0, height = 0, n_comp = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
	bool progressive = false, have_frame = false;
	int adobe_transform = -1;
	Component comp[3];
	uint16_t qt[4][64];    // natural order
	bool qt_present[4] = {false, false, false, false};
	HuffTable hdc[4], hac[4];
	int restart_interval = 0;
	// bit reader over the entropy-coded segment
	uint32_t bitbuf = 0; int bitcnt = 0; bool hit_marker = false;
	int eobrun = 0;

	uint8_t u8() { if (pos >= n) throw std::runtime_error{"JPEG: truncated file"}; return d[pos++]; }
	int u16() { const int a = u8(); return (a << 8) | u8(); }

	void fill() {
		while (bitcnt <= 24) {
			uint32_t b = 0;
			if (!hit_marker && pos < n) {
				b = d[pos];
				if (b == 0xFF) {
					const uint8_t nx = pos + 1 < n ? d[pos + 1] : 0xD9;
					if (nx == 0) pos += 2;                      // stuffed zero
					else { hit_marker = true; b = 0; }          // a marker ends the segment: feed zeros
				} else ++pos;
			}
			bitbuf |= b << (24 - bitcnt);
			bitcnt += 8;
		}
	}
	int getbits(int k) { if (k == 0) return 0; if (bitcnt < k) fill(); const int v = (int)(bitbuf >> (32 - k)); bitbuf <<= k; bitcnt -= k; return v; }
	int getbit() { if (bitcnt < 1) fill(); const int v = (int)(bitbuf >> 31); bitbuf <<= 1; bitcnt -= 1; return v; }
	int decode_symbol(const HuffTable& t) {
		if (bitcnt < 16) fill();
		const int look = (int)(bitbuf >> 23);
		int len = t.fast_len[look];
		if (len) { bitbuf <<= len; bitcnt -= len; return t.fast_sym[look]; }
		const int32_t code16 = (int32_t)(bitbuf >> 16);
		for (len = 10; len <= 16; ++len) if (code16 < t.maxcode[len]) break;
		if (len > 16) throw std::runtime_error{"JPEG: bad Huffman code"};
		const int idx = (code16 >> (16 - len)) + t.delta[len];
		if (idx < 0 || idx >= t.n_symbols) throw std::runtime_error{"JPEG: bad Huffman code"};
		bitbuf <<= len; bitcnt -= len;
		return t.symbols[idx];
	}
	// s: the magnitude category of T.81 F.1.2.1 — at most 11 for DC differences, 10 for AC coefficients of 8-bit data (Tables F.1 / F.2);
	// the callers check that bound, 16 is what the 32-bit reader can deliver
	int receive_extend(int s) {
		if (s == 0) return 0;
		if (s < 0 || s > 16) throw std::runtime_error{"JPEG: bad magnitude category"};
		const int v = getbits(s);
		return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
	}
	int dc_category(const HuffTable& t) { const int c = decode_symbol(t); if (c > 11) throw std::runtime_error{"JPEG: DC difference category above 11"}; return c; }
	static int ac_size(int s) { if (s > 10) throw std::runtime_error{"JPEG: AC coefficient size above 10"}; return s; }
	void reset_entropy() { bitbuf = 0; bitcnt = 0; hit_marker = false; eobrun = 0; for (auto& c : comp) c.pred = 0; }

	void parse_dqt(); void parse_dht(); void parse_sof(int marker); void parse_sos();
	void decode_block(Component& c, int16_t* blk, int ss, int se, int ah, int al);
	void finish(int& w, int& h, std::vector<uint8_t>& rgba);
};

void Decoder::parse_dqt() {
	int len = u16() - 2;
	while (len > 0) {
		const int pq_tq = u8(); const int pq = pq_tq >> 4, tq = pq_tq & 15;
		if (tq > 3 || pq > 1) throw std::runtime_error{"JPEG: bad quantisation table"};
		for (int i = 0; i < 64; ++i) qt[tq][ZIGZAG[i]] = (uint16_t)(pq ? u16() : u8());
		qt_present[tq] = true;
		len -= 1 + 64 * (pq + 1);
	}
}

void Decoder::parse_dht() {
	int len = u16() - 2;
	while (len > 0) {
		const int tc_th = u8(); const int tc = tc_th >> 4, th = tc_th & 15;
		if (tc > 1 || th > 3) throw std::runtime_error{"JPEG: bad Huffman table id"};
		uint8_t counts[16]; int total = 0;
		for (int i = 0; i < 16; ++i) { counts[i] = u8(); total += counts[i]; }
		if (total > 256) throw std::runtime_error{"JPEG: bad Huffman table"};
		uint8_t syms[256];
		for (int i = 0; i < total; ++i) syms[i] = u8();
		(tc ? hac[th] : hdc[th]).build(counts, syms, total);
		len -= 17 + total;
	}
}

void Decoder::parse_sof(int marker) {
	if (have_frame) throw std::runtime_error{"JPEG: more than one frame (hierarchical files are not supported)"};
	u16();
	if (u8() != 8) throw std::runtime_error{"JPEG: only 8-bit samples are supported"};
	height = u16(); width = u16(); n_comp = u8();
	if (width <= 0 || height <= 0) throw std::runtime_error{"JPEG: empty image"};
	if (n_comp != 1 && n_comp != 3) throw std::runtime_error{"JPEG: " + std::to_string(n_comp) + "-component (CMYK / YCCK) files are not supported"};
	progressive = marker == 0xC2;
	for (int i = 0; i < n_comp; ++i) {
		comp[i].id = u8();
		const int hv = u8(); comp[i].h = hv >> 4; comp[i].v = hv & 15; comp[i].tq = u8() & 3;
		if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2) throw std::runtime_error{"JPEG: sampling factors above 2 are not supported"};
		hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v);
	}
	if (n_comp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }
	mcus_x = (width + 8 * hmax - 1) / (8 * hmax); mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
	for (int i = 0; i < n_comp; ++i) {
		Component& c = comp[i];
		c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
		c.plane_w = (width * c.h + hmax - 1) / hmax; c.plane_h = (height * c.v + vmax - 1) / vmax;
		c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
	}
	have_frame = true;
}

void Decoder::decode_block(Component& c, int16_t* blk, int ss, int se, int ah, int al) {
	if (!progressive) {
		const int t = dc_category(hdc[c.td]);
		c.pred += receive_extend(t);
		blk[0] = (int16_t)c.pred;
		for (int k = 1; k < 64;) {
			const int rs = decode_symbol(hac[c.ta]), r = rs >> 4, s = rs & 15;
			if (s == 0) { if (r != 15) break; k += 16; continue; }
			k += r;
			if (k > 63) throw std::runtime_error{"JPEG: coefficient index out of range"};
			blk[ZIGZAG[k]] = (int16_t)receive_extend(ac_size(s));
			++k;
		}
		return;
	}
	if (ss == 0) {   // DC scan of a progressive file
		if (ah == 0) { const int t = dc_category(hdc[c.td]); c.pred += receive_extend(t); blk[0] = (int16_t)(c.pred * (1 << al)); }
		else if (getbit()) blk[0] |= (int16_t)(1 << al);
		return;
	}
	const HuffTable& ht = hac[c.ta];
	if (ah == 0) {   // AC, first pass of the band
		if (eobrun > 0) { --eobrun; return; }
		for (int k = ss; k <= se;) {
			const int rs = decode_symbol(ht), r = rs >> 4, s = rs & 15;
			if (s == 0) {
				if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); break; }
				k += 16;
			} else {
				k += r;
				if (k > se) throw std::runtime_error{"JPEG: coefficient index out of range"};
				blk[ZIGZAG[k]] = (int16_t)(receive_extend(ac_size(s)) * (1 << al));
				++k;
			}
		}
		return;
	}
	// AC refinement: one more bit for every coefficient that is already non-zero, new +-1 coefficients in between
	const int plus = 1 << al, minus = -(1 << al);
	auto refine = [&](int16_t& v) { if (getbit() && (v & plus) == 0) v = (int16_t)(v + (v >= 0 ? plus : minus)); };
	int k = ss;
	if (eobrun == 0) {
		while (k <= se) {
			const int rs = decode_symbol(ht); int r = rs >> 4; const int s = rs & 15;
			int value = 0;
			if (s) value = getbit() ? plus : minus;
			else if (r != 15) { eobrun = 1 << r; if (r) eobrun += getbits(r); break; }
			while (k <= se) {       // step over r coefficients that are still zero, refining the non-zero ones on the way
				int16_t& v = blk[ZIGZAG[k]];
				if (v != 0) refine(v);
				else if (--r < 0) break;
				++k;
			}
			if (s && k <= se) blk[ZIGZAG[k]] = (int16_t)value;
			++k;
		}
	}
	if (eobrun > 0) {
		for (; k <= se; ++k) { int16_t& v = blk[ZIGZAG[k]]; if (v != 0) refine(v); }
		--eobrun;
	}
}

void Decoder::parse_sos() {
	if (!have_frame) throw std::runtime_error{"JPEG: scan before frame header"};
	u16();
	const int ns = u8();
	if (ns < 1 || ns > n_comp) throw std::runtime_error{"JPEG: bad scan header"};
	Component* sc[3];
	for (int i = 0; i < ns; ++i) {
		const int id = u8(), tt = u8();
		sc[i] = nullptr;
		for (int k = 0; k < n_comp; ++k) if (comp[k].id == id) sc[i] = &comp[k];
		if (!sc[i]) throw std::runtime_error{"JPEG: scan refers to an unknown component"};
		sc[i]->td = tt >> 4; sc[i]->ta = tt & 15;
		if (sc[i]->td > 3 || sc[i]->ta > 3) throw std::runtime_error{"JPEG: bad table selector"};
	}
	int ss = u8(), se = u8(); const int ahl = u8(); int ah = ahl >> 4, al = ahl & 15;
	if (!progressive) { ss = 0; se = 63; ah = al = 0; }
	else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) throw std::runtime_error{"JPEG: bad progressive scan parameters"};
	for (int i = 0; i < ns; ++i) {
		const bool need_dc = ss == 0 && ah == 0, need_ac = se > 0;
		if ((need_dc && !hdc[sc[i]->td].present) || (need_ac && !hac[sc[i]->ta].present)) throw std::runtime_error{"JPEG: scan uses an undefined Huffman table"};
	}
	reset_entropy();
	int until_restart = restart_interval;
	auto restart_if_due = [&]() {
		if (!restart_interval || --until_restart > 0) return;
		// RSTn: byte-align (drop the buffered bits), step over the marker, reset the predictors
		while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) {
			if (d[pos] == 0xFF && d[pos + 1] != 0 && d[pos + 1] != 0xFF) return;   // some other marker: the scan is over
			++pos;
		}
		if (pos + 1 < n) pos += 2;
		reset_entropy();
		until_restart = restart_interval;
	};
	if (ns == 1) {   // non-interleaved: the component's own blocks, row by row
		Component& c = *sc[0];
		const int bw = (c.plane_w + 7) / 8, bh = (c.plane_h + 7) / 8;
		for (int by = 0; by < bh; ++by) for (int bx = 0; bx < bw; ++bx) {
			decode_block(c, &c.coef[((size_t)by * c.blocks_w + bx) * 64], ss, se, ah, al);
			restart_if_due();
		}
	} else {
		for (int my = 0; my < mcus_y; ++my) for (int mx = 0; mx < mcus_x; ++mx) {
			for (int i = 0; i < ns; ++i) {
				Component& c = *sc[i];
				for (int v = 0; v < c.v; ++v) for (int h = 0; h < c.h; ++h)
					decode_block(c, &c.coef[((size_t)(my * c.v + v) * c.blocks_w + (mx * c.h + h)) * 64], ss, se, ah, al);
			}
			restart_if_due();
		}
	}
	// leave `pos` on the marker that ended the segment
	while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] != 0 && !(d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7))) ++pos;
}

void Decoder::finish(int& w, int& h, std::vector<uint8_t>& rgba) {
	// dequantise + inverse DCT: out = M F M^T with M[x][u] = c(u)/2 cos((2x+1) u pi / 16), rounded once, level shift 128
	struct Basis {
		float m[8][8];
		Basis() { for (int x = 0; x < 8; ++x) for (int u = 0; u < 8; ++u) m[x][u] = (float)((u == 0 ? std::sqrt(0.5) : 1.0) * 0.5 * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0)); }
	};
	static const Basis basis;   // initialised once, thread-safe (frames are decoded by several threads)
	const float (*M)[8] = basis.m;
	for (int ci = 0; ci < n_comp; ++ci) {
		Component& c = comp[ci];
		if (!qt_present[c.tq]) throw std::runtime_error{"JPEG: missing quantisation table"};
		const uint16_t* q = qt[c.tq];
		const int pw = c.blocks_w * 8;
		c.plane.resize((size_t)pw * c.blocks_h * 8);
		for (int by = 0; by < c.blocks_h; ++by) for (int bx = 0; bx < c.blocks_w; ++bx) {
			const int16_t* blk = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
			float F[64], T[64];
			bool ac = false;
			for (int i = 0; i < 64; ++i) { F[i] = (float)((int)blk[i] * (int)q[i]); if (i && blk[i]) ac = true; }
			uint8_t* out = &c.plane[(size_t)by * 8 * pw + bx * 8];
			if (!ac) {
				const int v = (int)std::lrintf(F[0] * 0.125f) + 128;
				const uint8_t b = (uint8_t)std::min(255, std::max(0, v));
				for (int y = 0; y < 8; ++y) memset(out + (size_t)y * pw, b, 8);
				continue;
			}
			for (int v = 0; v < 8; ++v) for (int x = 0; x < 8; ++x) {   // T[v][x] = sum_u F[v][u] M[x][u]   (rows of F are vertical frequencies v)
				float s = 0.f;
				for (int u = 0; u < 8; ++u) s += F[v * 8 + u] * M[x][u];
				T[v * 8 + x] = s;
			}
			for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
				float s = 0.f;
				for (int v = 0; v < 8; ++v) s += M[y][v] * T[v * 8 + x];
				const int px = (int)std::lrintf(s) + 128;
				out[(size_t)y * pw + x] = (uint8_t)std::min(255, std::max(0, px));
			}
		}
		c.coef.clear(); c.coef.shrink_to_fit();
	}
	w = width; h = height;
	rgba.resize((size_t)width * height * 4);
	// full-resolution rows of every component: the triangle filter (3 * nearer + farther) across each subsampled direction
	std::vector<std::vector<uint8_t>> rows((size_t)n_comp, std::vector<uint8_t>((size_t)width + 2));
	std::vector<int> vrow;
	for (int y = 0; y < height; ++y) {
		for (int ci = 0; ci < n_comp; ++ci) {
			const Component& c = comp[ci];
			const int pw = c.blocks_w * 8, hs = hmax / c.h, vs = vmax / c.v;
			uint8_t* dst = rows[(size_t)ci].data();
			if (hs == 1 && vs == 1) { memcpy(dst, &c.plane[(size_t)y * pw], (size_t)width); continue; }
			const int cw = c.plane_w, chh = c.plane_h;
			vrow.resize((size_t)cw);
			if (vs == 2) {   // vertical: 3 * nearer row + farther row (scaled by 4)
				const int sy = y >> 1, oy = std::min(chh - 1, std::max(0, (y & 1) ? sy + 1 : sy - 1));
				const uint8_t* a = &c.plane[(size_t)std::min(sy, chh - 1) * pw]; const uint8_t* b = &c.plane[(size_t)oy * pw];
				for (int x = 0; x < cw; ++x) vrow[(size_t)x] = 3 * a[x] + b[x];
			} else {
				const uint8_t* a = &c.plane[(size_t)y * pw];
				for (int x = 0; x < cw; ++x) vrow[(size_t)x] = 4 * a[x];
			}
			if (hs == 2) {
				for (int x = 0; x < width; ++x) {
					const int sx = std::min(x >> 1, cw - 1), ox = std::min(cw - 1, std::max(0, (x & 1) ? sx + 1 : sx - 1));
					dst[x] = (uint8_t)((3 * vrow[(size_t)sx] + vrow[(size_t)ox] + 8) >> 4);
				}
				// stb_image's row resampler for horizontally-only subsampled chroma (stbi__resample_row_h_2, what the reference decodes 4:2:2 files
				// with) weights the LAST pair the other way round: 3 * in[w-2] + in[w-1] at the even position.  Mirrored, so that a training pixel in
				// that column is the one the reference sees.
				if (vs == 1 && cw >= 2 && 2 * (cw - 1) < width) dst[2 * (cw - 1)] = (uint8_t)((3 * vrow[(size_t)cw - 2] + vrow[(size_t)cw - 1] + 8) >> 4);
			} else {
				for (int x = 0; x < width; ++x) dst[x] = (uint8_t)((vrow[(size_t)std::min(x, cw - 1)] + 2) >> 2);
			}
		}
		uint8_t* o = &rgba[(size_t)y * width * 4];
		if (n_comp == 1) {
			for (int x = 0; x < width; ++x) { const uint8_t g = rows[0][(size_t)x]; o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = g; o[4 * x + 3] = 255; }
		} else if (adobe_transform == 0) {
			for (int x = 0; x < width; ++x) { o[4 * x] = rows[0][(size_t)x]; o[4 * x + 1] = rows[1][(size_t)x]; o[4 * x + 2] = rows[2][(size_t)x]; o[4 * x + 3] = 255; }
		} else {
			for (int x = 0; x < width; ++x) {   // JFIF: full-range BT.601
				const float Y = rows[0][(size_t)x], cb = (float)rows[1][(size_t)x] - 128.f, cr = (float)rows[2][(size_t)x] - 128.f;
				const int r = (int)std::lrintf(Y + 1.402f * cr), g = (int)std::lrintf(Y - 0.344136f * cb - 0.714136f * cr), b = (int)std::lrintf(Y + 1.772f * cb);
				o[4 * x] = (uint8_t)std::min(255, std::max(0, r)); o[4 * x + 1] = (uint8_t)std::min(255, std::max(0, g)); o[4 * x + 2] = (uint8_t)std::min(255, std::max(0, b)); o[4 * x + 3] = 255;
			}
		}
	}
}

void Decoder::decode(int& w, int& h, std::vector<uint8_t>& rgba) {
	if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) throw std::runtime_error{"JPEG: missing SOI marker"};
	pos = 2;
	bool done = false;
	while (!done) {
		while (pos < n && d[pos] != 0xFF) ++pos;                 // resynchronise on the next marker
		while (pos < n && d[pos] == 0xFF) ++pos;                 // fill bytes
		if (pos >= n) break;
		const int marker = d[pos++];
		switch (marker) {
			case 0xD9: done = true; break;                                             // EOI
			case 0xDB: parse_dqt(); break;
			case 0xC4: parse_dht(); break;
			case 0xC0: case 0xC1: case 0xC2: parse_sof(marker); break;
			case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
				throw std::runtime_error{"JPEG: lossless / hierarchical coding is not supported"};
			case 0xC9: case 0xCA: case 0xCC: throw std::runtime_error{"JPEG: arithmetic coding is not supported"};
			case 0xDD: u16(); restart_interval = u16(); break;
			case 0xDA: parse_sos(); break;
			case 0xEE: {                                                               // Adobe: colour transform flag
				const int len = u16(); const size_t end = pos + (size_t)len - 2;
				if (len >= 14 && pos + 12 <= n && memcmp(d + pos, "Adobe", 5) == 0) adobe_transform = d[pos + 11];
				pos = std::min(end, n);
				break;
			}
			default:
				if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) break;         // TEM / stray RSTn: no payload
				{ const int len = u16(); pos = std::min(pos + (size_t)len - 2, n); }      // APPn, COM, ...: skipped
		}
	}
	if (!have_frame) throw std::runtime_error{"JPEG: no frame header"};
	finish(w, h, rgba);
}

} // namespace

void decode_jpeg_rgba8(const uint8_t* data, size_t n_bytes, int& w, int& h, std::vector<uint8_t>& pixels) {
	Decoder(data, n_bytes).decode(w, h, pixels);
}

} // namespace ngp
