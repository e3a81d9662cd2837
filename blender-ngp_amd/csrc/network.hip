// network.hip — the tiny-cuda-nn replacement for the NeRF path, written for gfx950 (wave64, MFMA 32x32x16 f16).
//
// Replaces (reference call sites): NerfNetwork::inference_mixed_precision_impl / density / forward_impl / backward_impl
// (include/neural-graphics-primitives/nerf_network.h:103-284) and the tcnn GridEncoding + FullyFusedMLP + SphericalHarmonics
// kernels they dispatch to; Trainer::optimizer_step (src/testbed_nerf.cu:2950).
//
// Design (MI355X-first, not a port of tcnn's 128-sample/threadblock warp-MMA tiling):
//   * ONE wave owns 32 samples end to end.  Activations are kept TRANSPOSED (features x samples) so that the MFMA result
//     layout D[row = feature][col = sample = lane&31] of one layer is, after an in-register f32->f16 pack, directly the
//     B operand (K x N) of the next layer: register r of lane-group g holds feature (r&3)+8(r>>2)+4g, and because a
//     matrix product is invariant under a permutation of K applied to both operands, the weights (A operand) are simply
//     stored pre-permuted.  No LDS round trip, no cross-lane traffic between layers.
//   * The two 32-lane halves of the wave split the 16 hash levels (8 each): their 16 fp16 features are exactly the two
//     K-blocks that half contributes to the first MFMA.  The hash gather therefore feeds the matrix core from registers.
//   * Weights (20 KiB fp16) are staged ONCE per workgroup into LDS already in A-operand order (1 KiB per 32x16 tile, one
//     conflict-free ds_read_b128 per MFMA), which keeps the kernel at ~100 VGPRs so several waves per SIMD can hide the
//     gather latency.
//   * Backward: the same structure run in reverse with transposed, pre-permuted weights; the final dL/dx tile lands as
//     8 levels per lane, so the hash-grid scatter (packed-f16 atomics) is issued from registers.  Activations / deltas are
//     streamed out as [feature][sample] planes for the weight-gradient kernel, whose contraction runs over samples.
//
// Roofline: the hash pass is HBM/L2-request bound (512 B gathered per sample, 4 B per request); the MLP is ~20 kFLOP per
// sample, i.e. a few % of the MFMA peak by construction (SURVEY §7 "Tiny-N MFMA").
#include <mutex>
#include <string.h>
#include "ngp_device.cuh"
#include "ngp_dev_knobs.h"

#pragma clang fp contract(fast)

namespace ngp {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NGP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
// The backward chain runs beside the next step's march (stream B).  Its waves raise their issue priority (s_setprio: the SIMD's instruction
// arbiter prefers the higher level) so that the march — default priority 0 — only takes the issue slots the chain leaves idle.
#ifndef NGP_CHAIN_PRIO
#define NGP_CHAIN_PRIO 3
#endif
#define NGP_RAISE_CHAIN_PRIORITY() __builtin_amdgcn_s_setprio(NGP_CHAIN_PRIO)

// parameter offsets (nerf_network.h:361-394: density MLP, rgb MLP, grid)
constexpr uint32_t W1_OFF = 0;                    // [64][32]
constexpr uint32_t W2_OFF = 64 * 32;              // [16][64]
constexpr uint32_t W3_OFF = W2_OFF + 16 * 64;     // [64][32]
constexpr uint32_t W4_OFF = W3_OFF + 64 * 32;     // [64][64]
constexpr uint32_t W5_OFF = W4_OFF + 64 * 64;     // [16][64]
constexpr uint32_t GRID_OFF = W5_OFF + 16 * 64;   // 10240

// K-slot -> feature maps.  slot = (kb, g, e): K-block kb, lane group g = lane>>5, element e of the 8-half operand.
enum { MAP_ENC = 0, MAP_HID = 1, MAP_RGBIN = 2, MAP_CH = 3 };
__device__ __forceinline__ int slot_feature(int map, int kb, int g, int e) {
	switch (map) {
		case MAP_ENC: return 16 * g + 8 * kb + e;                         // hash features: levels 8g..8g+7
		case MAP_HID: return 16 * kb + 8 * (e >> 2) + 4 * g + (e & 3);    // rows of a 32x32 D tile pair
		case MAP_RGBIN: return kb == 0 ? (8 * (e >> 2) + 4 * g + (e & 3)) : (16 + 8 * g + e); // [density out | SH]
		case 4: return 32 + 8 * g + e;                                    // MAP_XTRA (network_netx_mfma.cuh): the extra-dims block behind [density out | SH]
		default: return 8 * g + e;                                        // 16 output channels
	}
}

// ----------------------------------------------------------------------------------------------------------------
// LDS weight tiles.  A tile = 64 lanes x 8 halves in A-operand order for one (mt, kb).
// forward tile ids
constexpr int T_W1 = 0;   // + mt*2 + kb   (4)
constexpr int T_W2 = 4;   // + kb          (4)
constexpr int T_W3 = 8;   // + mt*2 + kb   (4)
constexpr int T_W4 = 12;  // + mt*4 + kb   (8)
constexpr int T_W5 = 20;  // + kb          (4)
constexpr int N_FWD_TILES = 24;
// backward (transposed) tile ids, placed after the forward ones
constexpr int T_W5T = 24; // + mt          (2)   A[i = h3 feat][slot = channel]
constexpr int T_W4T = 26; // + mt*4 + kb   (8)   A[i = h2 feat][slot = h3 feat]
constexpr int T_W3T = 34; // + kb          (4)   A[i = rgb-in feat][slot = h2 feat]
constexpr int T_W2T = 38; // + mt          (2)   A[i = h1 feat][slot = density-out feat]
constexpr int T_W1T = 40; // + kb          (4)   A[i = x feat][slot = h1 feat]
constexpr int N_ALL_TILES = 44;

// Staging, measured in round 4 (tools/mlp_fixed_cost_probe.py): gathering the tiles straight from global memory — 8 two-byte loads per lane and tile, every lane of a
// forward tile in a different 64-byte line — cost 17.7 us of the fused backward kernel's 59 (44 tiles) and ~10 us of every forward launch: the same vector-L1 look-up
// rate that bounds the hash gather (profiles/r04_forward_counters.md).  Now the raw row-major matrices come in ONCE with coalesced 16-byte loads into the LDS region
// the tiles will occupy, the lanes pick their tile elements out of LDS (branch-free, tile geometry in scalar registers) into registers, and after a barrier the tiles
// overwrite the raw copy.
// Raw half k of the parameter block sits at LDS half k + 2 * (k >> 5): one dword of padding per 64 bytes, so that the 32 rows a forward tile reads at one column
// fall into 32 different banks (row pitch 17 or 34 dwords instead of 16 or 32).
__device__ __forceinline__ int raw_slot(int k) { return k + 2 * (k >> 5); }

// one A-operand tile of a row-major [n_out][n_in] weight matrix at raw offset w_off, out of the LDS copy R: element (i = mt*32 + lane&31, K-slot (kb, lane>>5, e)).
// Tile geometry (everything but `lane`) is wave-uniform.
__device__ __forceinline__ h8 gather_tile_spec(const half_t* R, int w_off, int n_out, int n_in, int mt, int kb, int map, bool transposed, int lane) {
	const int i = mt * 32 + (lane & 31), g = lane >> 5;
	// slot_feature(map, kb, g, e) == base + cg * g + ch * (e >> 2) + (e & 3) for every map
	int base, cg, ch;
	switch (map) {
		case MAP_ENC: base = 8 * kb; cg = 16; ch = 4; break;
		case MAP_HID: base = 16 * kb; cg = 4; ch = 8; break;
		case MAP_RGBIN: base = kb == 0 ? 0 : 16; cg = kb == 0 ? 4 : 8; ch = kb == 0 ? 8 : 4; break;
		case 4: base = 32; cg = 8; ch = 4; break;
		default: base = 0; cg = 8; ch = 4; break;
	}
	const int fg = base + cg * g;
	h8 r;
	if (!transposed) {
		// A[i = out][slot = in feature f]: the four features of a half (e & 3) are neighbours in the row — two 4-byte LDS reads each (base, cg, ch, n_in and the matrix
		// offsets are multiples of 4, and a raw slot never splits an even pair)
		const bool i_ok = i < n_out;
		const int k0 = w_off + i * n_in;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int f0 = fg + ch * h;
			const bool ok = i_ok && f0 < n_in;
			const uint32_t* q = (const uint32_t*)(R + raw_slot(ok ? k0 + f0 : 0));
			const uint32_t lo = ok ? q[0] : 0u, hi = ok ? q[1] : 0u;
			const h2 a = __builtin_bit_cast(h2, lo), b = __builtin_bit_cast(h2, hi);
			r[4 * h] = a[0]; r[4 * h + 1] = a[1]; r[4 * h + 2] = b[0]; r[4 * h + 3] = b[1];
		}
		return r;
	}
	// A[i = in][slot = out feature f]: a column of the row-major matrix, one half per row
	const bool i_ok = i < n_in;
	const int k0 = w_off + i;
#pragma unroll
	for (int e = 0; e < 8; ++e) {
		const int f = fg + ch * (e >> 2) + (e & 3);
		const bool ok = i_ok && f < n_out;
		const half_t v = R[raw_slot(ok ? k0 + f * n_in : 0)];
		r[e] = ok ? v : (half_t)0.0f;
	}
	return r;
}

// tile of the base network's parameter block
__device__ __forceinline__ h8 gather_tile(const half_t* R, int tile, int lane) {
	int w_off, n_out, n_in, mt, kb, map; bool transposed = false;
	if (tile < T_W2)       { w_off = W1_OFF; n_out = 64; n_in = 32; mt = (tile - T_W1) >> 1; kb = (tile - T_W1) & 1; map = MAP_ENC; }
	else if (tile < T_W3)  { w_off = W2_OFF; n_out = 16; n_in = 64; mt = 0; kb = tile - T_W2; map = MAP_HID; }
	else if (tile < T_W4)  { w_off = W3_OFF; n_out = 64; n_in = 32; mt = (tile - T_W3) >> 1; kb = (tile - T_W3) & 1; map = MAP_RGBIN; }
	else if (tile < T_W5)  { w_off = W4_OFF; n_out = 64; n_in = 64; mt = (tile - T_W4) >> 2; kb = (tile - T_W4) & 3; map = MAP_HID; }
	else if (tile < T_W5T) { w_off = W5_OFF; n_out = 16; n_in = 64; mt = 0; kb = tile - T_W5; map = MAP_HID; }
	else if (tile < T_W4T) { w_off = W5_OFF; n_out = 16; n_in = 64; mt = tile - T_W5T; kb = 0; map = MAP_CH; transposed = true; }
	else if (tile < T_W3T) { w_off = W4_OFF; n_out = 64; n_in = 64; mt = (tile - T_W4T) >> 2; kb = (tile - T_W4T) & 3; map = MAP_HID; transposed = true; }
	else if (tile < T_W2T) { w_off = W3_OFF; n_out = 64; n_in = 32; mt = 0; kb = tile - T_W3T; map = MAP_HID; transposed = true; }
	else if (tile < T_W1T) { w_off = W2_OFF; n_out = 16; n_in = 64; mt = tile - T_W2T; kb = 0; map = MAP_RGBIN; transposed = true; }
	else                   { w_off = W1_OFF; n_out = 64; n_in = 32; mt = 0; kb = tile - T_W1T; map = MAP_HID; transposed = true; }
	return gather_tile_spec(R, w_off, n_out, n_in, mt, kb, map, transposed, lane);
}

// The first N_RAW halves of `params` (the network's matrices) -> N_TILES A-operand tiles in lds_tiles, by the 4 waves of a 256-thread workgroup.
// tile_of(R, tile, lane) names the network: gather_tile, gm_gather_tile, nx_gather_tile<NH, XK>.
template <int N_RAW, int N_TILES, class TileOf>
__device__ __forceinline__ void stage_tiles(h8* lds_tiles, const half_t* __restrict__ params, TileOf tile_of) {
	static_assert(N_RAW % 8 == 0 && (N_RAW + 2 * (N_RAW / 32)) * 2 <= N_TILES * 1024, "the raw copy fits the region its tiles take");
	half_t* R = (half_t*)lds_tiles;
	if (((uintptr_t)params & 15u) == 0) {
		for (int c = threadIdx.x; c < N_RAW / 8; c += 256) {
			const uint4 v = ((const uint4*)params)[c];
			uint32_t* d = (uint32_t*)(R + raw_slot(8 * c));   // 8 | 32: a chunk never straddles a padding dword; 4-byte aligned
			d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
		}
	} else {
		for (int k = threadIdx.x; k < N_RAW; k += 256) R[raw_slot(k)] = params[k];
	}
	__syncthreads();
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	constexpr int PER_WAVE = (N_TILES + 3) / 4;
	h8 regs[PER_WAVE];
#pragma unroll
	for (int q = 0; q < PER_WAVE; ++q)
		if (w + 4 * q < N_TILES) regs[q] = tile_of((const half_t*)R, w + 4 * q, lane);
	__syncthreads();
#pragma unroll
	for (int q = 0; q < PER_WAVE; ++q)
		if (w + 4 * q < N_TILES) lds_tiles[(w + 4 * q) * 64 + lane] = regs[q];
	__syncthreads();
}

// f32 D tile (32 rows) -> two f16 K-blocks (B operand of the next layer), optional ReLU, optional positive-mask output
// Converted two at a time (v_cvt_pk_f16_f32, round to nearest even like the scalar conversion) and clamped AFTER the conversion with one packed
// SIGNED-INTEGER max on the fp16 bit patterns: rounding is monotone and maps 0 to 0, so this is fp16(max(v, 0)) bit for bit in half the instructions,
// and everything with the sign bit set (-0 included) becomes +0 — mask_delta below relies on "positive" == "bits != 0".
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2v __attribute__((ext_vector_type(2)));
typedef int16_t i16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 relu_pk(h2 v) {
	const i16x2v nil = {0, 0};
	return __builtin_bit_cast(h2, __builtin_elementwise_max(__builtin_bit_cast(i16x2v, v), nil));
}
__device__ __forceinline__ h2 cvt_pk_f16(float a, float b) {
	f32x2v v; v[0] = a; v[1] = b;
	return __builtin_convertvector(v, h2);
}
template <bool RELU>
__device__ __forceinline__ void d_to_b(const f32x16& d, h8& b0, h8& b1) {
#pragma unroll
	for (int e = 0; e < 8; e += 2) {
		h2 p0 = cvt_pk_f16(d[e], d[e + 1]), p1 = cvt_pk_f16(d[8 + e], d[9 + e]);
		if (RELU) { p0 = relu_pk(p0); p1 = relu_pk(p1); }
		b0[e] = p0[0]; b0[e + 1] = p0[1]; b1[e] = p1[0]; b1[e + 1] = p1[1];
	}
}

// ----------------------------------------------------------------------------------------------------------------
// hash-grid level (tcnn kernel_grid, 3-D, F = 2, linear interpolation, Hash grid type)
__device__ __forceinline__ uint32_t grid_index(const NgpGridLevel& lv, uint32_t x, uint32_t y, uint32_t z) {
	uint32_t stride = 1, index = 0;
	if (stride <= lv.size) { index += x * stride; stride *= lv.resolution; }
	if (stride <= lv.size) { index += y * stride; stride *= lv.resolution;
		if (stride <= lv.size) { index += z * stride; stride *= lv.resolution; } }
	if (lv.size < stride) index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
	if ((lv.size & (lv.size - 1)) == 0) return index & (lv.size - 1);
	return index >= lv.size ? index % lv.size : index;
}

// tcnn grid_index<N_DIMS>: dense strides while they fit the table, else the spatial hash of the first N_DIMS primes {1, 2654435761, 805459861}
template <int D>
__device__ __forceinline__ uint32_t grid_index_nd(const NgpGridLevel& lv, uint32_t x, uint32_t y, uint32_t z) {
	if (D == 3) return grid_index(lv, x, y, z);
	uint32_t stride = 1, index = 0;
	if (stride <= lv.size) { index += x * stride; stride *= lv.resolution;
		if (stride <= lv.size) { index += y * stride; stride *= lv.resolution; } }
	if (lv.size < stride) index = (x * 1u) ^ (y * 2654435761u);
	if ((lv.size & (lv.size - 1)) == 0) return index & (lv.size - 1);
	return index >= lv.size ? index % lv.size : index;
}

// position of one sample as ONE memory request (global_load_dwordx3; the record is only 4-byte aligned) instead of three
typedef float f3_t __attribute__((ext_vector_type(3)));
__device__ __forceinline__ f3_t load_pos3(const float* __restrict__ c) { f3_t v; __builtin_memcpy(&v, c, 12); return v; }

struct LevelPos { uint32_t gx, gy, gz; float fx, fy, fz; };
__device__ __forceinline__ LevelPos level_pos(const NgpGridLevel& lv, float px, float py, float pz) {
	LevelPos p;
	// [tcnn] pos_fract: fmaf(scale, input, 0.5f) — one rounding, spelled out so that it does not depend on the contraction mode
	float x = __builtin_fmaf(lv.scale, px, 0.5f), y = __builtin_fmaf(lv.scale, py, 0.5f), z = __builtin_fmaf(lv.scale, pz, 0.5f);
	float flx = floorf(x), fly = floorf(y), flz = floorf(z);
	p.gx = (uint32_t)(int)flx; p.gy = (uint32_t)(int)fly; p.gz = (uint32_t)(int)flz;
	p.fx = x - flx; p.fy = y - fly; p.fz = z - flz;
	return p;
}

// Trilinear blend of the eight corner entries, spelled out: products in x, y, z order, one fused multiply-add per corner and feature.  No implicit contraction in
// here — under `fp contract(fast)` the compiler chose per INSTANTIATION what to fuse, and the encoder kernel that can read de-hashed level copies rounded 23 of 20 000
// samples' features differently from the one that cannot (round 6).  Every gather flavour (4-byte, x-pair, de-hashed) ends in this one function.
__device__ __forceinline__ void blend_corners(const LevelPos& p, const h2 (&v)[8], half_t& o0, half_t& o1) {
#pragma clang fp contract(off)
	float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		// weight = prod_d (bit ? frac : 1 - frac), multiplied in x, y, z order
		float w = (c & 1) ? p.fx : (1.0f - p.fx);
		w = w * (((c >> 1) & 1) ? p.fy : (1.0f - p.fy));
		w = w * (((c >> 2) & 1) ? p.fz : (1.0f - p.fz));
		r0 = __builtin_fmaf(w, (float)v[c][0], r0);
		r1 = __builtin_fmaf(w, (float)v[c][1], r1);
	}
	o0 = (half_t)r0; o1 = (half_t)r1;
}

// PAIR: the two corners that differ in x only sit in one aligned 8-byte pair whenever their indices differ in bit 0 alone (always for
// a hashed level at even x, because the x term of the hash is x itself; for a dense level at even index).  One 8-byte load then
// replaces two 4-byte loads of the same cache line, the second of which would otherwise queue behind the pending miss of the first.
template <bool PAIR = false>
__device__ __forceinline__ void encode_level(const NgpGridLevel lv, const h2* __restrict__ grid, float px, float py, float pz, half_t& o0, half_t& o1) {
	const LevelPos p = level_pos(lv, px, py, pz);
	h2 v[8];
	if (!PAIR) {
#pragma unroll
		for (int c = 0; c < 8; ++c) {
			const uint32_t idx = grid_index(lv, p.gx + (c & 1), p.gy + ((c >> 1) & 1), p.gz + ((c >> 2) & 1));
			// uniform base + 32-bit per-lane byte offset (the whole table is < 4 GiB): saddr-form global_load_dword
			v[c] = *(const h2*)((const char*)grid + (size_t)((lv.offset + idx) * 4u));
		}
	} else {
#pragma unroll
		for (int c = 0; c < 8; c += 2) {
			const uint32_t i0 = grid_index(lv, p.gx, p.gy + ((c >> 1) & 1), p.gz + ((c >> 2) & 1));
			const uint32_t i1 = grid_index(lv, p.gx + 1, p.gy + ((c >> 1) & 1), p.gz + ((c >> 2) & 1));
			if ((i0 ^ i1) == 1u) {   // lv.offset is a multiple of 8 entries: the pair is 8-byte aligned
				const uint2 pr = *(const uint2*)((const char*)grid + (size_t)((lv.offset + (i0 & ~1u)) * 4u));
				const uint32_t w0 = (i0 & 1u) ? pr.y : pr.x, w1 = (i0 & 1u) ? pr.x : pr.y;
				v[c] = __builtin_bit_cast(h2, w0); v[c + 1] = __builtin_bit_cast(h2, w1);
			} else {
				v[c] = *(const h2*)((const char*)grid + (size_t)((lv.offset + i0) * 4u));
				v[c + 1] = *(const h2*)((const char*)grid + (size_t)((lv.offset + i1) * 4u));
			}
		}
	}
	blend_corners(p, v, o0, o1);
}

// 2-D (image fitting, 4 corners) or 3-D level; PAIR (3-D only): see encode_level
template <int D, bool PAIR = false>
__device__ __forceinline__ void encode_level_nd(const NgpGridLevel lv, const h2* __restrict__ grid, float px, float py, float pz, half_t& o0, half_t& o1) {
	if (D == 3) { encode_level<PAIR>(lv, grid, px, py, pz, o0, o1); return; }
	const LevelPos p = level_pos(lv, px, py, 0.0f);
	h2 v[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const uint32_t idx = grid_index_nd<2>(lv, p.gx + (c & 1), p.gy + ((c >> 1) & 1), 0u);
		v[c] = *(const h2*)((const char*)grid + (size_t)((lv.offset + idx) * 4u));
	}
	float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		float w = (c & 1) ? p.fx : (1.0f - p.fx);
		w *= ((c >> 1) & 1) ? p.fy : (1.0f - p.fy);
		r0 += w * (float)v[c][0];
		r1 += w * (float)v[c][1];
	}
	o0 = (half_t)r0; o1 = (half_t)r1;
}

// (measured: 8-byte pair loads cost the fused kernels 15-20 % — more instructions at 2 waves/SIMD — and gain the stand-alone encode 16 %)
#define NGP_FUSED_PAIR false
// lane (j, g) encodes levels 8g..8g+7 of its sample: x0 = levels 8g..8g+3, x1 = levels 8g+4..8g+7  (MAP_ENC)
__device__ __forceinline__ void encode_half(const NgpNetDesc* __restrict__ desc, const h2* __restrict__ grid, int g, float px, float py, float pz, h8& x0, h8& x1) {
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		half_t a, b;
		encode_level<NGP_FUSED_PAIR>(desc->levels[8 * g + m], grid, px, py, pz, a, b);
		x0[2 * m] = a; x0[2 * m + 1] = b;
	}
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		half_t a, b;
		encode_level<NGP_FUSED_PAIR>(desc->levels[8 * g + 4 + m], grid, px, py, pz, a, b);
		x1[2 * m] = a; x1[2 * m + 1] = b;
	}
}

// SphericalHarmonics degree 4 of 2d-1; lane group g takes coefficients 8g..8g+7
__device__ __forceinline__ h8 sh4_half(int g, float dx, float dy, float dz) {
	float x = dx * 2.0f - 1.0f, y = dy * 2.0f - 1.0f, z = dz * 2.0f - 1.0f;
	float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	h8 r;
	if (g == 0) {
		r[0] = (half_t)0.28209479177387814f;
		r[1] = (half_t)(-0.48860251190291987f * y);
		r[2] = (half_t)(0.48860251190291987f * z);
		r[3] = (half_t)(-0.48860251190291987f * x);
		r[4] = (half_t)(1.0925484305920792f * xy);
		r[5] = (half_t)(-1.0925484305920792f * yz);
		r[6] = (half_t)(0.94617469575755997f * z2 - 0.31539156525251999f);
		r[7] = (half_t)(-1.0925484305920792f * xz);
	} else {
		r[0] = (half_t)(0.54627421529603959f * x2 - 0.54627421529603959f * y2);
		r[1] = (half_t)(0.59004358992664352f * y * (-3.0f * x2 + y2));
		r[2] = (half_t)(2.8906114426405538f * xy * z);
		r[3] = (half_t)(0.45704579946446572f * y * (1.0f - 5.0f * z2));
		r[4] = (half_t)(0.3731763325901154f * z * (5.0f * z2 - 3.0f));
		r[5] = (half_t)(0.45704579946446572f * x * (1.0f - 5.0f * z2));
		r[6] = (half_t)(1.4453057213202769f * z * (x2 - y2));
		r[7] = (half_t)(0.59004358992664352f * x * (-x2 + 3.0f * y2));
	}
	return r;
}

// ----------------------------------------------------------------------------------------------------------------
// MLP forward for one 32-sample tile held by a wave.  Returns density-net D tile and (optionally) the rgb output tile;
// hidden activations are returned in B-operand form for the backward recompute.
struct FwdActs { h8 h1[4]; h8 rin[2]; h8 h2[4]; h8 h3[4]; };

template <bool DENSITY_ONLY, bool KEEP>
__device__ __forceinline__ void mlp_forward(const h8* __restrict__ lt, int lane, const h8& x0, const h8& x1, const h8& sh, f32x16& dd, f32x16& oo, FwdActs* acts) {
	const f32x16 zero = {};
	f32x16 a0 = NGP_MFMA(lt[(T_W1 + 0) * 64 + lane], x0, zero);
	a0 = NGP_MFMA(lt[(T_W1 + 1) * 64 + lane], x1, a0);
	f32x16 a1 = NGP_MFMA(lt[(T_W1 + 2) * 64 + lane], x0, zero);
	a1 = NGP_MFMA(lt[(T_W1 + 3) * 64 + lane], x1, a1);
	h8 h[4];
	d_to_b<true>(a0, h[0], h[1]);
	d_to_b<true>(a1, h[2], h[3]);
	if (KEEP) { for (int k = 0; k < 4; ++k) acts->h1[k] = h[k]; }
	dd = zero;
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) dd = NGP_MFMA(lt[(T_W2 + kb) * 64 + lane], h[kb], dd);
	if (DENSITY_ONLY) return;
	h8 r0, junk;
	d_to_b<false>(dd, r0, junk);
	if (KEEP) { acts->rin[0] = r0; acts->rin[1] = sh; }
	a0 = NGP_MFMA(lt[(T_W3 + 0) * 64 + lane], r0, zero);
	a0 = NGP_MFMA(lt[(T_W3 + 1) * 64 + lane], sh, a0);
	a1 = NGP_MFMA(lt[(T_W3 + 2) * 64 + lane], r0, zero);
	a1 = NGP_MFMA(lt[(T_W3 + 3) * 64 + lane], sh, a1);
	d_to_b<true>(a0, h[0], h[1]);
	d_to_b<true>(a1, h[2], h[3]);
	if (KEEP) { for (int k = 0; k < 4; ++k) acts->h2[k] = h[k]; }
	a0 = zero; a1 = zero;
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) {
		a0 = NGP_MFMA(lt[(T_W4 + kb) * 64 + lane], h[kb], a0);
		a1 = NGP_MFMA(lt[(T_W4 + 4 + kb) * 64 + lane], h[kb], a1);
	}
	d_to_b<true>(a0, h[0], h[1]);
	d_to_b<true>(a1, h[2], h[3]);
	if (KEEP) { for (int k = 0; k < 4; ++k) acts->h3[k] = h[k]; return; }
	oo = zero;
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) oo = NGP_MFMA(lt[(T_W5 + kb) * 64 + lane], h[kb], oo);
}

// ----------------------------------------------------------------------------------------------------------------
// Fused forward kernel: hash encode -> density MLP -> SH -> rgb MLP.  MODE 0 inference (rgb sigma), 1 density only,
// 2 training forward (also stores the encoded features for backward).
// PRE = 0: fully fused (gathers inside).  PRE = 1: the features were produced by encode_planes_kernel into
// x_planes[level][n_pad] (half2 per sample); the kernel is then the MLP alone and runs at twice the occupancy.
template <int MODE, int PRE>
__global__ void __launch_bounds__(256, PRE ? 4 : 2) nerf_forward_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params,
                                                           const float* __restrict__ coords, uint32_t coord_stride, uint32_t n,
                                                           half_t* __restrict__ out, uint32_t out_stride, half_t* __restrict__ x_saved,
                                                           const h2* __restrict__ x_planes, uint32_t n_pad) {
	__shared__ __attribute__((aligned(16))) h8 lds_tiles[N_FWD_TILES * 64];
	stage_tiles<(MODE == 1 ? W3_OFF : GRID_OFF), (MODE == 1 ? T_W3 : N_FWD_TILES)>(lds_tiles, params, gather_tile);

	const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
	const uint32_t n_tiles = (n + 31) / 32;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
	const h2* __restrict__ grid = (const h2*)(params + GRID_OFF);

	for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
		const uint32_t s = tile * 32 + j;
		const bool valid = s < n;
		const float* c = coords + (size_t)(valid ? s : 0) * coord_stride;
		h8 x0, x1;
		if (PRE == 1) {
			const h2* xp = x_planes + (size_t)(8 * g) * n_pad + (valid ? s : 0);
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const h2 a = xp[(size_t)m * n_pad], b = xp[(size_t)(4 + m) * n_pad];
				x0[2 * m] = a[0]; x0[2 * m + 1] = a[1];
				x1[2 * m] = b[0]; x1[2 * m + 1] = b[1];
			}
		} else {
			encode_half(desc, grid, g, c[0], c[1], c[2], x0, x1);
		}
		if (MODE == 2 && valid) {
			h8* dst = (h8*)(x_saved + (size_t)s * 32 + 16 * g);
			dst[0] = x0; dst[1] = x1;
		}
		h8 sh = {};
		if (MODE != 1) sh = sh4_half(g, c[4], c[5], c[6]);
		f32x16 dd, oo;
		uint32_t lt_off = 0;
		asm volatile("" : "+s"(lt_off)); // keep the LDS weight reads inside the loop (no LICM into 96 VGPRs)
		mlp_forward<MODE == 1, false>(lds_tiles + lt_off, lane, x0, x1, sh, dd, oo, nullptr);
		if (valid && g == 0) {
			if (MODE == 1) {
				out[s] = (half_t)dd[0];
			} else {
				typedef _Float16 h4 __attribute__((ext_vector_type(4)));
				h4 o; o[0] = (half_t)oo[0]; o[1] = (half_t)oo[1]; o[2] = (half_t)oo[2]; o[3] = (half_t)dd[0];
				*(h4*)(out + (size_t)s * out_stride) = o;
			}
		}
	}
}

// ----------------------------------------------------------------------------------------------------------------
// XCD-affine hash encode.  The 11 hashed levels are 2 MiB each (22 MiB together), the L2 of one XCD is 4 MiB: in the fused
// kernel every XCD touches every level and 60 % of the L2 requests miss (rocprof: TCC_MISS / TCC_REQ), i.e. go over the
// fabric, whose random-64-B-request rate (~75 G/s chip-wide, tools/gather_probe.hip) is 3.6x below the L2-hit rate (~270 G/s).
// Here work items are (level, 1024-sample chunk); the level-major item list is cut into 8 cost-balanced queues (hashed level =
// 2 units, dense = 1), one per XCD, and a persistent workgroup pulls from the queue of the XCD it RUNS on (HW_REG_XCC_ID), so
// that one XCD walks at most two tables in sequence and its L2 holds them.  Empty queue => steal from the next one.  The
// placement only affects speed; any block may process any item.
#ifndef NGP_ENC_CHUNK
#define NGP_ENC_CHUNK 1024
#endif
constexpr uint32_t ENC_CHUNK = NGP_ENC_CHUNK;               // samples of one level per work item (build knobs: tools/gpu_r05_enc.sh sweeps them on the fox step's own samples)
constexpr uint32_t ENC_QUEUE_STRIDE = 64;   // uint32 words between the 8 queue counters
constexpr uint32_t ENC_QUEUE_BYTES = 8 * ENC_QUEUE_STRIDE * 4;

__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }

template <int D>   // 3: NeRF / SDF positions; 2: image fitting (ngp_hip_gridmlp_forward_ws)
__global__ void __launch_bounds__(256) encode_planes_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ coords,
                                                            uint32_t coord_stride, uint32_t n, uint32_t n_pad, h2* __restrict__ planes, uint32_t* __restrict__ queues, uint32_t cost_model, uint32_t grid_off, uint32_t items_per_claim) {
	__shared__ uint32_t s_first[9];
	__shared__ uint32_t s_item;
	const uint32_t n_chunks = (n + ENC_CHUNK - 1) / ENC_CHUNK;
	if (threadIdx.x < 9) {
		// queue k = items [first[k], first[k+1]) of the level-major list, cut where the running cost passes k/8 of the total
		uint32_t cost[16], total = 0;
		for (int l = 0; l < 16; ++l) {
			const NgpGridLevel lv = desc->levels[l];
			const uint64_t dense = (uint64_t)lv.resolution * lv.resolution * (D == 3 ? lv.resolution : 1u);
			// relative cost of one (level, chunk) item = its L2 requests (the vector L1 -> L2 request path, ~0.5 per clock and CU, is what bounds a gather that hits
			// the L2): measured per level on ray-coherent samples (one launch per level, tools/fwd_path_trace.sh): dense levels 4.2 us / 2^19 samples each, hashed
			// levels 9 us at resolution 81 rising to 15 us from resolution ~600 on, where every sample touches its four (y, z) rows' lines alone
			if (cost_model == 0) cost[l] = dense > lv.size ? 2u : 1u;
			else cost[l] = dense > lv.size ? 56u + (lv.resolution < 600u ? lv.resolution * 64u / 600u : 64u) : 34u;
			total += cost[l] * n_chunks;
		}
		const uint32_t target = (uint32_t)((uint64_t)total * threadIdx.x / 8u);
		uint32_t first = 16u * n_chunks, run = 0;
		for (int l = 0; l < 16; ++l) {
			const uint32_t span = cost[l] * n_chunks;
			if (target < run + span) { first = l * n_chunks + (target - run + cost[l] - 1) / cost[l]; break; }
			run += span;
		}
		s_first[threadIdx.x] = first;
	}
	__syncthreads();
	const h2* __restrict__ grid = (const h2*)(params + grid_off);   // GRID_OFF for the base family
	const uint32_t home = xcc_id();
	for (uint32_t q = 0; q < 8; ++q) {
		const uint32_t k = (home + q) & 7u;
		const uint32_t begin = s_first[k], end = s_first[k + 1];
		uint32_t* counter = queues + k * ENC_QUEUE_STRIDE;  // one counter per 256-B line: same-address atomics serialise at the memory side
		const uint32_t n_claims = (end - begin + items_per_claim - 1) / items_per_claim;
		for (;;) {
			if (threadIdx.x == 0) {
				// stealing (q > 0) looks before it claims, so that the 2048 x 7 visits of drained queues stay plain loads
				uint32_t claim = n_claims;
				if (q == 0 || __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_claims) claim = atomicAdd(counter, 1u);
				s_item = claim;
			}
			__syncthreads();
			const uint32_t claim = __builtin_amdgcn_readfirstlane(s_item);
			__syncthreads();
			if (claim >= n_claims) break;
			const uint32_t item0 = begin + claim * items_per_claim;
			const uint32_t item1 = item0 + items_per_claim < end ? item0 + items_per_claim : end;
			for (uint32_t item = item0; item < item1; ++item) {
				const uint32_t level = item / n_chunks, chunk = item - level * n_chunks;
				const NgpGridLevel lv = desc->levels[level];
				h2* __restrict__ dst = planes + (size_t)level * n_pad;
				constexpr int PER_THREAD = ENC_CHUNK / 256;
				half_t a[PER_THREAD], b[PER_THREAD];
#pragma unroll
				for (int u = 0; u < PER_THREAD; ++u) {
					const uint32_t smp = chunk * ENC_CHUNK + u * 256 + threadIdx.x;
					const float* c = coords + (size_t)(smp < n ? smp : 0) * coord_stride;
					encode_level_nd<D, true>(lv, grid, c[0], c[1], D == 3 ? c[2] : 0.0f, a[u], b[u]);
				}
#pragma unroll
				for (int u = 0; u < PER_THREAD; ++u) {
					const uint32_t smp = chunk * ENC_CHUNK + u * 256 + threadIdx.x;
					h2 v; v[0] = a[u]; v[1] = b[u];
					if (smp < n) dst[smp] = v;
				}
			}
		}
	}
}

// Encode of the levels [l0, l1) for all samples, one thread per sample, into the level planes: one LAUNCH per group of levels whose tables fit an XCD's L2
// together.  Between two launches nothing else touches the tables, so each XCD fetches a level's 2 MiB once and serves every further touch from its L2.
template <bool PAIR>
__global__ void __launch_bounds__(256) encode_levels_planes_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ coords,
                                                                   uint32_t coord_stride, uint32_t n, uint32_t n_pad, h2* __restrict__ planes, uint32_t l0, uint32_t l1, uint32_t grid_off) {
	const h2* __restrict__ grid = (const h2*)(params + grid_off);
	for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < n; s += gridDim.x * 256u) {
		const float* c = coords + (size_t)s * coord_stride;
		// the positions stream through once per launch: keep them from displacing table lines in the L2
		const float px = __builtin_nontemporal_load(c), py = __builtin_nontemporal_load(c + 1), pz = __builtin_nontemporal_load(c + 2);
#pragma unroll 1
		for (uint32_t l = l0; l < l1; ++l) {
			half_t a, b;
			encode_level<PAIR>(desc->levels[l], grid, px, py, pz, a, b);
			h2 v; v[0] = a; v[1] = b;
			__builtin_nontemporal_store(__builtin_bit_cast(uint32_t, v), (uint32_t*)(planes + (size_t)l * n_pad + s));
		}
	}
}

// ----------------------------------------------------------------------------------------------------------------
// dL/d(pre-activation) of a ReLU layer as fp16: the fp32 gradient where the forward activation was positive, 0 elsewhere.  Activations are
// post-ReLU (>= +0), so "positive" is "bit pattern != 0": per PAIR of elements one packed conversion, min(bits, 1), 0 - that (0x0000 / 0xffff) and an AND.
__device__ __forceinline__ h8 mask_delta(const f32x16& t, int half_idx, const h8& fwd_act) {
	h8 r;
	const u16x2v one = {1, 1}, nil = {0, 0};
#pragma unroll
	for (int e = 0; e < 8; e += 2) {
		h2 a; a[0] = fwd_act[e]; a[1] = fwd_act[e + 1];
		const u16x2v keep = nil - __builtin_elementwise_min(__builtin_bit_cast(u16x2v, a), one);
		const u16x2v bits = __builtin_bit_cast(u16x2v, cvt_pk_f16(t[8 * half_idx + e], t[8 * half_idx + e + 1])) & keep;
		const h2 v = __builtin_bit_cast(h2, bits);
		r[e] = v[0]; r[e + 1] = v[1];
	}
	return r;
}

// The gradient terms must round like tcnn's (T)(weight * grad): fp32 product, THEN fp16; no contraction anywhere in this section.
#pragma clang fp contract(off)

// ----------------------------------------------------------------------------------------------------------------
// Hash-grid backward WITHOUT global atomics ("owner computes").
// Measured on MI355X (tools/atomic_probe.hip): scattered global atomics cap at ~21 Gop/s chip-wide whatever the footprint or the
// XCD locality, LDS atomics reach ~190 Gop/s and cost no HBM traffic.  tcnn's formulation (one atomicAdd(half2) per sample x level x
// corner, 33.5 M per step) therefore costs ~2.4 ms here.  Instead every workgroup OWNS a slice of one level's table (<= 32768
// entries = 128 KiB of LDS as half2), scans the samples of its chunk, recomputes the 8 corner indices of its level and
// accumulates only the corners that fall into its slice with ds_pk_add_f16.  Small (dense) levels, which are heavily contended,
// are additionally split over sample chunks (K private copies) and summed by grid_combine_kernel.  No global atomic, no memset of
// the gradient table (every entry is written exactly once by the combine pass).
constexpr uint32_t GB_SLICE = 16384;       // half2 entries per LDS slice of the float fallback (64 KiB: two workgroups per CU)
constexpr uint32_t GB_ITEMS = 64;          // its work items per level (slices x sample chunks)

struct GbSplit { uint32_t n_slices, k_chunks; };
__host__ __device__ __forceinline__ GbSplit gb_split(uint32_t level_size) {
	GbSplit s;
	s.n_slices = (level_size + GB_SLICE - 1) / GB_SLICE;
	s.k_chunks = GB_ITEMS / s.n_slices;
	if (s.k_chunks < 1) s.k_chunks = 1;
	return s;
}

// Hashed levels: BIN, then accumulate in FIXED POINT.
//  * tools/lds_atomic_probe.hip on MI355X: LDS float atomics (ds_pk_add_f16, ds_add_f32) run at ~200 G/s chip-wide whatever the bank pattern,
//    integer ones at ~3100 G/s (u32) / ~2400 G/s (u64) on random addresses.  Every fp16 term is an integer multiple of 2^-24 below 2^16, so
//    term * 2^24 is an exact 41-bit integer and a 64-bit accumulator per feature holds the EXACT sum of up to 2^22 terms: no scale to choose,
//    no overflow, no dependence on the order of the adds; the sum is rounded to fp16 once (tcnn rounds after every atomicAdd(half2)).
//  * "owner computes" makes every slice owner scan every sample; with 16-byte entries a slice is 8192 entries, i.e. 64 owners per level, and the
//    scan (not the atomics) becomes the cost.  The x term of the hash is x itself, so the slice of a corner is fixed by its (y, z) pair: a
//    counting sort by slice of the (sample, pair) items — count, exclusive scan, scatter; LDS integer atomics for the histograms — hands every
//    owner exactly its ~n/16 items, and the owner writes its 8192 final fp16 gradients directly (no partial copies, no combine pass).
constexpr uint32_t GB_FX_SLICE = GB_SLICE / 4;   // 4096 entries x 16 B = the same 64 KiB of LDS
constexpr uint32_t GB_FX_MAX_SLICES = 256;       // tables up to 2^20 entries
#ifndef NGP_GB_FX_CHUNK
#define NGP_GB_FX_CHUNK 2048
#endif
constexpr uint32_t GB_FX_CHUNK = NGP_GB_FX_CHUNK;   // samples per binning workgroup
constexpr uint32_t GB_FX_MAX_RESOLUTION = 1u << 20;   // (grid coordinates are exact in fp32 far beyond; the reference's finest level at aabb_scale 128 is 2^18)
__host__ __device__ __forceinline__ bool gb_uses_fx(uint32_t level_size, uint32_t resolution, bool dense) {
	return !dense && (level_size & (level_size - 1)) == 0 && level_size >= GB_FX_SLICE && level_size / GB_FX_SLICE <= GB_FX_MAX_SLICES && resolution <= GB_FX_MAX_RESOLUTION;
}
// scratch of the binned path: per level and slice {total, cursor, start} u32, then the item lists (one u32 per (sample, pair))
struct GbFxCounters { uint32_t totals[16][GB_FX_MAX_SLICES], cursors[16][GB_FX_MAX_SLICES]; };
constexpr uint32_t GB_FX_COUNTER_BYTES = 65536;

template <int D>
__host__ __device__ __forceinline__ bool level_is_dense(const NgpGridLevel& lv) {
	return (D == 3 ? (uint64_t)lv.resolution * lv.resolution * lv.resolution : (uint64_t)lv.resolution * lv.resolution) <= (uint64_t)lv.size;
}

// one term of tcnn's kernel_grid_backward, half(w * dL/dx), as an exact multiple of 2^-24 (integer part * 2^24 + fraction * 2^24, both native
// fp32 -> int32 conversions); inf / nan terms (a step the loss scaler is about to skip) would poison the integer sums and are dropped
__device__ __forceinline__ half_t gb_term_half(float w_times_g) {
	// tcnn: (T)(weight * grad) = fp32 product, THEN fp16.  The compiler would fold the caller's multiply into v_fma_mixlo_f16, which rounds
	// the exact product once; the empty asm keeps the fp32 rounding
	asm("" : "+v"(w_times_g));
	return (half_t)w_times_g;
}
__device__ __forceinline__ long long gb_half_fixed(half_t term) {
	const float t = (float)term;
	if (!(fabsf(t) < 65520.0f)) return 0ll;
	const float fl = floorf(t);
	return (long long)(int)fl * 16777216ll + (long long)(uint32_t)((t - fl) * 16777216.0f);
}
__device__ __forceinline__ long long gb_term_fixed(float w_times_g) { return gb_half_fixed(gb_term_half(w_times_g)); }
// exact fixed-point sum -> fp16 with ONE rounding: round to odd at 24 bits, then nearest-even to 11
__device__ __forceinline__ half_t gb_fixed_to_half(unsigned long long bits) {
	const long long a = (long long)bits;
	const unsigned long long m = a < 0 ? 0ull - (unsigned long long)a : (unsigned long long)a;
	float f;
	if (m < (1ull << 24)) f = (float)(uint32_t)m;
	else {
		const int sh = 40 - __builtin_clzll(m);
		unsigned long long top = m >> sh;
		if (m & ((1ull << sh) - 1ull)) top |= 1ull;
		f = ldexpf((float)(uint32_t)top, sh);
	}
	f *= 1.0f / 16777216.0f;
	return (half_t)(a < 0 ? -f : f);
}

// Dense (coarse) levels are binned too, by (slice, sample chunk): K private copies of a small table keep enough owners busy and
// grid_combine_kernel sums them.  Tens of consecutive ray samples share a cell there, i.e. the same 8 entries: the binning pass sums such
// runs itself (same exact 2^-24 fixed point) and emits one record {entry, sum0, sum1} per run and corner.  The owners of dense levels then
// add several times fewer, already merged, records and read them coalesced; the merging is spread over all the binning workgroups.
#ifndef NGP_GB_D_ITEMS
#define NGP_GB_D_ITEMS 16
#endif
constexpr uint32_t GB_D_ITEMS = NGP_GB_D_ITEMS;   // owners per dense level (slices x sample chunks) while the level has fewer slices
__host__ __device__ __forceinline__ GbSplit gb_dense_split(uint32_t level_size) {
	GbSplit s;
	s.n_slices = (level_size + GB_FX_SLICE - 1) / GB_FX_SLICE;
	s.k_chunks = GB_D_ITEMS / s.n_slices;
	if (s.k_chunks < 1) s.k_chunks = 1;   // then the owner writes the final fp16 values itself
	return s;
}
constexpr uint32_t GB_PARTIAL_LEVEL_BYTES = GB_ITEMS * GB_SLICE * 4u;   // private copies of one level: 4 MiB (dense: 16 x 64 KiB fixed point; fallback: 64 x 64 KiB half2)
static_assert(GB_D_ITEMS * GB_FX_SLICE * 16u <= GB_PARTIAL_LEVEL_BYTES, "partials");
constexpr uint32_t GB_ITEMS_PER_SAMPLE = 8;    // list stride per level: n * 8 items (hashed: 4 (y, z) pairs, dense: at most 8 records)

// where the list of `bin` starts inside its level: the exclusive prefix sum of the level's totals.  Every scatter workgroup (256 threads =
// 256 bins) scans them itself and every owner sums its own prefix — cheaper than a separate scan launch between count and scatter.
__device__ __forceinline__ uint32_t gb_block_exclusive_scan_256(uint32_t v, uint32_t* __restrict__ s_wave_totals /* [4] */) {
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off, 64); if ((int)lane >= off) incl += o; }
	if (lane == 63u) s_wave_totals[w] = incl;
	__syncthreads();
	uint32_t before = 0;
	for (uint32_t k = 0; k < w; ++k) before += s_wave_totals[k];
	return before + incl - v;
}
__device__ __forceinline__ uint32_t gb_owner_start(const GbFxCounters* __restrict__ ctr, uint32_t level, uint32_t bin, uint32_t* __restrict__ s_start) {
	if (threadIdx.x < 64u) {
		uint32_t part = 0;
#pragma unroll
		for (uint32_t q = 0; q < GB_FX_MAX_SLICES / 64u; ++q) { const uint32_t k = threadIdx.x + 64u * q; if (k < bin) part += ctr->totals[level][k]; }
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
		if (threadIdx.x == 0) *s_start = part;
	}
	__syncthreads();
	return *s_start;
}

// what a hashed level's item carries to its owner (12 bytes, in the level's run-sum region of the scratch): the two x-corner entries inside the slice, 12 bits each, and
// the four terms half(w_x0 * g0), half(w_x0 * g1), half(w_x1 * g0), half(w_x1 * g1)
struct GbRecord { uint32_t entries; half_t t[4]; };
static_assert(sizeof(GbRecord) == 12 && 4u * sizeof(GbRecord) <= GB_ITEMS_PER_SAMPLE * 16u, "records of a sample fit the level's run-sum slots");

// first bin of the sample chunk `chunk` (of n_chunks chunks of GB_FX_CHUNK samples) of a dense level: its k_chunks private copies are dealt to contiguous groups of chunks
__host__ __device__ __forceinline__ uint32_t gb_chunk_bin0(const GbSplit& sp, uint32_t chunk, uint32_t n_chunks) { return (uint32_t)(((uint64_t)chunk * sp.k_chunks) / n_chunks) * sp.n_slices; }

// passes 1 and 3 of the counting sort.  grid (ceil(n / GB_FX_CHUNK), 16 levels), block 256.  SCATTER = false: per-bin totals;
// SCATTER = true: reserve a range per bin (one global atomic per bin and workgroup) and write the items.
// hashed level: item = (sample, (y, z) pair) -> one GbRecord (two when the x corners straddle a slice boundary), bin = slice.
template <int D, bool SCATTER>
__device__ __forceinline__ void gb_bin_hashed(uint32_t* __restrict__ hist, uint32_t* __restrict__ base, uint32_t* __restrict__ side /* NI * GB_FX_CHUNK words of LDS */, const NgpGridLevel& lv, uint32_t level,
                                              const float* __restrict__ coords, uint32_t coord_stride, uint32_t n, const h2* __restrict__ dx_planes,
                                              GbFxCounters* __restrict__ ctr, ulonglong2* __restrict__ sums) {
	constexpr int NI = D == 3 ? 4 : 2;
	constexpr int PER = GB_FX_CHUNK / 256;
	const uint32_t hmask = lv.size - 1;
	const h2* __restrict__ dxl = dx_planes + (size_t)level * n;
	// The x term of the hash is x itself: below 4096 it never reaches the slice bits, so the slice of a pair is fixed by (y, z).  At finer levels (resolution >= 4096:
	// aabb_scale >= 4 with base.json, e.g. the fox scene's levels 14 and 15) corner x + 1 leaves corner x's slice only when x + 1 is a multiple of 4096 — such a pair
	// leaves as TWO records, one per corner, each with the other corner's terms zeroed (bit 31 of its code; the second record's rank waits in `side`).
	const bool straddle = lv.resolution >= GB_FX_SLICE;   // (uniform per workgroup)
	uint32_t code[PER][NI];   // bin << 16 | rank inside this workgroup, or ~0
	h2 gq[PER]; float px[PER], py[PER], pz[PER];
#pragma unroll
	for (int u = 0; u < PER; ++u) {
		const uint32_t s = blockIdx.x * GB_FX_CHUNK + u * 256 + threadIdx.x;
		const uint32_t sc = s < n ? s : 0;
		gq[u] = dxl[sc];
		const float* c = coords + (size_t)sc * coord_stride;
		if (D == 3) { const f3_t v = load_pos3(c); px[u] = v.x; py[u] = v.y; pz[u] = v.z; } else { px[u] = c[0]; py[u] = c[1]; pz[u] = 0.f; }
	}
#pragma unroll
	for (int u = 0; u < PER; ++u) {
		const uint32_t s = blockIdx.x * GB_FX_CHUNK + u * 256 + threadIdx.x;
		const uint32_t bits = __builtin_bit_cast(uint32_t, gq[u]) & 0x7fff7fffu;
		const bool live = s < n && bits != 0;   // adding +-0 never changes a sum
		const LevelPos p = level_pos(lv, px[u], py[u], pz[u]);
		const uint32_t hy[2] = {p.gy * 2654435761u, (p.gy + 1u) * 2654435761u};
		const uint32_t hz[2] = {D == 3 ? p.gz * 805459861u : 0u, D == 3 ? (p.gz + 1u) * 805459861u : 0u};
		const uint32_t xs = straddle ? p.gx : 0u;   // (x < 4096 cannot reach the slice bits)
#pragma unroll
		for (int m = 0; m < NI; ++m) {
			const uint32_t hb = hy[m & 1] ^ hz[m >> 1];
			const uint32_t bin = ((hb ^ xs) & hmask) / GB_FX_SLICE;
			uint32_t c = live ? ((bin << 16) | atomicAdd(&hist[bin], 1u)) : 0xffffffffu;
			if (straddle && live) {
				const uint32_t bin1 = ((hb ^ (p.gx + 1u)) & hmask) / GB_FX_SLICE;
				if (bin1 != bin) {
					const uint32_t r1 = atomicAdd(&hist[bin1], 1u);
					if (SCATTER) { side[(u * NI + m) * 256 + threadIdx.x] = (bin1 << 16) | r1; c |= 0x80000000u; }   // (the count pass has no `side`: its LDS is the histogram and one word per sample)
				}
			}
			code[u][m] = c;
		}
	}
	__syncthreads();
	if (!SCATTER) {
		if (threadIdx.x < GB_FX_MAX_SLICES && hist[threadIdx.x]) atomicAdd(&ctr->totals[level][threadIdx.x], hist[threadIdx.x]);
		return;
	}
	{
		static_assert(GB_FX_MAX_SLICES == 256, "one thread per bin");
		const uint32_t start = gb_block_exclusive_scan_256(ctr->totals[level][threadIdx.x], base /* scratch: 4 words, overwritten below */);
		__syncthreads();
		base[threadIdx.x] = hist[threadIdx.x] ? start + atomicAdd(&ctr->cursors[level][threadIdx.x], hist[threadIdx.x]) : 0u;
	}
	__syncthreads();
	// An item leaves as a finished RECORD: the two x-corner entries inside the slice and their four fp16 terms half(w * dL/dx) — this pass has the position and dL/dx in
	// registers anyway.  The owners then stream 12-byte records instead of gathering a position and a dL/dx pair per item (two random L2 requests each: what bound them).
	GbRecord* __restrict__ out = (GbRecord*)sums;   // the level's own record space (gb_level_records)
#pragma unroll
	for (int u = 0; u < PER; ++u) {
		const LevelPos p = level_pos(lv, px[u], py[u], pz[u]);
		const float g0 = (float)gq[u][0], g1 = (float)gq[u][1];
#pragma unroll
		for (int m = 0; m < NI; ++m) {
			const uint32_t c = code[u][m];
			if (c == 0xffffffffu) continue;
			const uint32_t yb = m & 1u, zb = m >> 1;
			const uint32_t hb = ((p.gy + yb) * 2654435761u) ^ (D == 3 ? (p.gz + zb) * 805459861u : 0u);
			const float wy = yb ? p.fy : (1.0f - p.fy), wz = zb ? p.fz : (1.0f - p.fz);
			GbRecord r;
			uint32_t e[2];
#pragma unroll
			for (uint32_t xb = 0; xb < 2; ++xb) {
				e[xb] = ((hb ^ (p.gx + xb)) & hmask) & (GB_FX_SLICE - 1);
				float w = (xb ? p.fx : (1.0f - p.fx)) * wy;
				if (D == 3) w *= wz;
				r.t[2 * xb] = gb_term_half(w * g0); r.t[2 * xb + 1] = gb_term_half(w * g1);
			}
			if (straddle && (c & 0x80000000u)) {   // one cell in 4096: the x + 1 corner's record goes to its own slice, this one keeps the x corner
				const uint32_t c1 = side[(u * NI + m) * 256 + threadIdx.x];
				GbRecord q;
				q.entries = e[1] | (e[1] << 12); q.t[0] = (half_t)0.0f; q.t[1] = (half_t)0.0f; q.t[2] = r.t[2]; q.t[3] = r.t[3];
				out[base[c1 >> 16] + (c1 & 0xffffu)] = q;
				e[1] = e[0]; r.t[2] = (half_t)0.0f; r.t[3] = (half_t)0.0f;
			}
			r.entries = e[0] | (e[1] << 12);
			out[base[(c >> 16) & 0xffu] + (c & 0xffffu)] = r;
		}
	}
}

// dense level: record = {entry inside its slice, two fixed-point run sums}, bin = slice + n_slices * sample chunk.  The workgroup stages its
// 2048 samples in LDS (coalesced loads), then every thread walks 8 CONSECUTIVE samples and sums the 8 corner terms in registers while the
// cell stays the same; a record per corner leaves when the cell changes.  (Padded LDS index i + i/8: the walk is bank-conflict free.)
constexpr uint32_t GB_STAGE = GB_FX_CHUNK + GB_FX_CHUNK / 8;
// levels whose slices fit the 256 bins are binned; larger dense tables (log2_hashmap_size >= 21: res^3 above 2^20 entries) take the float path of the owners
__host__ __device__ __forceinline__ bool gb_dense_binned(uint32_t level_size) { return (level_size + GB_FX_SLICE - 1) / GB_FX_SLICE <= GB_FX_MAX_SLICES; }
// The walk's notion of "same cell", shared by the two passes (they must emit the same records): the cell coordinates packed 10 bits each — every dense level the
// binning takes has a resolution below 1024 — or, for a coordinate outside that (a position far outside the unit cube), a key of its own that merges with nothing.
__device__ __forceinline__ uint32_t gb_cell_key(const LevelPos& p, uint32_t i) {
	return (p.gx | p.gy | p.gz) < 1024u ? (p.gx | (p.gy << 10) | (p.gz << 20)) : (0x80000000u | i);
}
// the 2^D corner entries of a cell of a DENSE level: grid_index_nd's x + y res + z res^2 (same uint32 arithmetic, same wrap) for corner 0 plus a per-corner offset
template <int D>
__device__ __forceinline__ uint32_t gb_dense_corner(const NgpGridLevel& lv, uint32_t raw0, int k) {
	uint32_t raw = raw0 + (uint32_t)(k & 1) + (((k >> 1) & 1) ? lv.resolution : 0u);
	if (D == 3 && ((k >> 2) & 1)) raw += lv.resolution * lv.resolution;
	return raw >= lv.size ? raw % lv.size : raw;   // (power-of-two sizes: the same value as grid_index's mask)
}
template <int D>
__device__ __forceinline__ uint32_t gb_dense_raw0(const NgpGridLevel& lv, uint32_t gx, uint32_t gy, uint32_t gz) {
	return gx + gy * lv.resolution + (D == 3 ? gz * lv.resolution * lv.resolution : 0u);
}
// count pass: the keys of the chunk's samples sit in LDS (~0 = nothing to add); one flush per run and thread, 2^D histogram increments each
template <int D>
__device__ __forceinline__ void gb_dense_count_walk(const uint32_t* __restrict__ s_key, uint32_t* __restrict__ counter, const NgpGridLevel& lv, uint32_t chunk_bin0,
                                                    const float* __restrict__ chunk_coords, uint32_t coord_stride) {
	constexpr int NC = 1 << D;
	constexpr uint32_t RUN = GB_FX_CHUNK / 256;
	uint32_t cur = 0xffffffffu;
	auto flush = [&]() {
		uint32_t gx = cur & 1023u, gy = (cur >> 10) & 1023u, gz = cur >> 20;
		if (cur & 0x80000000u) {   // (rare) a sample outside the packable range: its cell from the position itself
			const float* c = chunk_coords + (size_t)(cur & 0x7fffffffu) * coord_stride;
			const LevelPos p = level_pos(lv, c[0], c[1], D == 3 ? c[2] : 0.f);
			gx = p.gx; gy = p.gy; gz = p.gz;
		}
		const uint32_t raw0 = gb_dense_raw0<D>(lv, gx, gy, gz);
#pragma unroll
		for (int k = 0; k < NC; ++k) atomicAdd(&counter[chunk_bin0 + gb_dense_corner<D>(lv, raw0, k) / GB_FX_SLICE], 1u);
	};
#pragma unroll
	for (uint32_t u = 0; u < RUN; ++u) {
		const uint32_t i = threadIdx.x * RUN + u;
		const uint32_t key = s_key[i + (i >> 3)];
		if (key == 0xffffffffu) continue;
		if (key != cur) { if (cur != 0xffffffffu) flush(); cur = key; }
	}
	if (cur != 0xffffffffu) flush();
}

template <int D, bool WRITE>
__device__ __forceinline__ void gb_dense_walk(const uint32_t* __restrict__ s_g, const float* __restrict__ s_px, const float* __restrict__ s_py, const float* __restrict__ s_pz,
                                              uint32_t* __restrict__ counter, const NgpGridLevel& lv, uint32_t chunk_bin0, uint32_t n_live,
                                              uint32_t* __restrict__ out_e, ulonglong2* __restrict__ out_v) {
	constexpr int NC = 1 << D;
	constexpr uint32_t RUN = GB_FX_CHUNK / 256;
	uint32_t cgx = 0, cgy = 0, cgz = 0, cur = 0xffffffffu;
	long long a0[NC], a1[NC];
#pragma unroll
	for (int k = 0; k < NC; ++k) { a0[k] = 0; a1[k] = 0; }
	auto flush = [&]() {
		const uint32_t raw0 = gb_dense_raw0<D>(lv, cgx, cgy, cgz);
#pragma unroll
		for (int k = 0; k < NC; ++k) {
			const uint32_t idx = gb_dense_corner<D>(lv, raw0, k);
			const uint32_t pos = atomicAdd(&counter[chunk_bin0 + idx / GB_FX_SLICE], 1u);
			if (WRITE) { out_e[pos] = idx % GB_FX_SLICE; out_v[pos] = make_ulonglong2((unsigned long long)a0[k], (unsigned long long)a1[k]); }
		}
	};
#pragma unroll
	for (uint32_t u = 0; u < RUN; ++u) {
		const uint32_t i = threadIdx.x * RUN + u, ip = i + (i >> 3);
		const uint32_t gbits = s_g[ip];
		if (i >= n_live || (gbits & 0x7fff7fffu) == 0) continue;   // adding +-0 never changes a sum
		const LevelPos p = level_pos(lv, s_px[ip], s_py[ip], D == 3 ? s_pz[ip] : 0.f);
		const uint32_t key = gb_cell_key(p, i);
		if (key != cur) {
			if (cur != 0xffffffffu) flush();
			cur = key; cgx = p.gx; cgy = p.gy; cgz = p.gz;
#pragma unroll
			for (int k = 0; k < NC; ++k) { a0[k] = 0; a1[k] = 0; }
		}
		if (WRITE) {
			const h2 g = __builtin_bit_cast(h2, gbits);
			const float g0 = (float)g[0], g1 = (float)g[1];
#pragma unroll
			for (int k = 0; k < NC; ++k) {
				float w = (k & 1) ? p.fx : (1.0f - p.fx);
				w *= ((k >> 1) & 1) ? p.fy : (1.0f - p.fy);
				if (D == 3) w *= ((k >> 2) & 1) ? p.fz : (1.0f - p.fz);
				a0[k] += gb_term_fixed(w * g0); a1[k] += gb_term_fixed(w * g1);
			}
		}
	}
	if (cur != 0xffffffffu) flush();
}

template <int D, bool SCATTER>
__device__ __forceinline__ void gb_bin_dense(uint32_t* __restrict__ hist, uint32_t* __restrict__ base, uint32_t* __restrict__ stage, const NgpGridLevel& lv, uint32_t level,
                                             const float* __restrict__ coords, uint32_t coord_stride, uint32_t n, const h2* __restrict__ dx_planes,
                                             GbFxCounters* __restrict__ ctr, uint32_t* __restrict__ items, ulonglong2* __restrict__ sums, uint32_t* __restrict__ wg_hist) {
	constexpr int PER = GB_FX_CHUNK / 256;
	const GbSplit sp = gb_dense_split(lv.size);
	// the count pass leaves this workgroup's per-bin record counts behind (256 words): the scatter pass picks them up instead of walking twice
	const uint32_t n_chunks = (n + GB_FX_CHUNK - 1) / GB_FX_CHUNK;   // (not gridDim.x: the scatter launch's grid is cut for the staged levels' smaller chunks)
	uint32_t* __restrict__ my_hist = wg_hist + ((size_t)level * n_chunks + blockIdx.x) * GB_FX_MAX_SLICES;
	uint32_t saved = 0;
	if (SCATTER && threadIdx.x < GB_FX_MAX_SLICES) saved = my_hist[threadIdx.x];
	const uint32_t chunk_bin0 = gb_chunk_bin0(sp, blockIdx.x, (n + GB_FX_CHUNK - 1) / GB_FX_CHUNK);
	const h2* __restrict__ dxl = dx_planes + (size_t)level * n;
	const uint32_t s_first = blockIdx.x * GB_FX_CHUNK;
	const uint32_t n_live = n - s_first < GB_FX_CHUNK ? n - s_first : GB_FX_CHUNK;
	if (!SCATTER) {
		// count: only the cell KEYS go through LDS (one word per sample: the pass fits 11 KiB and every workgroup of the launch is resident at once)
#pragma unroll
		for (int u = 0; u < PER; ++u) {
			const uint32_t i = u * 256 + threadIdx.x;
			uint32_t key = 0xffffffffu;
			if (i < n_live) {
				const uint32_t gbits = __builtin_bit_cast(uint32_t, dxl[s_first + i]);
				if (gbits & 0x7fff7fffu) {   // adding +-0 never changes a sum
					const float* c = coords + (size_t)(s_first + i) * coord_stride;
					float x, y, z = 0.f;
					if (D == 3) { const f3_t v = load_pos3(c); x = v.x; y = v.y; z = v.z; } else { x = c[0]; y = c[1]; }
					key = gb_cell_key(level_pos(lv, x, y, z), i);
				}
			}
			stage[i + (i >> 3)] = key;
		}
		__syncthreads();
		gb_dense_count_walk<D>(stage, hist, lv, chunk_bin0, coords + (size_t)s_first * coord_stride, coord_stride);
		__syncthreads();
		if (threadIdx.x < GB_FX_MAX_SLICES) {
			my_hist[threadIdx.x] = hist[threadIdx.x];
			if (hist[threadIdx.x]) atomicAdd(&ctr->totals[level][threadIdx.x], hist[threadIdx.x]);
		}
		return;
	}
	uint32_t* s_g = stage;
	float* s_px = (float*)(stage + GB_STAGE); float* s_py = (float*)(stage + 2 * GB_STAGE); float* s_pz = (float*)(stage + 3 * GB_STAGE);
#pragma unroll
	for (int u = 0; u < PER; ++u) {
		const uint32_t i = u * 256 + threadIdx.x, ip = i + (i >> 3);
		const uint32_t sc = i < n_live ? s_first + i : s_first;
		const float* c = coords + (size_t)sc * coord_stride;
		s_g[ip] = __builtin_bit_cast(uint32_t, dxl[sc]);
		if (D == 3) { const f3_t v = load_pos3(c); s_px[ip] = v.x; s_py[ip] = v.y; s_pz[ip] = v.z; } else { s_px[ip] = c[0]; s_py[ip] = c[1]; }
	}
	if (SCATTER && threadIdx.x < GB_FX_MAX_SLICES) hist[threadIdx.x] = saved;
	__syncthreads();
	// reserve the ranges (base[] then serves as the running cursor of each bin), walk again with the sums
	{
		static_assert(GB_FX_MAX_SLICES == 256, "one thread per bin");
		const uint32_t start = gb_block_exclusive_scan_256(ctr->totals[level][threadIdx.x], base /* scratch: 4 words, overwritten below */);
		__syncthreads();
		base[threadIdx.x] = hist[threadIdx.x] ? start + atomicAdd(&ctr->cursors[level][threadIdx.x], hist[threadIdx.x]) : 0u;
	}
	__syncthreads();
	gb_dense_walk<D, true>(s_g, s_px, s_py, s_pz, base, lv, chunk_bin0, n_live, items, sums);
}

// Dense level of a batch WITHOUT ray order (image fitting, SDF: stratified / random positions — consecutive samples share no cell, so the run-merging walk above emits 8 (4 in 2-D) 20-byte
// records per sample and merges nothing): the level is binned like a hashed one instead — one 12-byte GbRecord per (sample, pair of x-neighbour corners), whose entries are adjacent in a dense
// table; a pair that straddles a 4096-entry slice leaves as two records — into the dense owners' bins (slice + n_slices * sample chunk group).  A third of the bytes, no serial walk.
template <int D, bool SCATTER>
__device__ __forceinline__ void gb_bin_dense_pairs(uint32_t* __restrict__ hist, uint32_t* __restrict__ base, uint32_t* __restrict__ side, const NgpGridLevel& lv, uint32_t level,
                                                   const float* __restrict__ coords, uint32_t coord_stride, uint32_t n, const h2* __restrict__ dx_planes,
                                                   GbFxCounters* __restrict__ ctr, ulonglong2* __restrict__ sums) {
	constexpr int NI = D == 3 ? 4 : 2;
	constexpr int PER = GB_FX_CHUNK / 256;
	const GbSplit sp = gb_dense_split(lv.size);
	const uint32_t chunk_bin0 = gb_chunk_bin0(sp, blockIdx.x, (n + GB_FX_CHUNK - 1) / GB_FX_CHUNK);
	const h2* __restrict__ dxl = dx_planes + (size_t)level * n;
	uint32_t code[PER][NI];   // bin << 16 | rank inside this workgroup, bit 31: the pair straddles two slices (second rank in `side`), or ~0
	h2 gq[PER]; float px[PER], py[PER], pz[PER];
#pragma unroll
	for (int u = 0; u < PER; ++u) {
		const uint32_t s = blockIdx.x * GB_FX_CHUNK + u * 256 + threadIdx.x;
		const uint32_t sc = s < n ? s : 0;
		gq[u] = dxl[sc];
		const float* c = coords + (size_t)sc * coord_stride;
		if (D == 3) { const f3_t v = load_pos3(c); px[u] = v.x; py[u] = v.y; pz[u] = v.z; } else { px[u] = c[0]; py[u] = c[1]; pz[u] = 0.f; }
	}
#pragma unroll
	for (int u = 0; u < PER; ++u) {
		const uint32_t s = blockIdx.x * GB_FX_CHUNK + u * 256 + threadIdx.x;
		const uint32_t bits = __builtin_bit_cast(uint32_t, gq[u]) & 0x7fff7fffu;
		const bool live = s < n && bits != 0;   // adding +-0 never changes a sum
		const LevelPos p = level_pos(lv, px[u], py[u], pz[u]);
#pragma unroll
		for (int m = 0; m < NI; ++m) {
			const uint32_t i0 = grid_index_nd<D>(lv, p.gx, p.gy + (m & 1), p.gz + (m >> 1)), i1 = grid_index_nd<D>(lv, p.gx + 1u, p.gy + (m & 1), p.gz + (m >> 1));
			const uint32_t bin = chunk_bin0 + i0 / GB_FX_SLICE, bin1 = chunk_bin0 + i1 / GB_FX_SLICE;
			uint32_t c = live ? ((bin << 16) | atomicAdd(&hist[bin], 1u)) : 0xffffffffu;
			if (live && bin1 != bin) {
				const uint32_t r1 = atomicAdd(&hist[bin1], 1u);
				if (SCATTER) { side[(u * NI + m) * 256 + threadIdx.x] = (bin1 << 16) | r1; c |= 0x80000000u; }
			}
			code[u][m] = c;
		}
	}
	__syncthreads();
	if (!SCATTER) {
		if (threadIdx.x < GB_FX_MAX_SLICES && hist[threadIdx.x]) atomicAdd(&ctr->totals[level][threadIdx.x], hist[threadIdx.x]);
		return;
	}
	{
		const uint32_t start = gb_block_exclusive_scan_256(ctr->totals[level][threadIdx.x], base /* scratch: 4 words, overwritten below */);
		__syncthreads();
		base[threadIdx.x] = hist[threadIdx.x] ? start + atomicAdd(&ctr->cursors[level][threadIdx.x], hist[threadIdx.x]) : 0u;
	}
	__syncthreads();
	GbRecord* __restrict__ out = (GbRecord*)sums;   // the level's own record space (gb_level_records)
#pragma unroll
	for (int u = 0; u < PER; ++u) {
		const LevelPos p = level_pos(lv, px[u], py[u], pz[u]);
		const float g0 = (float)gq[u][0], g1 = (float)gq[u][1];
#pragma unroll
		for (int m = 0; m < NI; ++m) {
			const uint32_t c = code[u][m];
			if (c == 0xffffffffu) continue;
			const uint32_t yb = m & 1u, zb = m >> 1;
			const float wy = yb ? p.fy : (1.0f - p.fy), wz = zb ? p.fz : (1.0f - p.fz);
			GbRecord r;
			uint32_t e[2];
#pragma unroll
			for (uint32_t xb = 0; xb < 2; ++xb) {
				e[xb] = grid_index_nd<D>(lv, p.gx + xb, p.gy + yb, p.gz + zb) & (GB_FX_SLICE - 1);
				float w = (xb ? p.fx : (1.0f - p.fx)) * wy;
				if (D == 3) w *= wz;
				r.t[2 * xb] = gb_term_half(w * g0); r.t[2 * xb + 1] = gb_term_half(w * g1);
			}
			if (c & 0x80000000u) {   // the x + 1 corner's record goes to its own slice, this one keeps the x corner
				const uint32_t c1 = side[(u * NI + m) * 256 + threadIdx.x];
				GbRecord q;
				q.entries = e[1] | (e[1] << 12); q.t[0] = (half_t)0.0f; q.t[1] = (half_t)0.0f; q.t[2] = r.t[2]; q.t[3] = r.t[3];
				out[base[c1 >> 16] + (c1 & 0xffffu)] = q;
				e[1] = e[0]; r.t[2] = (half_t)0.0f; r.t[3] = (half_t)0.0f;
			}
			r.entries = e[0] | (e[1] << 12);
			out[base[(c >> 16) & 0xffu] + (c & 0xffffu)] = r;
		}
	}
}

// The sum of the fused MLP backward's per-workgroup weight-gradient partials rides in the count pass's launch as extra rows of workgroups (it depends on the same
// kernel and on nothing else; as a launch of its own it was 7 us of the chain).  Same arithmetic as wgrad_reduce_kernel, bit for bit: "wave" q of 16 sums every 16th
// partial with 8 accumulators, then the 16 sums in order — here 4 waves take 4 q each, their loads in flight together.
struct WgradJob { const float* partials; uint32_t n_chunks; half_t* grads; uint32_t n_params; };
__device__ __forceinline__ void wgrad_reduce_rows(const WgradJob& j, uint32_t wg, float* __restrict__ red /* [16][64] */) {
	if (wg * 64u >= j.n_params) return;
	const uint32_t p = threadIdx.x & 63u, w = threadIdx.x >> 6;
	const uint32_t i = wg * 64u + p, n = j.n_params;
#pragma unroll 1
	for (uint32_t half = 0; half < 2u; ++half) {   // two q at a time: 16 accumulators + 16 loads in flight keep the count pass at 8 waves per SIMD
		float acc[2][8];
#pragma unroll
		for (int qq = 0; qq < 2; ++qq)
#pragma unroll
			for (int u = 0; u < 8; ++u) acc[qq][u] = 0.f;
		const uint32_t q0 = w + 8u * half;   // q0 and q0 + 4
		if (i < n) {
			uint32_t t0 = 0;   // chunks [t0, t0 + 128) are full steps for every q
			for (; t0 + 128u <= j.n_chunks; t0 += 128u) {
#pragma unroll
				for (int qq = 0; qq < 2; ++qq)
#pragma unroll
					for (int u = 0; u < 8; ++u) acc[qq][u] += j.partials[(size_t)(t0 + q0 + 4u * qq + 16u * u) * n + i];
			}
#pragma unroll
			for (int qq = 0; qq < 2; ++qq) {
				uint32_t c = t0 + q0 + 4u * qq;
				for (; c + 7u * 16u < j.n_chunks; c += 128u) {
#pragma unroll
					for (int u = 0; u < 8; ++u) acc[qq][u] += j.partials[(size_t)(c + 16u * u) * n + i];
				}
				for (int u = 0; c < j.n_chunks; c += 16u, ++u) acc[qq][u & 7] += j.partials[(size_t)c * n + i];
			}
		}
#pragma unroll
		for (int qq = 0; qq < 2; ++qq) red[(q0 + 4u * qq) * 64u + p] = ((acc[qq][0] + acc[qq][1]) + (acc[qq][2] + acc[qq][3])) + ((acc[qq][4] + acc[qq][5]) + (acc[qq][6] + acc[qq][7]));
	}
	__syncthreads();
	if (w == 0 && i < n) {
		float sum = 0.0f;
#pragma unroll
		for (uint32_t k = 0; k < 16u; ++k) sum += red[k * 64u + p];
		j.grads[i] = (half_t)sum;
	}
}

// Record space of the binned path: the levels' records are packed back to back, each level with the worst case of ITS kind — a hashed level one 12-byte record per
// (sample, (y, z) corner pair), two where the level is fine enough for the x corners to straddle a slice (resolution >= 4096); a dense one at most 8 merged records per
// sample, {entry} (4 bytes, the level's first n * 8 words) and {two 64-bit sums} (16 bytes, behind them), or, for a batch that is not in ray order, its pair records (at
// most two per pair: 96 bytes); a level of the float path nothing.  (Up to round 5 every level had the dense worst case: 160 bytes per sample and level, 671 MB at
// n = 2^18 whatever the level table; base.json's 5 dense + 11 hashed levels need 348 MB.)  The host lays the levels out and hands the offsets to the kernels.
template <int D>
__host__ __device__ __forceinline__ uint64_t gb_level_record_bytes(const NgpGridLevel& lv, uint32_t n) {
	const bool dense = level_is_dense<D>(lv);
	if (gb_uses_fx(lv.size, lv.resolution, dense)) return (uint64_t)n * (D == 3 ? 4u : 2u) * (lv.resolution >= GB_FX_SLICE ? 2u : 1u) * sizeof(GbRecord);
	if (dense && gb_dense_binned(lv.size)) return (uint64_t)n * GB_ITEMS_PER_SAMPLE * (4u + 16u);
	return 0;
}
struct GbRecordOffsets { uint64_t off[17]; };   // byte offset of every level's record space; [16] = the total
template <int D>
static GbRecordOffsets gb_record_offsets(const NgpNetDesc* desc_host, uint32_t n) {   // no level table at hand: every level at the dense worst case
	GbRecordOffsets r;
	uint64_t total = 0;
	for (int l = 0; l < 16; ++l) { r.off[l] = total; total += desc_host ? gb_level_record_bytes<D>(desc_host->levels[l], n) : (uint64_t)n * GB_ITEMS_PER_SAMPLE * (4u + 16u); }
	r.off[16] = total;
	return r;
}
struct GbLevelRecords { uint32_t* items; ulonglong2* sums; };
__device__ __forceinline__ GbLevelRecords gb_level_records(const GbRecordOffsets& offsets, uint32_t level, uint32_t n, void* records, bool dense_merged) {
	const uint64_t off = offsets.off[level];
	GbLevelRecords r;
	r.items = (uint32_t*)((char*)records + off);
	r.sums = (ulonglong2*)((char*)records + off + (dense_merged ? (uint64_t)n * GB_ITEMS_PER_SAMPLE * 4u : 0ull));
	return r;
}
static uint64_t gb_wg_hist_bytes(uint32_t n) { return (uint64_t)16 * ((n + GB_FX_CHUNK - 1) / GB_FX_CHUNK) * GB_FX_MAX_SLICES * 4u; }
template <int D>
static uint64_t gb_records_bytes(const NgpNetDesc* desc_host, uint32_t n) { return gb_record_offsets<D>(desc_host, n).off[16]; }

// private copies of the levels: a dense level's (level * GB_D_ITEMS + bin) fixed-point slices fill 1 MiB per level; only a level of the float path needs its 4 MiB
template <int D>
static uint64_t gb_partials_bytes(const NgpNetDesc* desc_host) {
	if (!desc_host) return (uint64_t)16 * GB_PARTIAL_LEVEL_BYTES;
	for (int l = 0; l < 16; ++l) {
		const NgpGridLevel& lv = desc_host->levels[l];
		const bool dense = level_is_dense<D>(lv);
		if (!gb_uses_fx(lv.size, lv.resolution, dense) && !(dense && gb_dense_binned(lv.size))) return (uint64_t)16 * GB_PARTIAL_LEVEL_BYTES;
	}
	return (uint64_t)16 * GB_D_ITEMS * GB_FX_SLICE * 16u;
}

// waves per SIMD of the scatter pass for batches that are not in ray order (SDF: dense levels as pair records): at 4 (128 registers) the kernel spills 37 registers; at 3 it does
// not and the SDF backward group is 2.5-5 % shorter (237-245 -> 231-233 us; 2: no gain) — tools: NGP_EXTRA_HIP_FLAGS=-DNGP_GB_UNORDERED_SCATTER_WAVES=n python build.py --force
#ifndef NGP_GB_ORDERED_SCATTER_WAVES
#define NGP_GB_ORDERED_SCATTER_WAVES 4   // ray-ordered batches; round 5, lego step 0.512-0.518 ms at 4, 0.532-0.547 at 3, 0.536-0.540 at 5 (fox 0.638-0.639 / 0.642-0.653 / 0.641-0.654)
#endif
#ifndef NGP_GB_UNORDERED_SCATTER_WAVES
#define NGP_GB_UNORDERED_SCATTER_WAVES 3
#endif
template <int D, bool SCATTER, bool ORDERED = true>   // ORDERED: the batch is in ray order (NeRF training): dense levels merge runs of samples that share a cell; false: pair records
// (38 KiB of LDS: four workgroups per CU = four waves per SIMD; the register cap keeps the kernel there — and inside what the run-ahead march leaves beside it)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SCATTER ? (ORDERED ? NGP_GB_ORDERED_SCATTER_WAVES : NGP_GB_UNORDERED_SCATTER_WAVES) : 8, 8))) gb_fx_bin_kernel(const NgpNetDesc* __restrict__ desc, const float* __restrict__ coords, uint32_t coord_stride, uint32_t n,
                                                        const h2* __restrict__ dx_planes, GbFxCounters* __restrict__ ctr, void* __restrict__ records, const GbRecordOffsets rec_off, uint32_t* __restrict__ wg_hist, uint32_t level_mask,
                                                        WgradJob wgrad, const uint32_t* __restrict__ n_live) {
	NGP_RAISE_CHAIN_PRIORITY();
	static_assert(GB_FX_CHUNK * 8 <= 65536, "rank field");
	static_assert(GB_STAGE * 4u >= 16u * 64u * 4u, "the weight-gradient rows' 16 x 64 sums fit the count pass's LDS words");
	static_assert(4 * GB_FX_CHUNK <= 4 * GB_STAGE, "the hashed path's side ranks fit the dense path's staging words");
	__shared__ uint32_t hist[GB_FX_MAX_SLICES], base[GB_FX_MAX_SLICES], stage[SCATTER ? 4 * GB_STAGE : GB_STAGE];   // count: 11 KiB, every workgroup of the launch resident at once; scatter: 38 KiB
	const uint32_t level = blockIdx.y;
	if (level >= 16u) {   // rows behind the 16 levels (count pass only, when the caller handed a job over)
		if (!SCATTER && wgrad.partials) wgrad_reduce_rows(wgrad, (level - 16u) * gridDim.x + blockIdx.x, (float*)stage);
		return;
	}
	if (!((level_mask >> level) & 1u)) return;   // dev-only ablation, see grid_backward_kernel
	if (n_live && blockIdx.x * GB_FX_CHUNK >= *n_live) return;   // a chunk behind the live samples: its dL/dx is all zeros, it leaves no record (both passes skip it alike)
	const NgpGridLevel lv = desc->levels[level];
	const bool dense = level_is_dense<D>(lv) && gb_dense_binned(lv.size);
	const bool fx = gb_uses_fx(lv.size, lv.resolution, level_is_dense<D>(lv));
	if (!fx && !dense) return;   // float path of the owners: no binning
	if (threadIdx.x < GB_FX_MAX_SLICES) hist[threadIdx.x] = 0;
	__syncthreads();
	const GbLevelRecords rec = gb_level_records(rec_off, level, n, records, dense && ORDERED);
	if (dense && !ORDERED) gb_bin_dense_pairs<D, SCATTER>(hist, base, stage, lv, level, coords, coord_stride, n, dx_planes, ctr, rec.sums);
	else if (dense) gb_bin_dense<D, SCATTER>(hist, base, stage, lv, level, coords, coord_stride, n, dx_planes, ctr, rec.items, rec.sums, wg_hist);
	else gb_bin_hashed<D, SCATTER>(hist, base, stage, lv, level, coords, coord_stride, n, dx_planes, ctr, rec.sums);
}

// pass 4: the owner of (level, slice) adds its items into 8192 x 2 64-bit fixed-point words in LDS and writes the final fp16 gradients
template <int D>
__device__ __forceinline__ void gb_fx_accumulate(unsigned long long* __restrict__ slice64, const NgpGridLevel& lv, uint32_t level, uint32_t sl, uint32_t n,
                                                 const GbFxCounters* __restrict__ ctr, const ulonglong2* __restrict__ sums, h2* __restrict__ grid_grad, uint32_t* __restrict__ s_start) {
	if (sl >= lv.size / GB_FX_SLICE) return;
	h2* __restrict__ dst = grid_grad + lv.offset + (size_t)sl * GB_FX_SLICE;
	const uint32_t count = ctr->totals[level][sl];
	if (count == 0) {
		const h2 z2 = {(half_t)0.0f, (half_t)0.0f};
		for (uint32_t i = threadIdx.x; i < GB_FX_SLICE; i += blockDim.x) dst[i] = z2;
		return;
	}
	for (uint32_t i = threadIdx.x; i < 2 * GB_FX_SLICE; i += blockDim.x) slice64[i] = 0ull;
	__syncthreads();
	const GbRecord* __restrict__ my = (const GbRecord*)sums + gb_owner_start(ctr, level, sl, s_start);
	constexpr uint32_t UN = 8;
	for (uint32_t i0 = threadIdx.x; i0 < count; i0 += blockDim.x * UN) {
		GbRecord r[UN];
#pragma unroll
		for (uint32_t u = 0; u < UN; ++u) { const uint32_t i = i0 + u * blockDim.x; r[u] = my[i < count ? i : 0]; }
#pragma unroll
		for (uint32_t u = 0; u < UN; ++u) {
			if (i0 + u * blockDim.x >= count) continue;
#pragma unroll
			for (uint32_t xb = 0; xb < 2; ++xb) {
				// tcnn kernel_grid_backward adds half2(w * dL/dx): the terms were rounded to fp16 like there (by the binning pass), their sum is exact
				const long long v0 = gb_half_fixed(r[u].t[2 * xb]), v1 = gb_half_fixed(r[u].t[2 * xb + 1]);
				const uint32_t e = (r[u].entries >> (12u * xb)) & (GB_FX_SLICE - 1);
				if (v0) atomicAdd(&slice64[2 * e], (unsigned long long)v0);
				if (v1) atomicAdd(&slice64[2 * e + 1], (unsigned long long)v1);
			}
		}
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < GB_FX_SLICE; i += blockDim.x) {
		h2 o; o[0] = gb_fixed_to_half(slice64[2 * i]); o[1] = gb_fixed_to_half(slice64[2 * i + 1]);
		dst[i] = o;
	}
}

// Dense (coarse) levels: the owner of (slice, sample chunk) adds its merged records (see gb_bin_dense)
__device__ __forceinline__ void gb_dense_owner(unsigned long long* __restrict__ slice64, const NgpGridLevel& lv, uint32_t level, uint32_t bin, uint32_t n,
                                               const GbFxCounters* __restrict__ ctr, const uint32_t* __restrict__ items, const ulonglong2* __restrict__ sums,
                                               unsigned long long* __restrict__ partials, h2* __restrict__ grid_grad, uint32_t* __restrict__ s_start) {
	const GbSplit sp = gb_dense_split(lv.size);
	if (bin >= sp.n_slices * sp.k_chunks) return;
	const uint32_t sl = bin % sp.n_slices;
	const uint32_t lo = sl * GB_FX_SLICE;
	const uint32_t cnt = (lv.size - lo) < GB_FX_SLICE ? (lv.size - lo) : GB_FX_SLICE;
	const uint32_t count = ctr->totals[level][bin];
	for (uint32_t i = threadIdx.x; i < 2 * cnt; i += blockDim.x) slice64[i] = 0ull;
	__syncthreads();
	const size_t first = gb_owner_start(ctr, level, bin, s_start);
	const uint32_t* __restrict__ my_e = items + first;
	const ulonglong2* __restrict__ my_v = sums + first;
	constexpr uint32_t UN = 4;
	for (uint32_t i0 = threadIdx.x; i0 < count; i0 += blockDim.x * UN) {
		uint32_t e[UN]; ulonglong2 v[UN];
#pragma unroll
		for (uint32_t u = 0; u < UN; ++u) { const uint32_t i = i0 + u * blockDim.x; const uint32_t ic = i < count ? i : 0; e[u] = my_e[ic]; v[u] = my_v[ic]; }
#pragma unroll
		for (uint32_t u = 0; u < UN; ++u) {
			if (i0 + u * blockDim.x >= count) continue;
			if (v[u].x) atomicAdd(&slice64[2 * e[u]], v[u].x);
			if (v[u].y) atomicAdd(&slice64[2 * e[u] + 1], v[u].y);
		}
	}
	__syncthreads();
	if (sp.k_chunks == 1) {
		h2* __restrict__ dst = grid_grad + lv.offset + lo;
		for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) { h2 o; o[0] = gb_fixed_to_half(slice64[2 * i]); o[1] = gb_fixed_to_half(slice64[2 * i + 1]); dst[i] = o; }
	} else {
		unsigned long long* __restrict__ dst = partials + (size_t)(level * GB_D_ITEMS + bin) * (2 * GB_FX_SLICE);
		for (uint32_t i = threadIdx.x; i < 2 * cnt; i += blockDim.x) dst[i] = slice64[i];
	}
}

// ... and the owner of (slice, sample chunk group) of a dense level whose bins hold pair records (gb_bin_dense_pairs)
__device__ __forceinline__ void gb_dense_pairs_owner(unsigned long long* __restrict__ slice64, const NgpGridLevel& lv, uint32_t level, uint32_t bin, uint32_t n,
                                                     const GbFxCounters* __restrict__ ctr, const ulonglong2* __restrict__ sums,
                                                     unsigned long long* __restrict__ partials, h2* __restrict__ grid_grad, uint32_t* __restrict__ s_start) {
	const GbSplit sp = gb_dense_split(lv.size);
	if (bin >= sp.n_slices * sp.k_chunks) return;
	const uint32_t sl = bin % sp.n_slices;
	const uint32_t lo = sl * GB_FX_SLICE;
	const uint32_t cnt = (lv.size - lo) < GB_FX_SLICE ? (lv.size - lo) : GB_FX_SLICE;
	const uint32_t count = ctr->totals[level][bin];
	for (uint32_t i = threadIdx.x; i < 2 * cnt; i += blockDim.x) slice64[i] = 0ull;
	__syncthreads();
	const GbRecord* __restrict__ my = (const GbRecord*)sums + gb_owner_start(ctr, level, bin, s_start);
	constexpr uint32_t UN = 8;
	for (uint32_t i0 = threadIdx.x; i0 < count; i0 += blockDim.x * UN) {
		GbRecord r[UN];
#pragma unroll
		for (uint32_t u = 0; u < UN; ++u) { const uint32_t i = i0 + u * blockDim.x; r[u] = my[i < count ? i : 0]; }
#pragma unroll
		for (uint32_t u = 0; u < UN; ++u) {
			if (i0 + u * blockDim.x >= count) continue;
#pragma unroll
			for (uint32_t xb = 0; xb < 2; ++xb) {
				const long long v0 = gb_half_fixed(r[u].t[2 * xb]), v1 = gb_half_fixed(r[u].t[2 * xb + 1]);
				const uint32_t e = (r[u].entries >> (12u * xb)) & (GB_FX_SLICE - 1);
				if (v0) atomicAdd(&slice64[2 * e], (unsigned long long)v0);
				if (v1) atomicAdd(&slice64[2 * e + 1], (unsigned long long)v1);
			}
		}
	}
	__syncthreads();
	if (sp.k_chunks == 1) {
		h2* __restrict__ dst = grid_grad + lv.offset + lo;
		for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) { h2 o; o[0] = gb_fixed_to_half(slice64[2 * i]); o[1] = gb_fixed_to_half(slice64[2 * i + 1]); dst[i] = o; }
	} else {
		unsigned long long* __restrict__ dst = partials + (size_t)(level * GB_D_ITEMS + bin) * (2 * GB_FX_SLICE);
		for (uint32_t i = threadIdx.x; i < 2 * cnt; i += blockDim.x) dst[i] = slice64[i];
	}
}

// grid (GB_FX_MAX_SLICES, 16 levels), block 1024, 64 KiB of LDS (two workgroups per CU hide each other's load -> atomics -> store chain).  dx planes: [level][sample] half2.  One launch serves the three kinds of level so
// that their workgroups overlap: binned fixed-point owners (hashed, power-of-two tables), dense owners, and the float fallback for hashed
// levels whose resolution exceeds the slice (the x term then reaches the slice bits).
// partials (GB_PARTIAL_LEVEL_BYTES per level): dense [bin][GB_FX_SLICE][2] fixed point, fallback [item][GB_SLICE] half2.
// The owners' launch, compacted: a (level, slice) grid of 16 x 256 workgroups holds ~1500 that own something for base.json (a hashed level has 128 slices, a dense one
// at most 16 bins ...) and 2600 that read the level descriptor and leave — each of them 16 waves and 64 KiB of LDS for a microsecond, with two such workgroups per CU.
// When the host knows the level table it launches only the owners: workgroup b belongs to the level whose [start, next start) holds b.
struct GbOwnerMap { uint32_t start[17]; uint32_t compact; };
template <int D>
__host__ __device__ __forceinline__ uint32_t gb_level_owners(const NgpGridLevel& lv) {
	const bool dense = level_is_dense<D>(lv);
	if (gb_uses_fx(lv.size, lv.resolution, dense)) return lv.size / GB_FX_SLICE;
	if (dense && gb_dense_binned(lv.size)) { const GbSplit sp = gb_dense_split(lv.size); return sp.n_slices * sp.k_chunks; }
	const GbSplit sp = gb_split(lv.size);
	return sp.n_slices * sp.k_chunks;
}

template <int D, bool ORDERED = true>   // D = 3: NeRF / SDF; D = 2: image fitting (4 corners, no z term); ORDERED: as gb_fx_bin_kernel
__global__ void __launch_bounds__(1024, 8) grid_backward_kernel(const NgpNetDesc* __restrict__ desc, const float* __restrict__ coords, uint32_t coord_stride, uint32_t n,
                                                             const h2* __restrict__ dx_planes, void* __restrict__ partials_raw,
                                                             const GbFxCounters* __restrict__ ctr, void* __restrict__ records, const GbRecordOffsets rec_off, h2* __restrict__ grid_grad, uint32_t level_mask,
                                                             const GbOwnerMap map) {
	NGP_RAISE_CHAIN_PRIORITY();
	constexpr int NC = 1 << D;
	__shared__ unsigned long long slice64[2 * GB_FX_SLICE];
	__shared__ uint32_t s_start;
	uint32_t level = blockIdx.y, item = blockIdx.x;
	if (map.compact) {
		level = 0;
#pragma unroll
		for (int l = 1; l < 16; ++l) level += blockIdx.x >= map.start[l] ? 1u : 0u;
		item = blockIdx.x - map.start[level];
	}
	if (!((level_mask >> level) & 1u)) return;   // dev-only ablation (tools/gb_level_probe.py); all ones in production
	const NgpGridLevel lv = desc->levels[level];
	const bool dense = level_is_dense<D>(lv);
	if (gb_uses_fx(lv.size, lv.resolution, dense)) { gb_fx_accumulate<D>(slice64, lv, level, item, n, ctr, gb_level_records(rec_off, level, n, records, false).sums, grid_grad, &s_start); return; }
	if (dense && gb_dense_binned(lv.size)) {
		const GbLevelRecords rec = gb_level_records(rec_off, level, n, records, ORDERED);
		if (ORDERED) gb_dense_owner(slice64, lv, level, item, n, ctr, rec.items, rec.sums, (unsigned long long*)partials_raw, grid_grad, &s_start);
		else gb_dense_pairs_owner(slice64, lv, level, item, n, ctr, rec.sums, (unsigned long long*)partials_raw, grid_grad, &s_start);
		return;
	}
	h2* __restrict__ slice = (h2*)slice64;
	h2* __restrict__ partials = (h2*)partials_raw;
	const GbSplit sp = gb_split(lv.size);
	if (item >= sp.n_slices * sp.k_chunks) return;
	const uint32_t sl = item % sp.n_slices, chunk = item / sp.n_slices;
	const uint32_t lo = sl * GB_SLICE;
	const uint32_t cnt = (lv.size - lo) < GB_SLICE ? (lv.size - lo) : GB_SLICE;
	const h2 zero2 = {(half_t)0.0f, (half_t)0.0f};
	for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) slice[i] = zero2;
	__syncthreads();
	const uint32_t s_begin = (uint32_t)(((uint64_t)n * chunk) / sp.k_chunks), s_end = (uint32_t)(((uint64_t)n * (chunk + 1)) / sp.k_chunks);
	const h2* __restrict__ dxl = dx_planes + (size_t)level * n;
	{
		// Every slice owner scans its chunk of samples; only ~1/n_slices of the corners land in its slice.  The scan is a chain of
		// dependent global loads, so every thread first fetches GB_UNROLL samples (coalesced across lanes).  The level size is a power
		// of two here, so a corner index is (x ^ y*P1 ^ z*P2) & (size-1) with the three products shared by the 8 corners.
		constexpr uint32_t GB_UNROLL = 8;
		const bool pow2 = (lv.size & (lv.size - 1)) == 0;
		const uint32_t hmask = lv.size - 1;
		for (uint32_t s0 = s_begin + threadIdx.x; s0 < s_end; s0 += blockDim.x * GB_UNROLL) {
			h2 gq[GB_UNROLL];
			float px[GB_UNROLL], py[GB_UNROLL], pz[GB_UNROLL];
#pragma unroll
			for (uint32_t u = 0; u < GB_UNROLL; ++u) {
				const uint32_t s = s0 + u * blockDim.x;
				const uint32_t sc = s < s_end ? s : s_begin;   // clamp: the load is always in range, the result is masked below
				gq[u] = dxl[sc];
				const float* c = coords + (size_t)sc * coord_stride;
				if (D == 3) { const f3_t v = load_pos3(c); px[u] = v.x; py[u] = v.y; pz[u] = v.z; } else { px[u] = c[0]; py[u] = c[1]; pz[u] = 0.f; }
			}
#pragma unroll
			for (uint32_t u = 0; u < GB_UNROLL; ++u) {
				const uint32_t s = s0 + u * blockDim.x;
				const float g0 = (float)gq[u][0], g1 = (float)gq[u][1];
				if (s >= s_end || (g0 == 0.0f && g1 == 0.0f)) continue;  // adding +-0 never changes a sum
				const LevelPos p = level_pos(lv, px[u], py[u], pz[u]);
				const uint32_t hx[2] = {p.gx, p.gx + 1u};
				const uint32_t hy[2] = {p.gy * 2654435761u, (p.gy + 1u) * 2654435761u};
				const uint32_t hz[2] = {D == 3 ? p.gz * 805459861u : 0u, D == 3 ? (p.gz + 1u) * 805459861u : 0u};
#pragma unroll
				for (int k = 0; k < NC; ++k) {
					const uint32_t h = hx[k & 1] ^ hy[(k >> 1) & 1] ^ hz[(k >> 2) & 1];
					const uint32_t rel = (dense ? grid_index_nd<D>(lv, p.gx + (k & 1), p.gy + ((k >> 1) & 1), p.gz + ((k >> 2) & 1)) : pow2 ? (h & hmask) : (h % lv.size)) - lo;
					if (rel < cnt) {
						float w = (k & 1) ? p.fx : (1.0f - p.fx);
						w *= ((k >> 1) & 1) ? p.fy : (1.0f - p.fy);
						if (D == 3) w *= ((k >> 2) & 1) ? p.fz : (1.0f - p.fz);
						float t0 = w * g0, t1 = w * g1;
						asm("" : "+v"(t0), "+v"(t1));                     // fp32 product first, like tcnn's half2(w * dL/dx)
						h2 val; val[0] = (half_t)t0; val[1] = (half_t)t1;
						__builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)&slice[rel], val);
					}
				}
			}
		}
	}
	__syncthreads();
	// one owner per slice (tables above 2^20 entries: more slices than private-copy slots): the table itself; else a private copy for grid_combine_kernel
	h2* __restrict__ dst = sp.n_slices > GB_ITEMS ? grid_grad + lv.offset + lo : partials + (size_t)level * (GB_PARTIAL_LEVEL_BYTES / 4u) + (size_t)item * GB_SLICE;
	for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) dst[i] = slice[i];
}

// sums the K chunk-copies of every entry in fp32 and writes the fp16 gradient table (each entry exactly once)
__global__ void __launch_bounds__(256) grid_combine_kernel(const NgpNetDesc* __restrict__ desc, const void* __restrict__ partials_raw, h2* __restrict__ grid_grad, uint32_t dims) {
	NGP_RAISE_CHAIN_PRIORITY();
	const uint32_t level = blockIdx.y;
	const NgpGridLevel lv = desc->levels[level];
	const bool dense = (dims == 3 ? (uint64_t)lv.resolution * lv.resolution * lv.resolution : (uint64_t)lv.resolution * lv.resolution) <= (uint64_t)lv.size;
	if (gb_uses_fx(lv.size, lv.resolution, dense)) return;   // written directly by the owners
	if (dense && gb_dense_binned(lv.size)) {
		const GbSplit sp = gb_dense_split(lv.size);
		if (sp.k_chunks == 1) return;                        // likewise
		const ulonglong2* __restrict__ partials = (const ulonglong2*)partials_raw;
		for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < lv.size; e += gridDim.x * blockDim.x) {
			const uint32_t sl = e / GB_FX_SLICE, rel = e % GB_FX_SLICE;
			unsigned long long a0 = 0ull, a1 = 0ull;   // two's-complement sums of the K private copies: still exact
			for (uint32_t c = 0; c < sp.k_chunks; ++c) {
				const ulonglong2 v = partials[(size_t)(level * GB_D_ITEMS + c * sp.n_slices + sl) * GB_FX_SLICE + rel];
				a0 += v.x; a1 += v.y;
			}
			h2 o; o[0] = gb_fixed_to_half(a0); o[1] = gb_fixed_to_half(a1);
			grid_grad[lv.offset + e] = o;
		}
		return;
	}
	const h2* __restrict__ partials = (const h2*)partials_raw;
	const GbSplit sp = gb_split(lv.size);
	if (sp.n_slices > GB_ITEMS) return;                      // likewise
	for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < lv.size; e += gridDim.x * blockDim.x) {
		const uint32_t sl = e / GB_SLICE, rel = e % GB_SLICE;
		float a0 = 0.0f, a1 = 0.0f;
		for (uint32_t c = 0; c < sp.k_chunks; ++c) {
			const h2 v = partials[(size_t)level * (GB_PARTIAL_LEVEL_BYTES / 4u) + (size_t)(c * sp.n_slices + sl) * GB_SLICE + rel];
			a0 += (float)v[0]; a1 += (float)v[1];
		}
		h2 o; o[0] = (half_t)a0; o[1] = (half_t)a1;
		grid_grad[lv.offset + e] = o;
	}
}

#pragma clang fp contract(fast)

// ----------------------------------------------------------------------------------------------------------------
// Backward: recompute the MLPs from the saved encoding, dgrad chain (dL/dx planes for the hash-grid backward) PLUS the weight-gradient
// contraction in the same kernel, so that no [480][n] activation / delta planes (0.25 GB written and read back per step) exist.  The contraction dW[o][i] = sum_s dY[o][s] * H[i][s] runs over SAMPLES, i.e. the MFMA
// operands must hold 8 consecutive samples of one feature per lane, while the chain keeps 8 features of one sample per lane: a transpose.
// It goes through LDS, one layer at a time: the 4 waves of the workgroup write the two operand matrices of the layer for their 4 x 32 samples
// as [feature][128 samples] fp16 rows (2-byte scattered writes, 272-byte row pitch), and read them back as 16-byte K-blocks that ARE the A / B
// operands (conflict-free: pitch = 4 banks mod 64).  The 12 output tiles of the five weight matrices are split over the 4 waves, three each
// (48 accumulator registers per wave, kept across the grid-stride loop); one fp32 partial per workgroup at the end, summed by
// wgrad_reduce_kernel.
constexpr int FB_PITCH = 128 * 2 + 16;       // bytes per staged row: 128 samples fp16 + 16 B
constexpr int FB_STAGE_ROWS = 128;
constexpr int FB_STAGE_BYTES = FB_STAGE_ROWS * FB_PITCH;   // 34 816 B

// (Measured in round 4 and dropped: lanes j and j ^ 1 swapping half of every feature pair through DPP so that each writes one dword — two samples of one feature — instead of two
// halves, i.e. 4 ds_write_b32 per K-block and lane instead of 8 ds_write_b16: the fused backward kernel went 63.6 -> 83.1 us.  The kernel is bound by VALU issue and
// dependent-instruction latency at 2 waves per SIMD, not by its LDS writes; the DPP moves and the 16-bit merges cost more than the halved writes save.)
__device__ __forceinline__ void fb_put(char* __restrict__ stage, int row0, int map, int kb, int g, int col, const h8& v) {
#pragma unroll
	for (int e = 0; e < 8; ++e) *(half_t*)(stage + (row0 + slot_feature(map, kb, g, e)) * FB_PITCH + col * 2) = v[e];
}
__device__ __forceinline__ h8 fb_get(const char* __restrict__ stage, int row, int kbs, int g) { return *(const h8*)(stage + row * FB_PITCH + (kbs * 16 + 8 * g) * 2); }

// job with a 16-row dY (padded to one 32-row M tile) and a 64-row H: N tile nt, all 8 K-blocks
__device__ __forceinline__ void fb_job_m1n2(const char* __restrict__ stage, int dy_row0, int h_row0, int nt, int lane, f32x16& acc) {
	const int r32 = lane & 31, g = lane >> 5;
#pragma unroll
	for (int kbs = 0; kbs < 8; ++kbs) {
		// rows 16 .. 31 of the padded M tile repeat rows 0 .. 15: their D rows are never read (fb_flush / gm_flush keep o < 16), and a row of D depends on its own row of A only
		const h8 a = fb_get(stage, dy_row0 + (r32 & 15), kbs, g);
		const h8 b = fb_get(stage, h_row0 + nt * 32 + r32, kbs, g);
		acc = NGP_MFMA(a, b, acc);
	}
}
// job with a 64-row dY and a 32-row H: M tile mt, all 8 K-blocks
__device__ __forceinline__ void fb_job_m2n1(const char* __restrict__ stage, int dy_row0, int h_row0, int mt, int lane, f32x16& acc) {
	const int r32 = lane & 31, g = lane >> 5;
#pragma unroll
	for (int kbs = 0; kbs < 8; ++kbs) {
		const h8 a = fb_get(stage, dy_row0 + mt * 32 + r32, kbs, g);
		const h8 b = fb_get(stage, h_row0 + r32, kbs, g);
		acc = NGP_MFMA(a, b, acc);
	}
}
// the 64 x 64 job: wave w takes tile (w >> 1, w & 1)
__device__ __forceinline__ void fb_job_m2n2(const char* __restrict__ stage, int dy_row0, int h_row0, int w, int lane, f32x16& acc) {
	const int r32 = lane & 31, g = lane >> 5, mt = w >> 1, nt = w & 1;
#pragma unroll
	for (int kbs = 0; kbs < 8; ++kbs) {
		const h8 a = fb_get(stage, dy_row0 + mt * 32 + r32, kbs, g);
		const h8 b = fb_get(stage, h_row0 + nt * 32 + r32, kbs, g);
		acc = NGP_MFMA(a, b, acc);
	}
}

// epilogue: the 4 waves' accumulator tiles of one kind -> LDS -> this workgroup's partial dW.
// kind 0: W4, wave t holds tile (t >> 1, t & 1).  kind 1: waves 0, 1 hold the two N tiles of W5, waves 2, 3 those of W2 (16 x 64 each).
// kind 2: waves 0, 1 hold the two M tiles of W3, waves 2, 3 those of W1 (64 x 32 each).
__device__ __forceinline__ void fb_flush(float* __restrict__ red /* [4][16][64] */, const f32x16& acc, int w, int lane, int kind, float* __restrict__ dst /* this workgroup's [10240] */) {
#pragma unroll
	for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
	__syncthreads();
	// D[row = o in tile][col = i in tile]: lane = (col, g), register r -> row (r & 3) + 8 (r >> 2) + 4 g
	for (int idx = threadIdx.x; idx < 4 * 16 * 64; idx += 256) {
		const int t = idx >> 10, r = (idx >> 6) & 15, l = idx & 63;
		const float v = red[idx];
		const int row_in_tile = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col_in_tile = l & 31;
		int o, i, n_out, n_in, off;
		if (kind == 0)      { o = (t >> 1) * 32 + row_in_tile; i = (t & 1) * 32 + col_in_tile; n_out = 64; n_in = 64; off = (int)W4_OFF; }
		else if (kind == 1) { o = row_in_tile; i = (t & 1) * 32 + col_in_tile; n_out = 16; n_in = 64; off = t < 2 ? (int)W5_OFF : (int)W2_OFF; }
		else                { o = (t & 1) * 32 + row_in_tile; i = col_in_tile; n_out = 64; n_in = 32; off = t < 2 ? (int)W3_OFF : (int)W1_OFF; }
		if (o < n_out && i < n_in) dst[off + o * n_in + i] = v;
	}
	__syncthreads();
}

// SH degree 4 backward to the direction for the 8 coefficients lane group g holds after the W3^T product (rows 16 + {0-3, 8-11} for g = 0,
// 16 + {4-7, 12-15} for g = 1): sum_k g_k * d(SH_k)/d(x, y, z) over those 8, with x = 2 d - 1 ([tcnn] SphericalHarmonics backward to the input)
__device__ __forceinline__ v3 sh4_grad_half(int g, float dx_, float dy_, float dz_, const float* __restrict__ gk /* 8 */) {
	const float x = dx_ * 2.0f - 1.0f, y = dy_ * 2.0f - 1.0f, z = dz_ * 2.0f - 1.0f;
	const float x2 = x * x, y2 = y * y, z2 = z * z;
	const float A = 0.48860251190291987f, B = 1.0925484305920792f, C = 0.94617469575755997f, E = 0.54627421529603959f, F = 0.59004358992664352f,
	            G = 2.8906114426405538f, Hh = 0.45704579946446572f, K = 0.3731763325901154f, M = 1.4453057213202769f;
	float ax = 0.0f, ay = 0.0f, az = 0.0f;
	if (g == 0) {   // coefficients 0, 1, 2, 3, 8, 9, 10, 11
		ay += gk[1] * -A;
		az += gk[2] * A;
		ax += gk[3] * -A;
		ax += gk[4] * (2.0f * E * x);      ay += gk[4] * (-2.0f * E * y);
		ax += gk[5] * (-6.0f * F * x * y); ay += gk[5] * (F * (-3.0f * x2 + 3.0f * y2));
		ax += gk[6] * (G * y * z);         ay += gk[6] * (G * x * z);          az += gk[6] * (G * x * y);
		ay += gk[7] * (Hh * (1.0f - 5.0f * z2)); az += gk[7] * (-10.0f * Hh * y * z);
	} else {        // coefficients 4, 5, 6, 7, 12, 13, 14, 15
		ax += gk[0] * (B * y);             ay += gk[0] * (B * x);
		ay += gk[1] * (-B * z);            az += gk[1] * (-B * y);
		az += gk[2] * (2.0f * C * z);
		ax += gk[3] * (-B * z);            az += gk[3] * (-B * x);
		az += gk[4] * (K * (15.0f * z2 - 3.0f));
		ax += gk[5] * (Hh * (1.0f - 5.0f * z2)); az += gk[5] * (-10.0f * Hh * x * z);
		ax += gk[6] * (2.0f * M * x * z);  ay += gk[6] * (-2.0f * M * y * z);  az += gk[6] * (M * (x2 - y2));
		ax += gk[7] * (F * (-3.0f * x2 + 3.0f * y2)); ay += gk[7] * (6.0f * F * x * y);
	}
	return mk(ax, ay, az);
}

// a batch whose first *n_live slots carry the samples with a non-zero loss gradient (ngp_hip_compact_live_samples); zero_next: a word the kernel clears (the counter of the next step)
struct LiveSamples { const uint32_t* n_live; uint32_t* zero_next; const uint32_t* src_index /* slot -> row of x_saved / dL_dout / coords (NULL: the slot itself) */;
                     const uint32_t* x_row /* row of coords / dL_dout -> row of x_saved (NULL: the same row): the loss kernel's x_row_index_out */; };

// DIR_GRAD: additionally dL/d(direction) of every sample into dL_dinput[s][3..5] (fp32) — the input gradient a training step asks for when
// camera parameters train (NerfNetwork::backward_impl with dL_dinput, nerf_network.h:187-266; testbed_nerf.cu:3324-3346)
template <bool DIR_GRAD>
__global__ void __launch_bounds__(256, 2) nerf_backward_fused_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params,
                                                                  const float* __restrict__ coords, uint32_t coord_stride, uint32_t n,
                                                                  const half_t* __restrict__ x_saved, const half_t* __restrict__ dL_dout, uint32_t dl_stride,
                                                                  h2* __restrict__ dx_planes, float* __restrict__ partials /* [gridDim.x][10240] */, uint32_t* __restrict__ zero_words, uint32_t n_zero_words,
                                                                  float* __restrict__ dL_dinput /* [n][6] or NULL */, const LiveSamples live) {
	NGP_RAISE_CHAIN_PRIORITY();
	__shared__ __attribute__((aligned(16))) h8 lds_tiles[N_ALL_TILES * 64];
	__shared__ __attribute__((aligned(16))) char stage[FB_STAGE_BYTES];
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_zero_words; k += gridDim.x * blockDim.x) zero_words[k] = 0u;   // the counters of the hash-grid backward that follows are cleared here instead of by a memset launch of their own
	stage_tiles<GRID_OFF, N_ALL_TILES>(lds_tiles, params, gather_tile);

	const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5, w = threadIdx.x >> 6;
	// live.n_live (ngp_hip_nerf_backward_live): only the first *n_live samples of the batch carry a gradient — the rest of the n slots is not read, its dL/dx is written as zeros
	uint32_t n_eff = n;
	if (live.n_live) { const uint32_t v = *live.n_live; n_eff = v < n ? v : n; }
	if (live.zero_next && blockIdx.x == 0 && threadIdx.x == 0) *live.zero_next = 0u;   // the other step parity's counter: its last reader (the previous step's launch of this kernel) is done
	const uint32_t n_quads = (n_eff + 127u) / 128u;   // 4 tiles of 32 samples per workgroup iteration (n % 256 == 0)
	const f32x16 zero = {};
	// 12 output tiles over 4 waves, each over all 128 samples of an iteration: W4 one tile per wave; waves 0 / 1 additionally the two tiles of W5
	// and of W3, waves 2 / 3 those of W2 and of W1 (24 MFMAs per wave and iteration either way) — 48 accumulator registers per wave
	f32x16 acc_w4 = zero, acc_a = zero /* W5 | W2 */, acc_b = zero /* W3 | W1 */;
	const int col = w * 32 + j;
	const bool low_pair = __builtin_amdgcn_readfirstlane(w) < 2;

	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const uint32_t s = (quad * 4 + w) * 32 + j;
		const bool on = s < n_eff;               // (the last quad of a live batch: slots behind the live samples hold nothing — zero encoding, zero gradient)
		const uint32_t sl = on ? (live.src_index ? live.src_index[s] : s) : 0u;
		const float* c = coords + (size_t)sl * coord_stride;
		const uint32_t xrow = live.x_row ? live.x_row[sl] : sl;   // (the compaction left an index instead of a copy of the row)
		const h8* xs = (const h8*)(x_saved + (size_t)xrow * 32 + 16 * g);
		h8 x0 = xs[0], x1 = xs[1];
		if (!on) { x0 = h8{}; x1 = h8{}; }
		const h8 sh = sh4_half(g, c[4], c[5], c[6]);
		FwdActs a;
		f32x16 dd, oo;
		uint32_t lt_off = 0;
		asm volatile("" : "+s"(lt_off)); // keep the LDS weight reads inside the loop
		const h8* lt = lds_tiles + lt_off;
		mlp_forward<false, true>(lt, lane, x0, x1, sh, dd, oo, &a);

		const half_t* dl = dL_dout + (size_t)sl * dl_stride;
		h8 dout = {};
		if (g == 0 && on) { dout[0] = dl[0]; dout[1] = dl[1]; dout[2] = dl[2]; }
		const half_t dsigma = on ? dl[3] : (half_t)0.0f;

		// ---- W5: dY = dout (16 rows), H = h3
		fb_put(stage, 0, MAP_CH, 0, g, col, dout);
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 16, MAP_HID, kb, g, col, a.h3[kb]);
		__syncthreads();
		if (low_pair) fb_job_m1n2(stage, 0, 16, w, lane, acc_a);
		// d_h3 = relu'(h3) * (W5^T dout)
		f32x16 t0 = NGP_MFMA(lt[(T_W5T + 0) * 64 + lane], dout, zero);
		f32x16 t1 = NGP_MFMA(lt[(T_W5T + 1) * 64 + lane], dout, zero);
		h8 dh[4];
		dh[0] = mask_delta(t0, 0, a.h3[0]); dh[1] = mask_delta(t0, 1, a.h3[1]);
		dh[2] = mask_delta(t1, 0, a.h3[2]); dh[3] = mask_delta(t1, 1, a.h3[3]);
		__syncthreads();

		// ---- W4: dY = d_h3, H = h2
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) { fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]); fb_put(stage, 64, MAP_HID, kb, g, col, a.h2[kb]); }
		__syncthreads();
		fb_job_m2n2(stage, 0, 64, w, lane, acc_w4);
		// d_h2 = relu'(h2) * (W4^T d_h3)
		t0 = zero; t1 = zero;
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) {
			t0 = NGP_MFMA(lt[(T_W4T + kb) * 64 + lane], dh[kb], t0);
			t1 = NGP_MFMA(lt[(T_W4T + 4 + kb) * 64 + lane], dh[kb], t1);
		}
		dh[0] = mask_delta(t0, 0, a.h2[0]); dh[1] = mask_delta(t0, 1, a.h2[1]);
		dh[2] = mask_delta(t1, 0, a.h2[2]); dh[3] = mask_delta(t1, 1, a.h2[3]);
		__syncthreads();

		// ---- W3: dY = d_h2, H = rgb-net input [density out | SH]
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]);
		fb_put(stage, 64, MAP_RGBIN, 0, g, col, a.rin[0]);
		fb_put(stage, 64, MAP_RGBIN, 1, g, col, a.rin[1]);
		__syncthreads();
		if (low_pair) fb_job_m2n1(stage, 0, 64, w, lane, acc_b);
		// d_in = W3^T d_h2 (rows 0..15 = density-net output gradient)
		t0 = zero;
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) t0 = NGP_MFMA(lt[(T_W3T + kb) * 64 + lane], dh[kb], t0);
		h8 dden;
#pragma unroll
		for (int e = 0; e < 8; e += 2) { const h2 v = cvt_pk_f16(t0[e], t0[e + 1]); dden[e] = v[0]; dden[e + 1] = v[1]; }
		if (g == 0) dden[0] = (half_t)((float)dden[0] + (float)dsigma); // add_density_gradient (nerf_network.h:63-74): fp16 += fp16
		if (DIR_GRAD) {
			// rows 16..31 of d_in are dL/d(SH coefficients), fp16 like the rgb network's dL_dinput matrix
			float gk[8];
#pragma unroll
			for (int e = 0; e < 8; ++e) gk[e] = (float)(half_t)t0[8 + e];
			v3 dd3 = sh4_grad_half(g, c[4], c[5], c[6], gk);
			dd3.x += __shfl_xor(dd3.x, 32, 64); dd3.y += __shfl_xor(dd3.y, 32, 64); dd3.z += __shfl_xor(dd3.z, 32, 64);
			if (g == 0) { float* o = dL_dinput + (size_t)s * 6; o[3] = 2.0f * dd3.x; o[4] = 2.0f * dd3.y; o[5] = 2.0f * dd3.z; }
		}
		__syncthreads();

		// ---- W2: dY = d_dens (16 rows), H = h1
		fb_put(stage, 0, MAP_RGBIN, 0, g, col, dden);
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 16, MAP_HID, kb, g, col, a.h1[kb]);
		__syncthreads();
		if (!low_pair) fb_job_m1n2(stage, 0, 16, w - 2, lane, acc_a);
		// d_h1 = relu'(h1) * (W2^T d_dens)
		t0 = NGP_MFMA(lt[(T_W2T + 0) * 64 + lane], dden, zero);
		t1 = NGP_MFMA(lt[(T_W2T + 1) * 64 + lane], dden, zero);
		dh[0] = mask_delta(t0, 0, a.h1[0]); dh[1] = mask_delta(t0, 1, a.h1[1]);
		dh[2] = mask_delta(t1, 0, a.h1[2]); dh[3] = mask_delta(t1, 1, a.h1[3]);
		__syncthreads();

		// ---- W1: dY = d_h1, H = x (the saved encoding)
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]);
		{   // the encoding again (64 B per sample, an L2 hit): keeping x0 / x1 live through the chain costs 8 registers of a kernel that is at its limit
			const h8* xr = (const h8*)(x_saved + (size_t)xrow * 32 + 16 * g);
			uint32_t zero_off = 0;
			asm volatile("" : "+v"(zero_off));
			xr = (const h8*)((const char*)xr + zero_off);
			h8 xa = xr[0], xb = xr[1];
			if (!on) { xa = h8{}; xb = h8{}; }
			fb_put(stage, 64, MAP_ENC, 0, g, col, xa);
			fb_put(stage, 64, MAP_ENC, 1, g, col, xb);
		}
		__syncthreads();
		if (!low_pair) fb_job_m2n1(stage, 0, 64, w - 2, lane, acc_b);
		// d_x = W1^T d_h1
		t0 = zero;
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) t0 = NGP_MFMA(lt[(T_W1T + kb) * 64 + lane], dh[kb], t0);
		// t0 row = x feature (r&3)+8(r>>2)+4g  =>  this lane owns levels 4q+2g (regs 4q,4q+1) and 4q+2g+1 (regs 4q+2,4q+3), q = 0..3.
		// dL/dx goes out in fp16 as per-level planes [level][sample] (coalesced 128 B per half-wave) for grid_backward_kernel.
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int lvl = 4 * q + 2 * g;
			h2 u, v;
			u[0] = (half_t)t0[4 * q + 0]; u[1] = (half_t)t0[4 * q + 1];
			v[0] = (half_t)t0[4 * q + 2]; v[1] = (half_t)t0[4 * q + 3];
			dx_planes[(size_t)lvl * n + s] = u;
			dx_planes[(size_t)(lvl + 1) * n + s] = v;
		}
		__syncthreads();
	}
	if (live.n_live) {   // dL/dx of the slots behind the live samples: zeros (the hash-grid backward skips zero pairs; its binning skips whole chunks of them)
		const uint32_t first = n_quads * 128u, tail = n - first;
		const h2 z2 = {(half_t)0.0f, (half_t)0.0f};
		for (uint32_t l = 0; l < 16u; ++l)
			for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tail; i += gridDim.x * blockDim.x) dx_planes[(size_t)l * n + first + i] = z2;
	}
	// ---- this workgroup's partial weight gradients
	float* __restrict__ dst = partials + (size_t)blockIdx.x * NGP_MLP_N_PARAMS;
	float* red = (float*)stage;   // 4 x 16 x 64 floats = 16 KiB
	fb_flush(red, acc_w4, w, lane, 0, dst);
	fb_flush(red, acc_a, w, lane, 1, dst);
	fb_flush(red, acc_b, w, lane, 2, dst);
}

// dL/d(position) through the hash encoding ([tcnn] GridEncoding: dy/dx of the trilinear interpolation — scale * prod over the other two
// dimensions of (1 - w | w), times (value at the +1 corner - value at the 0 corner) — summed with dL/dy over levels and features in fp32, in
// the order levels, dimension, corner pair, feature; oracle: orc_nerf_input_gradient).  One thread per sample; dL/dy = the dL/dx planes the
// backward kernel left.  Only run when camera parameters train.
#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) nerf_input_pos_gradient_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ coords,
                                                                      uint32_t coord_stride, uint32_t n, const h2* __restrict__ dx_planes, float* __restrict__ dL_dinput, uint32_t grid_off) {
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	const float* c = coords + (size_t)s * coord_stride;
	const f3_t pv = load_pos3(c);
	const h2* __restrict__ grid = (const h2*)(params + grid_off);
	float gp[3] = {0.0f, 0.0f, 0.0f};
	for (int l = 0; l < 16; ++l) {
		const NgpGridLevel lv = desc->levels[l];
		const LevelPos p = level_pos(lv, pv.x, pv.y, pv.z);
		const uint32_t pg[3] = {p.gx, p.gy, p.gz};
		const float w[3] = {p.fx, p.fy, p.fz};
		h2 v[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) v[k] = grid[lv.offset + grid_index(lv, pg[0] + (k & 1), pg[1] + ((k >> 1) & 1), pg[2] + ((k >> 2) & 1))];
		const h2 gq = dx_planes[(size_t)l * n + s];
		const float g0 = (float)gq[0], g1 = (float)gq[1];
#pragma unroll
		for (int gd = 0; gd < 3; ++gd) {
			float dy0 = 0.0f, dy1 = 0.0f;
#pragma unroll
			for (int idx = 0; idx < 4; ++idx) {
				float weight = lv.scale;
				int corner = 0;
#pragma unroll
				for (int nd = 0; nd < 2; ++nd) {
					const int dim = nd >= gd ? nd + 1 : nd;
					if ((idx >> nd) & 1) { weight *= w[dim]; corner |= 1 << dim; } else weight *= 1.0f - w[dim];
				}
				const h2 lo = v[corner], hi = v[corner | (1 << gd)];
				dy0 += weight * ((float)hi[0] - (float)lo[0]);
				dy1 += weight * ((float)hi[1] - (float)lo[1]);
			}
			gp[gd] += g0 * dy0;
			gp[gd] += g1 * dy1;
		}
	}
	float* o = dL_dinput + (size_t)s * 6;
	o[0] = gp[0]; o[1] = gp[1]; o[2] = gp[2];
}
#pragma clang fp contract(fast)

// ---- visualisation passes of the renderer (not on the training path): [tcnn] Network::visualize_activation and the in-place write-back of
// DifferentiableObject::input_gradient.  One thread per sample, scalar fp32 accumulation over the row-major fp16 weights (their index is
// wave-uniform: scalar loads), activations rounded to fp16 between layers like the MFMA kernels store them.
__device__ __forceinline__ float vis_dot(const half_t* __restrict__ w, const half_t* __restrict__ x, int n) {
	float acc = 0.0f;
	for (int i = 0; i < n; ++i) acc += (float)w[i] * (float)x[i];
	return acc;
}
__global__ void __launch_bounds__(256) nerf_visualize_activation_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, uint32_t layer, uint32_t dim,
                                                                        const float* __restrict__ coords, uint32_t coord_stride, uint32_t n, float* __restrict__ out, uint32_t out_stride) {
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	const float* c = coords + (size_t)s * coord_stride;
	const f3_t pv = load_pos3(c);
	const float d0 = c[4], d1 = c[5], d2 = c[6];
	const h2* __restrict__ grid = (const h2*)(params + GRID_OFF);
	half_t x[32];
	for (int l = 0; l < 16; ++l) encode_level<false>(desc->levels[l], grid, pv.x, pv.y, pv.z, x[2 * l], x[2 * l + 1]);
	float v;
	if (layer == 0) {
		v = (float)x[dim & 31];
	} else if (layer == 1) {
		v = fmaxf((float)(half_t)vis_dot(params + W1_OFF + (dim & 63) * 32, x, 32), 0.0f);
	} else {
		half_t rin[32];
		if (layer == 2 && dim >= 16) {
			const h8 a = sh4_half(0, d0, d1, d2), b = sh4_half(1, d0, d1, d2);
			v = (float)((dim & 31) < 24 ? a[dim & 7] : b[dim & 7]);
		} else {
			half_t h[64];
			for (int o = 0; o < 64; ++o) h[o] = (half_t)fmaxf(vis_dot(params + W1_OFF + o * 32, x, 32), 0.0f);
			if (layer == 2) {
				v = (float)(half_t)vis_dot(params + W2_OFF + (dim & 15) * 64, h, 64);
			} else {
				for (int o = 0; o < 16; ++o) rin[o] = (half_t)vis_dot(params + W2_OFF + o * 64, h, 64);
				const h8 a = sh4_half(0, d0, d1, d2), b = sh4_half(1, d0, d1, d2);
				for (int o = 0; o < 8; ++o) { rin[16 + o] = a[o]; rin[24 + o] = b[o]; }
				if (layer == 3) {
					v = fmaxf((float)(half_t)vis_dot(params + W3_OFF + (dim & 63) * 32, rin, 32), 0.0f);
				} else {
					for (int o = 0; o < 64; ++o) h[o] = (half_t)fmaxf(vis_dot(params + W3_OFF + o * 32, rin, 32), 0.0f);
					v = fmaxf((float)(half_t)vis_dot(params + W4_OFF + (dim & 63) * 64, h, 64), 0.0f);
				}
			}
		}
	}
	// [tcnn] extract_dimension_pos_neg_kernel: row 0 = negative part, row 1 = positive part, row 2 = 0, further rows = 1
	float* o = out + (size_t)s * out_stride;
	if (out_stride == 1) { o[0] = v; return; }
	for (uint32_t k = 0; k < out_stride; ++k) o[k] = k == 0 ? fmaxf(-v, 0.0f) : k == 1 ? fmaxf(v, 0.0f) : k == 2 ? 0.0f : 1.0f;
}

__global__ void __launch_bounds__(256) one_hot_dl_kernel(uint32_t n, uint32_t dim, float scale, half_t* __restrict__ dL_dout /* [n][4] */) {
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	for (uint32_t k = 0; k < 4; ++k) dL_dout[(size_t)s * 4 + k] = (half_t)(k == dim ? scale : 0.0f);
}
// d_output_d_input aliases the input: position and direction rows <- gradient / scale, every other row (dt) <- old value / scale
__global__ void __launch_bounds__(256) input_gradient_writeback_kernel(uint32_t n, float inv_scale, const float* __restrict__ dL_dinput /* [n][6] */, float* __restrict__ coords, uint32_t coord_stride) {
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	float* c = coords + (size_t)s * coord_stride;
	const float* g = dL_dinput + (size_t)s * 6;
	c[0] = g[0] * inv_scale; c[1] = g[1] * inv_scale; c[2] = g[2] * inv_scale;
	c[3] = c[3] * inv_scale;
	c[4] = g[3] * inv_scale; c[5] = g[4] * inv_scale; c[6] = g[5] * inv_scale;
	for (uint32_t k = 7; k < coord_stride; ++k) c[k] = c[k] * inv_scale;
}

// ================================================================================================================
// Plumbing configs P1 / P2: ONE grid encoding (2-D or 3-D, 16 levels x 2 features) -> ONE FullyFusedMLP 32 -> 64 -> 64 -> 16
// (tcnn NetworkWithInputEncoding as built by Testbed::reset_network for Image / Sdf mode, src/testbed.cu:2397-2445; configs/image/base.json,
// configs/sdf/base.json).  Parameter order: MLP (input 64x32, hidden 64x64, output 16x64 = 7168), then the grid.  Same wave-owns-32-samples
// MFMA chaining as the NeRF kernels; the topology equals the NeRF colour network fed by the encoding instead of [density | SH].
constexpr uint32_t GM_L0_OFF = 0, GM_L1_OFF = 64 * 32, GM_L2_OFF = GM_L1_OFF + 64 * 64, GM_GRID_OFF = GM_L2_OFF + 16 * 64;   // 7168
static_assert(GM_GRID_OFF == NGP_GRIDMLP_N_PARAMS, "GridMLP parameter count");
constexpr int G_L0 = 0;    // + mt*2 + kb (4)   [64][32], K slots = encoding features (MAP_ENC)
constexpr int G_L1 = 4;    // + mt*4 + kb (8)   [64][64]
constexpr int G_L2 = 12;   // + kb        (4)   [16][64]
constexpr int GM_FWD_TILES = 16;
constexpr int G_L2T = 16;  // + mt        (2)   A[i = h2 feat][slot = channel]
constexpr int G_L1T = 18;  // + mt*4 + kb (8)   A[i = h1 feat][slot = h2 feat]
constexpr int G_L0T = 26;  // + kb        (4)   A[i = x feat][slot = h1 feat]
constexpr int GM_ALL_TILES = 30;

__device__ __forceinline__ h8 gm_gather_tile(const half_t* P, int tile, int lane) {
	if (tile < G_L1)  return gather_tile_spec(P, GM_L0_OFF, 64, 32, (tile - G_L0) >> 1, (tile - G_L0) & 1, MAP_ENC, false, lane);
	if (tile < G_L2)  return gather_tile_spec(P, GM_L1_OFF, 64, 64, (tile - G_L1) >> 2, (tile - G_L1) & 3, MAP_HID, false, lane);
	if (tile < G_L2T) return gather_tile_spec(P, GM_L2_OFF, 16, 64, 0, tile - G_L2, MAP_HID, false, lane);
	if (tile < G_L1T) return gather_tile_spec(P, GM_L2_OFF, 16, 64, tile - G_L2T, 0, MAP_CH, true, lane);
	if (tile < G_L0T) return gather_tile_spec(P, GM_L1_OFF, 64, 64, (tile - G_L1T) >> 2, (tile - G_L1T) & 3, MAP_HID, true, lane);
	return gather_tile_spec(P, GM_L0_OFF, 64, 32, 0, tile - G_L0T, MAP_HID, true, lane);
}

struct GmActs { h8 h1[4]; h8 h2[4]; };
template <bool KEEP>
__device__ __forceinline__ void gm_mlp_forward(const h8* __restrict__ lt, int lane, const h8& x0, const h8& x1, f32x16& oo, GmActs* acts) {
	const f32x16 zero = {};
	f32x16 a0 = NGP_MFMA(lt[(G_L0 + 0) * 64 + lane], x0, zero);
	a0 = NGP_MFMA(lt[(G_L0 + 1) * 64 + lane], x1, a0);
	f32x16 a1 = NGP_MFMA(lt[(G_L0 + 2) * 64 + lane], x0, zero);
	a1 = NGP_MFMA(lt[(G_L0 + 3) * 64 + lane], x1, a1);
	h8 h[4];
	d_to_b<true>(a0, h[0], h[1]);
	d_to_b<true>(a1, h[2], h[3]);
	if (KEEP) { for (int k = 0; k < 4; ++k) acts->h1[k] = h[k]; }
	a0 = zero; a1 = zero;
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) {
		a0 = NGP_MFMA(lt[(G_L1 + kb) * 64 + lane], h[kb], a0);
		a1 = NGP_MFMA(lt[(G_L1 + 4 + kb) * 64 + lane], h[kb], a1);
	}
	d_to_b<true>(a0, h[0], h[1]);
	d_to_b<true>(a1, h[2], h[3]);
	if (KEEP) { for (int k = 0; k < 4; ++k) acts->h2[k] = h[k]; }
	oo = zero;
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) oo = NGP_MFMA(lt[(G_L2 + kb) * 64 + lane], h[kb], oo);
}

// forward: out[s*out_stride + 0..3] = network outputs 0..3 (fp16; 3 used by the image config, 1 by the SDF config); x_saved optional
// PRE = 1: the MLP alone over the level planes of encode_planes_kernel<D> (ngp_hip_gridmlp_forward_ws), as nerf_forward_kernel<., 1>
template <int D, int PRE>
__global__ void __launch_bounds__(256, PRE ? 4 : 2) gridmlp_forward_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ pos, uint32_t pos_stride,
                                                                 uint32_t n, half_t* __restrict__ out, uint32_t out_stride, half_t* __restrict__ x_saved,
                                                                 const h2* __restrict__ x_planes, uint32_t n_pad) {
	__shared__ __attribute__((aligned(16))) h8 lds_tiles[GM_FWD_TILES * 64];
	stage_tiles<GM_GRID_OFF, GM_FWD_TILES>(lds_tiles, params, gm_gather_tile);
	const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
	const uint32_t n_tiles = (n + 31) / 32;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
	const h2* __restrict__ grid = (const h2*)(params + GM_GRID_OFF);
	for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
		const uint32_t s = tile * 32 + j;
		const bool valid = s < n;
		const float* c = pos + (size_t)(valid ? s : 0) * pos_stride;
		h8 x0, x1;
		if (PRE == 1) {
			const h2* xp = x_planes + (size_t)(8 * g) * n_pad + (valid ? s : 0);
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				const h2 a = xp[(size_t)m * n_pad], b = xp[(size_t)(4 + m) * n_pad];
				x0[2 * m] = a[0]; x0[2 * m + 1] = a[1];
				x1[2 * m] = b[0]; x1[2 * m + 1] = b[1];
			}
		} else {
			const float px = c[0], py = c[1], pz = D == 3 ? c[2] : 0.0f;
#pragma unroll
			for (int m = 0; m < 4; ++m) {
				half_t a, b;
				encode_level_nd<D>(desc->levels[8 * g + m], grid, px, py, pz, a, b);
				x0[2 * m] = a; x0[2 * m + 1] = b;
				encode_level_nd<D>(desc->levels[8 * g + 4 + m], grid, px, py, pz, a, b);
				x1[2 * m] = a; x1[2 * m + 1] = b;
			}
		}
		if (x_saved && valid) { h8* dst = (h8*)(x_saved + (size_t)s * 32 + 16 * g); dst[0] = x0; dst[1] = x1; }
		f32x16 oo;
		uint32_t lt_off = 0;
		asm volatile("" : "+s"(lt_off));
		gm_mlp_forward<false>(lds_tiles + lt_off, lane, x0, x1, oo, nullptr);
		if (valid && g == 0) {
			typedef _Float16 h4 __attribute__((ext_vector_type(4)));
			h4 o; o[0] = (half_t)oo[0]; o[1] = (half_t)oo[1]; o[2] = (half_t)oo[2]; o[3] = (half_t)oo[3];
			*(h4*)(out + (size_t)s * out_stride) = o;
		}
	}
}

// backward: recompute the MLP from the saved encoding, dgrad chain (channels 0..3 of dL_dout) and the weight-gradient contraction in the SAME kernel, operands
// transposed through LDS one layer at a time — the scheme of nerf_backward_fused_kernel with one network instead of two (the [304][n] fp16 activation / delta
// planes and the second kernel that read them back are gone).  The 8 output tiles of the three matrices over the 4 waves: the 64 x 64 hidden matrix one tile per
// wave; waves 0 / 1 the two N tiles of the 16 x 64 output matrix, waves 2 / 3 the two M tiles of the 64 x 32 input matrix: 32 accumulator registers per wave.
__device__ __forceinline__ void gm_flush(float* __restrict__ red /* [4][16][64] */, const f32x16& acc, int w, int lane, int kind, float* __restrict__ dst /* this workgroup's [7168] */) {
#pragma unroll
	for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
	__syncthreads();
	for (int idx = threadIdx.x; idx < 4 * 16 * 64; idx += 256) {
		const int t = idx >> 10, r = (idx >> 6) & 15, l = idx & 63;
		const float v = red[idx];
		const int row_in_tile = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col_in_tile = l & 31;
		int o, i, n_out, n_in, off;
		if (kind == 0)  { o = (t >> 1) * 32 + row_in_tile; i = (t & 1) * 32 + col_in_tile; n_out = 64; n_in = 64; off = (int)GM_L1_OFF; }
		else if (t < 2) { o = row_in_tile; i = (t & 1) * 32 + col_in_tile; n_out = 16; n_in = 64; off = (int)GM_L2_OFF; }
		else            { o = (t & 1) * 32 + row_in_tile; i = col_in_tile; n_out = 64; n_in = 32; off = (int)GM_L0_OFF; }
		if (o < n_out && i < n_in) dst[off + o * n_in + i] = v;
	}
	__syncthreads();
}

__global__ void __launch_bounds__(256, 2) gridmlp_backward_fused_kernel(const half_t* __restrict__ params, uint32_t n, const half_t* __restrict__ x_saved, const half_t* __restrict__ dL_dout,
                                                                        uint32_t dl_stride, h2* __restrict__ dx_planes, float* __restrict__ partials /* [gridDim.x][7168] */,
                                                                        uint32_t* __restrict__ zero_words, uint32_t n_zero_words) {
	__shared__ __attribute__((aligned(16))) h8 lds_tiles[GM_ALL_TILES * 64];
	__shared__ __attribute__((aligned(16))) char stage[FB_STAGE_BYTES];
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_zero_words; k += gridDim.x * blockDim.x) zero_words[k] = 0u;   // see nerf_backward_fused_kernel
	stage_tiles<GM_GRID_OFF, GM_ALL_TILES>(lds_tiles, params, gm_gather_tile);
	const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5, w = threadIdx.x >> 6;
	const uint32_t n_quads = n / 128;   // 4 tiles of 32 samples per workgroup iteration (n % 256 == 0)
	const f32x16 zero = {};
	f32x16 acc_l1 = zero, acc_a = zero /* waves 0, 1: output matrix | waves 2, 3: input matrix */;
	const int col = w * 32 + j;
	const bool low_pair = __builtin_amdgcn_readfirstlane(w) < 2;
	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const uint32_t s = (quad * 4 + w) * 32 + j;
		const h8* xs = (const h8*)(x_saved + (size_t)s * 32 + 16 * g);
		const h8 x0 = xs[0], x1 = xs[1];
		GmActs a;
		f32x16 oo;
		uint32_t lt_off = 0;
		asm volatile("" : "+s"(lt_off));   // keep the LDS weight reads inside the loop
		const h8* lt = lds_tiles + lt_off;
		gm_mlp_forward<true>(lt, lane, x0, x1, oo, &a);
		const half_t* dl = dL_dout + (size_t)s * dl_stride;
		h8 dout = {};
		if (g == 0) { dout[0] = dl[0]; dout[1] = dl[1]; dout[2] = dl[2]; dout[3] = dl[3]; }

		// ---- output matrix: dY = dout (16 rows), H = h2
		fb_put(stage, 0, MAP_CH, 0, g, col, dout);
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 16, MAP_HID, kb, g, col, a.h2[kb]);
		__syncthreads();
		if (low_pair) fb_job_m1n2(stage, 0, 16, w, lane, acc_a);
		// d_h2 = relu'(h2) * (L2^T dout)
		f32x16 t0 = NGP_MFMA(lt[(G_L2T + 0) * 64 + lane], dout, zero);
		f32x16 t1 = NGP_MFMA(lt[(G_L2T + 1) * 64 + lane], dout, zero);
		h8 dh[4];
		dh[0] = mask_delta(t0, 0, a.h2[0]); dh[1] = mask_delta(t0, 1, a.h2[1]);
		dh[2] = mask_delta(t1, 0, a.h2[2]); dh[3] = mask_delta(t1, 1, a.h2[3]);
		__syncthreads();

		// ---- hidden matrix: dY = d_h2, H = h1
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) { fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]); fb_put(stage, 64, MAP_HID, kb, g, col, a.h1[kb]); }
		__syncthreads();
		fb_job_m2n2(stage, 0, 64, w, lane, acc_l1);
		// d_h1 = relu'(h1) * (L1^T d_h2)
		t0 = zero; t1 = zero;
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) {
			t0 = NGP_MFMA(lt[(G_L1T + kb) * 64 + lane], dh[kb], t0);
			t1 = NGP_MFMA(lt[(G_L1T + 4 + kb) * 64 + lane], dh[kb], t1);
		}
		dh[0] = mask_delta(t0, 0, a.h1[0]); dh[1] = mask_delta(t0, 1, a.h1[1]);
		dh[2] = mask_delta(t1, 0, a.h1[2]); dh[3] = mask_delta(t1, 1, a.h1[3]);
		__syncthreads();

		// ---- input matrix: dY = d_h1, H = x (the saved encoding)
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]);
		fb_put(stage, 64, MAP_ENC, 0, g, col, x0);
		fb_put(stage, 64, MAP_ENC, 1, g, col, x1);
		__syncthreads();
		if (!low_pair) fb_job_m2n1(stage, 0, 64, w - 2, lane, acc_a);
		// d_x = L0^T d_h1
		t0 = zero;
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) t0 = NGP_MFMA(lt[(G_L0T + kb) * 64 + lane], dh[kb], t0);
#pragma unroll
		for (int q = 0; q < 4; ++q) {   // same row -> level mapping as nerf_backward_fused_kernel
			const int lvl = 4 * q + 2 * g;
			h2 u, v;
			u[0] = (half_t)t0[4 * q + 0]; u[1] = (half_t)t0[4 * q + 1];
			v[0] = (half_t)t0[4 * q + 2]; v[1] = (half_t)t0[4 * q + 3];
			dx_planes[(size_t)lvl * n + s] = u;
			dx_planes[(size_t)(lvl + 1) * n + s] = v;
		}
		__syncthreads();
	}
	float* __restrict__ dst = partials + (size_t)blockIdx.x * NGP_GRIDMLP_N_PARAMS;
	float* red = (float*)stage;   // 4 x 16 x 64 floats = 16 KiB
	gm_flush(red, acc_l1, w, lane, 0, dst);
	gm_flush(red, acc_a, w, lane, 1, dst);
}

// sums the per-chunk partial weight gradients.  64 parameters x 4 chunk groups per block; every thread keeps 8 independent loads in
// flight (the straightforward one-thread-per-parameter loop is a chain of n_chunks dependent L2 latencies: 40 us for 5 MB)
constexpr uint32_t WR_WAVES = 16;   // waves per workgroup: each sums every 16th partial of the workgroup's 64 parameters (512 partials: 32 loads per thread, 8 in flight)
__global__ void __launch_bounds__(64 * WR_WAVES) wgrad_reduce_kernel(const float* __restrict__ partials, uint32_t n_chunks, half_t* __restrict__ grads, uint32_t n_mlp_params) {
	NGP_RAISE_CHAIN_PRIORITY();
	__shared__ float red[WR_WAVES][64];
	const uint32_t p = threadIdx.x & 63u, q = threadIdx.x >> 6;
	const uint32_t i = blockIdx.x * 64u + p;
	float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
	if (i < n_mlp_params) {
		uint32_t c = q;
		for (; c + 7 * WR_WAVES < n_chunks; c += 8 * WR_WAVES) {
#pragma unroll
			for (int u = 0; u < 8; ++u) acc[u] += partials[(size_t)(c + WR_WAVES * u) * n_mlp_params + i];
		}
		for (int u = 0; c < n_chunks; c += WR_WAVES, ++u) acc[u & 7] += partials[(size_t)c * n_mlp_params + i];
	}
	red[q][p] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
	__syncthreads();
	if (q == 0 && i < n_mlp_params) {
		float sum = 0.0f;
#pragma unroll
		for (uint32_t k = 0; k < WR_WAVES; ++k) sum += red[k][p];
		grads[i] = (half_t)sum;
	}
}

// ----------------------------------------------------------------------------------------------------------------
// parameter init (nerf_network.h:396-441 order; tcnn Xavier-uniform / U(-1e-4, 1e-4)); element k <- k-th draw of pcg32(seed)
__global__ void init_params_kernel(uint32_t n_params, uint64_t seed_state, uint64_t seed_inc, float* __restrict__ master, half_t* __restrict__ params, half_t* __restrict__ inference) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_params) return;
	Pcg32 rng; rng.state = seed_state; rng.inc = seed_inc;
	rng.advance(k);
	float scale;
	if (k < W2_OFF) scale = sqrtf(6.0f / (float)(64 + 32));
	else if (k < W3_OFF) scale = sqrtf(6.0f / (float)(16 + 64));
	else if (k < W4_OFF) scale = sqrtf(6.0f / (float)(64 + 32));
	else if (k < W5_OFF) scale = sqrtf(6.0f / (float)(64 + 64));
	else if (k < GRID_OFF) scale = sqrtf(6.0f / (float)(16 + 64));
	else scale = 1e-4f;
	float v;
	{
#pragma clang fp contract(off)
		v = rng.next_float() * (scale - (-scale)) + (-scale);
	}
	master[k] = v;
	params[k] = (half_t)v;
	inference[k] = (half_t)v;
}

__global__ void gridmlp_init_params_kernel(uint32_t n_params, uint64_t seed_state, uint64_t seed_inc, float* __restrict__ master, half_t* __restrict__ params, half_t* __restrict__ inference) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_params) return;
	Pcg32 rng; rng.state = seed_state; rng.inc = seed_inc;
	rng.advance(k);
	float scale;
	if (k < GM_L1_OFF) scale = sqrtf(6.0f / (float)(64 + 32));
	else if (k < GM_L2_OFF) scale = sqrtf(6.0f / (float)(64 + 64));
	else if (k < GM_GRID_OFF) scale = sqrtf(6.0f / (float)(16 + 64));
	else scale = 1e-4f;
	float v;
	{
#pragma clang fp contract(off)
		v = rng.next_float() * (scale - (-scale)) + (-scale);
	}
	master[k] = v;
	params[k] = (half_t)v;
	inference[k] = (half_t)v;
}

// Ema o ExponentialDecay o Adam (tcnn optimizers as configured by configs/nerf/base.json:5-22); one streaming pass.
__global__ void adam_ema_kernel(uint32_t n_params, uint32_t n_matrix_params, float lr, float beta1, float beta2, float epsilon, float l2_reg,
                                float loss_scale, float ema_decay, float ema_debias_old, float ema_debias_new, uint32_t optimize_mask,
                                const half_t* __restrict__ grads, float* __restrict__ master, half_t* __restrict__ params,
                                float* __restrict__ m1, float* __restrict__ m2, float* __restrict__ ema, half_t* __restrict__ inference) {
#pragma clang fp contract(off)
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_params) return;
	// optimize_mask bit 3 (NGP_OPT_EMA_ONLY): the Ema stage alone, no gradient is read (sharded optimizer step: the weights of other ranks' shards arrive by all-gather)
	float g = (optimize_mask & 8u) ? 0.0f : (float)grads[i] / loss_scale;
	half_t p16 = params[i];
	// optimize_mask: bit 0 = matrix (MLP) parameters, bit 1 = the others (encoding) — tcnn Adam's optimize_matrix_params / optimize_non_matrix_params
	const bool skip = (optimize_mask & 8u) ? true : (i >= n_matrix_params ? (g == 0.0f || !(optimize_mask & 2u)) : !(optimize_mask & 1u));
	if (!skip) {
		float w = master[i];
		if (i < n_matrix_params) g += l2_reg * w;
		const float gsq = g * g;
		const float fm = beta1 * m1[i] + (1.0f - beta1) * g;
		const float sm = beta2 * m2[i] + (1.0f - beta2) * gsq;
		m1[i] = fm; m2[i] = sm;
		const float eff = lr / (sqrtf(sm) + epsilon);
		const float nw = w - eff * fm;
		master[i] = nw;
		// the fp32 value must be rounded to fp32 BEFORE the fp16 conversion: without the barrier the compiler folds the last
		// multiply into v_fma_mixlo_f16 (one rounding instead of two) and ties round differently from a plain CPU evaluation
		float nw_rounded = nw;
		asm volatile("" : "+v"(nw_rounded));
		p16 = (half_t)nw_rounded;
		params[i] = p16;
	}
	if (optimize_mask & 4u) return;   // NGP_OPT_NO_EMA: the Adam stage alone
	float filtered = (ema[i] * ema_decay * ema_debias_old + (float)p16 * (1.0f - ema_decay)) * ema_debias_new;
	asm volatile("" : "+v"(filtered));
	ema[i] = filtered;
	inference[i] = (half_t)filtered;
}

// the same update, four consecutive parameters per thread with 8 / 16-byte accesses (the scalar kernel moves 2- and 4-byte words and
// tops out near 3 TB/s).  Element-wise identical arithmetic; the moments / master weights of a group are only touched when one of its
// four parameters has work (hash-grid entries without gradient: most of them), and then rewritten with unchanged bits for the others.
struct alignas(8) half4_t { half_t v[4]; };
__global__ void __launch_bounds__(256) adam_ema_vec4_kernel(uint32_t n_groups, uint32_t n_matrix_params, float lr, float beta1, float beta2, float epsilon, float l2_reg,
                                     float loss_scale, float ema_decay, float ema_debias_old, float ema_debias_new, uint32_t optimize_mask,
                                     const half4_t* __restrict__ grads, float4* __restrict__ master, half4_t* __restrict__ params,
                                     float4* __restrict__ m1, float4* __restrict__ m2, float4* __restrict__ ema, half4_t* __restrict__ inference) {
#pragma clang fp contract(off)
	NGP_RAISE_CHAIN_PRIORITY();
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_groups) return;
	half4_t g4 = {};
	if (!(optimize_mask & 8u)) g4 = grads[t];
	half4_t p4 = params[t];
	float g[4]; bool skip[4]; bool any = false;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		g[k] = (float)g4.v[k] / loss_scale;
		skip[k] = (optimize_mask & 8u) ? true : (t * 4u + k >= n_matrix_params ? (g[k] == 0.0f || !(optimize_mask & 2u)) : !(optimize_mask & 1u));
		any |= !skip[k];
	}
	if (any) {
		float4 w4 = master[t], a4 = m1[t], b4 = m2[t];
		float* w = &w4.x; float* a = &a4.x; float* b = &b4.x;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (skip[k]) continue;
			float gk = g[k];
			if (t * 4u + k < n_matrix_params) gk += l2_reg * w[k];
			const float gsq = gk * gk;
			const float fm = beta1 * a[k] + (1.0f - beta1) * gk;
			const float sm = beta2 * b[k] + (1.0f - beta2) * gsq;
			a[k] = fm; b[k] = sm;
			const float eff = lr / (sqrtf(sm) + epsilon);
			float nw = w[k] - eff * fm;
			asm volatile("" : "+v"(nw));   // fp32 rounding before the fp16 conversion (see adam_ema_kernel)
			w[k] = nw;
			p4.v[k] = (half_t)nw;
		}
		master[t] = w4; m1[t] = a4; m2[t] = b4; params[t] = p4;
	}
	if (optimize_mask & 4u) return;
	const float4 e4 = ema[t];
	float4 f4; float* f = &f4.x; const float* e = &e4.x;
	half4_t i4;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		float filtered = (e[k] * ema_decay * ema_debias_old + (float)p4.v[k] * (1.0f - ema_decay)) * ema_debias_new;
		asm volatile("" : "+v"(filtered));
		f[k] = filtered;
		i4.v[k] = (half_t)filtered;
	}
	ema[t] = f4; inference[t] = i4;
}

static uint32_t next_multiple_u32(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

static int fwd_grid(uint32_t n) {
	uint32_t tiles = div_up(n, 32);
	uint32_t blocks = div_up(tiles, 4);
	// grid-stride beyond 4 workgroups per CU: measured over the whole training step (caps 512 ... 2048 in steps of 128-256, repeated), 1024 is
	// ~3 % ahead of the former 2048 — fewer, longer-lived workgroups leave the concurrently running march and the side stream more room
	// (round 5, profiles/r05_launch_constants.md: the fox photographs do not care — 512 ... 2048 within 1.5 % —, the lego stand-in keeps its optimum at 1024)
	static const uint32_t cap = ngp_dev_knob_u32("NGP_HIP_FWD_CAP", 256 * 4);
	return (int)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

// hash-grid backward for all 16 levels: binned fixed-point path for the hashed levels, LDS owner-computes path for the dense ones
template <int D>
static int launch_grid_backward(hipStream_t st, const NgpNetDesc* desc_dev, const float* pos, uint32_t stride, uint32_t n, const h2* dx_planes, h2* gb_partials, void* fx_scratch, h2* grid_grad,
                                bool counters_cleared = false /* by the kernel that produced dx_planes */, bool ordered = true /* the batch is in ray order: NeRF training */,
                                WgradJob wgrad = WgradJob{nullptr, 0u, nullptr, 0u} /* sum these weight-gradient partials in the count pass's launch */,
                                const NgpNetDesc* desc_host = nullptr /* the level table on the host: only the owners that own something are launched */,
                                const uint32_t* n_live = nullptr /* device word: the samples behind the first *n_live carry zero gradients (the binning skips their chunks) */) {
	GbFxCounters* ctr = (GbFxCounters*)fx_scratch;
	static_assert(sizeof(GbFxCounters) <= GB_FX_COUNTER_BYTES, "counter block");
	uint32_t* wg_hist = (uint32_t*)((char*)fx_scratch + GB_FX_COUNTER_BYTES);   // the dense levels' per-workgroup bin counts
	void* records = (char*)wg_hist + gb_wg_hist_bytes(n);                         // gb_level_records: the levels' record spaces, packed
	const GbRecordOffsets rec_off = gb_record_offsets<D>(desc_host, n);          // (no host level table: every level at the worst case, as sized by the *_scratch_bytes(n) functions)
	const uint32_t level_mask = ngp_dev_knob_u32("NGP_HIP_GB_LEVELS", 0xffffu);   // dev-only timing ablation, re-read per launch in the development build (tools/gb_level_probe.py flips it); a constant in the product library (ngp_dev_knobs.h)
	if (!counters_cleared) NGP_HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(GbFxCounters), st));
	const dim3 bin_grid(div_up(n, GB_FX_CHUNK), 16);
	const dim3 count_grid(bin_grid.x, 16u + (wgrad.partials ? div_up(div_up(wgrad.n_params, 64u), bin_grid.x) : 0u));
	const WgradJob no_job{nullptr, 0u, nullptr, 0u};
	if (ordered) hipLaunchKernelGGL((gb_fx_bin_kernel<D, false, true>), count_grid, dim3(256), 0, st, desc_dev, pos, stride, n, dx_planes, ctr, records, rec_off, wg_hist, level_mask, wgrad, n_live);
	else hipLaunchKernelGGL((gb_fx_bin_kernel<D, false, false>), count_grid, dim3(256), 0, st, desc_dev, pos, stride, n, dx_planes, ctr, records, rec_off, wg_hist, level_mask, wgrad, n_live);
	NGP_LAUNCH_CHECK("gb_fx_bin_kernel<count>");
	// (round 5: write-combining the records through LDS — ranks and a per-bin image in LDS, coalesced copy-out — was built in two versions and measured slower than these
	// register-to-global stores: the pass is within 1.5x of the rate at which the part writes its 76-138 MB of records; profiles/r05_experiments.md)
	if (ordered) hipLaunchKernelGGL((gb_fx_bin_kernel<D, true, true>), bin_grid, dim3(256), 0, st, desc_dev, pos, stride, n, dx_planes, ctr, records, rec_off, wg_hist, level_mask, no_job, n_live);
	else hipLaunchKernelGGL((gb_fx_bin_kernel<D, true, false>), bin_grid, dim3(256), 0, st, desc_dev, pos, stride, n, dx_planes, ctr, records, rec_off, wg_hist, level_mask, no_job, n_live);
	NGP_LAUNCH_CHECK("gb_fx_bin_kernel<scatter>");
	static const uint32_t owner_threads = ngp_dev_knob_u32("NGP_HIP_GB_OWNER_THREADS", 1024u);   // dev: sweep (256 / 512 / 1024)
	GbOwnerMap map{};
	dim3 owner_grid(GB_FX_MAX_SLICES, 16);
	if (desc_host) {
		uint32_t total = 0;
		for (int l = 0; l < 16; ++l) { map.start[l] = total; total += gb_level_owners<D>(desc_host->levels[l]); }
		map.start[16] = total; map.compact = 1u;
		owner_grid = dim3(total ? total : 1u, 1);
	}
	if (ordered) hipLaunchKernelGGL((grid_backward_kernel<D, true>), owner_grid, dim3(owner_threads), 0, st, desc_dev, pos, stride, n, dx_planes, (void*)gb_partials, (const GbFxCounters*)ctr, records, rec_off, grid_grad, level_mask, map);
	else hipLaunchKernelGGL((grid_backward_kernel<D, false>), owner_grid, dim3(owner_threads), 0, st, desc_dev, pos, stride, n, dx_planes, (void*)gb_partials, (const GbFxCounters*)ctr, records, rec_off, grid_grad, level_mask, map);
	NGP_LAUNCH_CHECK("grid_backward_kernel");
	hipLaunchKernelGGL(grid_combine_kernel, dim3(128, 16), dim3(256), 0, st, desc_dev, (const void*)gb_partials, grid_grad, (uint32_t)D);
	NGP_LAUNCH_CHECK("grid_combine_kernel");
	return 0;
}

#include "network_generic.cuh"
#include "network_netx_mfma.cuh"

// the same read-out for a network variant (layer map of NerfNetwork::width, nerf_network.h:474-484: 0 encoding, 1 density hidden layer, 2 colour-network input
// [density out | SH | extra dims], 3 .. 2 + n_hidden the colour network's hidden layers); layout from gen_layout.  One thread per sample, activations rounded to fp16 between layers.
__global__ void __launch_bounds__(256) netx_visualize_activation_kernel(GenLayout L, const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, uint32_t layer, uint32_t dim,
                                                                        const float* __restrict__ coords, uint32_t coord_stride, uint32_t n, float* __restrict__ out, uint32_t out_stride,
                                                                        const float* __restrict__ extra_dims, const uint32_t* __restrict__ sample_slot) {
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n) return;
	const float* c = coords + (size_t)s * coord_stride;
	const f3_t pv = load_pos3(c);
	const h2* __restrict__ grid = (const h2*)(params + L.n_mlp);
	half_t x[32];
	for (int l = 0; l < 16; ++l) encode_level<false>(desc->levels[l], grid, pv.x, pv.y, pv.z, x[2 * l], x[2 * l + 1]);
	float v = 0.0f;
	if (layer == 0) v = (float)x[dim & 31];
	else {
		half_t cur[64], nxt[64];
		for (int o = 0; o < 64; ++o) nxt[o] = (half_t)fmaxf(vis_dot(params + L.off[0] + o * 32, x, 32), 0.0f);
		if (layer == 1) v = (float)nxt[dim & 63];
		else {
			for (int o = 0; o < 16; ++o) cur[o] = (half_t)vis_dot(params + L.off[1] + o * 64, nxt, 64);
			const h8 a = sh4_half(0, c[4], c[5], c[6]), b = sh4_half(1, c[4], c[5], c[6]);
			for (int o = 0; o < 8; ++o) { cur[16 + o] = a[o]; cur[24 + o] = b[o]; }
			for (uint32_t o = 32; o < 64; ++o) cur[o] = (half_t)0.0f;
			if (L.n_extra && extra_dims) {
				const float* row = extra_dims + (size_t)(sample_slot ? sample_slot[s] : 0u) * L.n_extra;
				for (uint32_t o = 0; o < L.n_extra; ++o) cur[32 + o] = (half_t)row[o];
			}
			if (layer == 2) v = (float)cur[dim < L.rgb_in ? dim : 0];
			else {
				uint32_t width = L.rgb_in;
				for (uint32_t m = 2; m + 1 < L.n_mats; ++m) {   // the colour network's hidden matrices (the last matrix is the output layer)
					for (int o = 0; o < 64; ++o) nxt[o] = (half_t)fmaxf(vis_dot(params + L.off[m] + o * width, cur, (int)width), 0.0f);
					if (layer == m + 1) { v = (float)nxt[dim & 63]; break; }
					for (int o = 0; o < 64; ++o) cur[o] = nxt[o];
					width = 64;
				}
			}
		}
	}
	float* o = out + (size_t)s * out_stride;
	if (out_stride == 1) { o[0] = v; return; }
	for (uint32_t k = 0; k < out_stride; ++k) o[k] = k == 0 ? fmaxf(-v, 0.0f) : k == 1 ? fmaxf(v, 0.0f) : k == 2 ? 0.0f : 1.0f;
}


// NgpNetVariant -> what the generic kernels take; returns false for the base family (which stays on the fused kernels)
static bool variant_is_generic(const NgpNetVariant* v) { return v && (v->n_extra_dims != 0 || v->n_rgb_hidden_layers != 2); }
static int variant_check(const NgpNetVariant* v, const char* who) {
	if (v && (v->n_extra_dims > 16 || v->n_rgb_hidden_layers > 3)) { set_last_error(who, hipErrorInvalidValue); return -1; }
	return 0;
}
static int gen_grid(uint32_t n_groups) { const uint32_t b = div_up(n_groups, 4u); return (int)(b < 2048u ? (b ? b : 1u) : 2048u); }

} // namespace ngp

using namespace ngp;

extern "C" {

uint32_t ngp_hip_net_mlp_params_host(const NgpNetVariant* variant) {
	if (!variant) return NGP_MLP_N_PARAMS;
	return gen_layout(variant->n_extra_dims, variant->n_rgb_hidden_layers).n_mlp;
}

int ngp_hip_net_make_desc_host(uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, NgpNetDesc* d) {
	if (n_levels != 16 || !d) { set_last_error("ngp_hip_net_make_desc_host: n_levels must be 16", hipErrorInvalidValue); return -1; }
	uint32_t offset = 0;
	const float log2_pls = log2f(per_level_scale);
	for (uint32_t l = 0; l < n_levels; ++l) {
		const float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
		const uint32_t res = (uint32_t)ceilf(scale) + 1u;
		const uint64_t dense = (uint64_t)res * res * res;
		const uint32_t max_params = 0xffffffffu / 2u;
		uint32_t cnt = dense > max_params ? max_params : (uint32_t)dense;
		cnt = (cnt + 7u) / 8u * 8u;
		const uint32_t hashmap = 1u << log2_hashmap_size;
		const uint32_t size = cnt < hashmap ? cnt : hashmap;
		d->levels[l].scale = scale; d->levels[l].resolution = res; d->levels[l].offset = offset; d->levels[l].size = size;
		// the hash-grid backward's owners cover at most 256 slices of 16384 entries per level (log2_hashmap_size <= 22); refused here rather than trained wrongly
		if (size > GB_FX_MAX_SLICES * GB_SLICE) { set_last_error("make_desc: a level above 2^22 entries (log2_hashmap_size > 22) is outside what the hash-grid backward covers", hipErrorInvalidValue); return -1; }
		offset += size;
	}
	d->n_levels = n_levels;
	d->n_grid_entries = offset;
	return 0;
}

uint32_t ngp_hip_net_n_params_host(const NgpNetDesc* d) { return NGP_MLP_N_PARAMS + 2u * d->n_grid_entries; }

int ngp_hip_nerf_init_params(void* stream, const NgpNetDesc* desc_host, uint64_t seed, float* master, uint16_t* params, uint16_t* inference_params, const NgpNetVariant* variant) {
	// pcg32(initstate = seed, initseq = 1)
	uint64_t state = 0u, inc = (1ull << 1u) | 1u;
	state = state * 0x5851f42d4c957f2dULL + inc;
	state += seed;
	state = state * 0x5851f42d4c957f2dULL + inc;
	if (variant_check(variant, "ngp_hip_nerf_init_params: at most 16 extra dims and 3 hidden colour layers")) return -1;
	if (variant_is_generic(variant)) {
		const GenLayout L = gen_layout(variant->n_extra_dims, variant->n_rgb_hidden_layers);
		const uint32_t n = L.n_mlp + 2u * desc_host->n_grid_entries;
		hipLaunchKernelGGL(gen_init_params_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, L, n, state, inc, master, (half_t*)params, (half_t*)inference_params);
		NGP_LAUNCH_CHECK("gen_init_params_kernel");
		return 0;
	}
	const uint32_t n = ngp_hip_net_n_params_host(desc_host);
	hipLaunchKernelGGL(init_params_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, state, inc, master, (half_t*)params, (half_t*)inference_params);
	NGP_LAUNCH_CHECK("init_params_kernel");
	return 0;
}

// ---- the variants on the MFMA kernels of network_netx_mfma.cuh.  NgpNetVariant::flags & NGP_NETX_SCALAR keeps the scalar kernels of network_generic.cuh (the checker).
static bool variant_scalar(const NgpNetVariant* v) { return v && (v->flags & NGP_NETX_SCALAR); }
// mode 0 inference, 1 density, 2 training forward; planes != NULL: the features were encoded into level planes (the _ws entry points)
static int nx_forward(void* stream, int mode, const NgpNetVariant* v, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t stride, uint32_t n, uint16_t* out, uint32_t out_stride,
                      uint16_t* x_saved, const h2* planes, uint32_t n_pad) {
	NxFwdArgs a{desc_dev, (const half_t*)params, coords, stride, n, (half_t*)out, out_stride, (half_t*)x_saved, planes, n_pad, v->extra_dims, v->sample_slot, v->n_extra_dims,
	            gen_layout(v->n_extra_dims, v->n_rgb_hidden_layers).n_mlp};
	const dim3 grid(fwd_grid(n));
	hipStream_t st = (hipStream_t)stream;
	int rc = 0;
	if (mode == 1) { if (planes) nx_fwd_launch<1, 1, 0, 0>(grid, st, a); else nx_fwd_launch<1, 0, 0, 0>(grid, st, a); }
	else if (mode == 2) {
		if (planes) { set_last_error("nx_forward: the training forward of a network variant has no plane path", hipErrorInvalidValue); return -1; }
		rc = nx_fwd_dispatch<2, 0>(v->n_rgb_hidden_layers, v->n_extra_dims != 0, grid, st, a);
	}
	else if (planes) rc = nx_fwd_dispatch<0, 1>(v->n_rgb_hidden_layers, v->n_extra_dims != 0, grid, st, a);
	else rc = nx_fwd_dispatch<0, 0>(v->n_rgb_hidden_layers, v->n_extra_dims != 0, grid, st, a);
	if (rc) { set_last_error("network variant outside 0..3 hidden colour layers x extra dims or not", hipErrorInvalidValue); return -1; }
	NGP_LAUNCH_CHECK("nx_forward_kernel");
	return 0;
}

static int gen_forward(void* stream, int mode, const NgpNetVariant* v, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t stride, uint32_t n, uint16_t* out, uint32_t out_stride, uint16_t* x_saved) {
	const GenLayout L = gen_layout(v->n_extra_dims, v->n_rgb_hidden_layers);
	GenExtra ex; ex.extra_dims = v->extra_dims; ex.sample_slot = v->sample_slot;
	const int grid = gen_grid(div_up(n, (uint32_t)GEN_SG));
	if (mode == 1) hipLaunchKernelGGL(gen_forward_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, L, desc_dev, (const half_t*)params, coords, stride, n, (half_t*)out, 1u, (half_t*)nullptr, ex);
	else if (mode == 2) hipLaunchKernelGGL(gen_forward_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, L, desc_dev, (const half_t*)params, coords, stride, n, (half_t*)out, out_stride, (half_t*)x_saved, ex);
	else hipLaunchKernelGGL(gen_forward_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, L, desc_dev, (const half_t*)params, coords, stride, n, (half_t*)out, out_stride, (half_t*)nullptr, ex);
	NGP_LAUNCH_CHECK("gen_forward_kernel");
	return 0;
}

int ngp_hip_nerf_inference(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                           uint32_t n, uint16_t* out, uint32_t out_stride, const NgpNetVariant* variant) {
	if (n == 0) return 0;
	if (out_stride < 4 || (out_stride & 3)) { set_last_error("ngp_hip_nerf_inference: out_stride must be a multiple of 4", hipErrorInvalidValue); return -1; }
	if (variant_check(variant, "ngp_hip_nerf_inference: at most 16 extra dims and 3 hidden colour layers")) return -1;
	if (variant_is_generic(variant)) return variant_scalar(variant) ? gen_forward(stream, 0, variant, desc_dev, params, coords, coord_stride_floats, n, out, out_stride, nullptr)
	                                                                 : nx_forward(stream, 0, variant, desc_dev, params, coords, coord_stride_floats, n, out, out_stride, nullptr, nullptr, 0u);
	hipLaunchKernelGGL((nerf_forward_kernel<0, 0>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, coord_stride_floats, n, (half_t*)out, out_stride, (half_t*)nullptr, (const h2*)nullptr, 0u);
	NGP_LAUNCH_CHECK("nerf_forward_kernel<0>");
	return 0;
}

int ngp_hip_nerf_density(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n, uint16_t* out0, const NgpNetVariant* variant) {
	if (n == 0) return 0;
	if (variant_is_generic(variant)) {   // the density network is the same, the grid sits behind a different number of MLP parameters; the direction / extra dims are not read
		NgpNetVariant v = *variant; v.extra_dims = nullptr; v.sample_slot = nullptr;
		if (!variant_scalar(variant)) return nx_forward(stream, 1, &v, desc_dev, params, pos, pos_stride_floats, n, out0, 1, nullptr, nullptr, 0u);
		return gen_forward(stream, 1, &v, desc_dev, params, pos, pos_stride_floats, n, out0, 1, nullptr);
	}
	hipLaunchKernelGGL((nerf_forward_kernel<1, 0>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, pos, pos_stride_floats, n, (half_t*)out0, 1u, (half_t*)nullptr, (const h2*)nullptr, 0u);
	NGP_LAUNCH_CHECK("nerf_forward_kernel<1>");
	return 0;
}

int ngp_hip_nerf_forward(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                         uint32_t n, uint16_t* out, uint32_t out_stride, uint16_t* x_saved, const NgpNetVariant* variant) {
	if (n == 0) return 0;
	if (out_stride < 4 || (out_stride & 3)) { set_last_error("ngp_hip_nerf_forward: out_stride must be a multiple of 4", hipErrorInvalidValue); return -1; }
	if (variant_check(variant, "ngp_hip_nerf_forward: at most 16 extra dims and 3 hidden colour layers")) return -1;
	if (variant_is_generic(variant)) return variant_scalar(variant) ? gen_forward(stream, 2, variant, desc_dev, params, coords, coord_stride_floats, n, out, out_stride, x_saved)
	                                                                 : nx_forward(stream, 2, variant, desc_dev, params, coords, coord_stride_floats, n, out, out_stride, x_saved, nullptr, 0u);
	hipLaunchKernelGGL((nerf_forward_kernel<2, 0>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, coord_stride_floats, n, (half_t*)out, out_stride, (half_t*)x_saved, (const h2*)nullptr, 0u);
	NGP_LAUNCH_CHECK("nerf_forward_kernel<2>");
	return 0;
}


// ---- two-kernel variants: XCD-affine encode into level planes (workspace), then the MLP kernel
uint64_t ngp_hip_nerf_encode_workspace_bytes(uint32_t n) { return ENC_QUEUE_BYTES + (uint64_t)16 * next_multiple_u32(n, ENC_CHUNK) * 4u; }

static int launch_encode(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t stride, uint32_t n, void* workspace, uint64_t workspace_bytes,
                         const h2** planes_out, uint32_t* n_pad_out, const char* who, const NgpNetDesc* desc_host_for_groups = nullptr, uint32_t grid_off = GRID_OFF, uint32_t n_dims = 3) {
	if (!workspace || workspace_bytes < ngp_hip_nerf_encode_workspace_bytes(n)) { set_last_error(who, hipErrorInvalidValue); return -1; }
	const uint32_t n_pad = next_multiple_u32(n, ENC_CHUNK);
	uint32_t* queues = (uint32_t*)workspace;
	h2* planes = (h2*)((char*)workspace + ENC_QUEUE_BYTES);
	static const int enc_mode = (int)ngp_dev_knob_u32("NGP_HIP_ENC_MODE", 0);   // dev: 1 = one launch per group of levels (NGP_HIP_ENC_GROUP_KIB of table each)
	static NgpNetDesc cached_desc; static const NgpNetDesc* cached_for = nullptr;   // dev experiment: the level sizes decide the grouping
	if (enc_mode == 1 && n_dims == 3 && !desc_host_for_groups) {
		if (cached_for != desc_dev) { NGP_HIP_TRY(hipMemcpy(&cached_desc, desc_dev, sizeof(NgpNetDesc), hipMemcpyDeviceToHost)); cached_for = desc_dev; }
		desc_host_for_groups = &cached_desc;
	}
	if (enc_mode == 1 && n_dims == 3 && desc_host_for_groups) {
		static const uint32_t cap_kib = ngp_dev_knob_u32("NGP_HIP_ENC_GROUP_KIB", 2560u);
		static const int pair = (int)ngp_dev_knob_u32("NGP_HIP_ENC_PAIR", 1);
		const uint32_t blocks = div_up(n, 256u) < 2048u ? div_up(n, 256u) : 2048u;
		uint32_t l0 = 0;
		while (l0 < 16) {
			uint64_t bytes = (uint64_t)desc_host_for_groups->levels[l0].size * 4u;
			uint32_t l1 = l0 + 1;
			while (l1 < 16 && bytes + (uint64_t)desc_host_for_groups->levels[l1].size * 4u <= (uint64_t)cap_kib * 1024u) { bytes += (uint64_t)desc_host_for_groups->levels[l1].size * 4u; ++l1; }
			if (pair) hipLaunchKernelGGL(encode_levels_planes_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, stride, n, n_pad, planes, l0, l1, grid_off);
			else hipLaunchKernelGGL(encode_levels_planes_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, stride, n, n_pad, planes, l0, l1, grid_off);
			NGP_LAUNCH_CHECK("encode_levels_planes_kernel");
			l0 = l1;
		}
		*planes_out = planes; *n_pad_out = n_pad;
		return 0;
	}
	NGP_HIP_TRY(hipMemsetAsync(queues, 0, ENC_QUEUE_BYTES, (hipStream_t)stream));
	const uint32_t items = 16u * (n_pad / ENC_CHUNK);
	// persistent workgroups and items per claim by the size of the pass (round 5, profiles/r05_encoder_constants.md: swept on a fox step's own 515 k samples and on 2^18 ... 8 M random points): a
	// small pass wants single-item claims (its tail is a third of it), a large one four items per claim; four workgroups per CU beat eight at every size
	// (a pass of 2^18 samples or fewer — the SDF config's batch, a tracer's late passes — is 6-8 % faster on 768 workgroups: 104 / 123 us against 113 / 131)
	const uint32_t blocks_default = items <= 4096u ? 768u : 1024u, claim_default = items <= 12288u ? 1u : items <= 32768u ? 2u : 4u;
	const uint32_t blocks_cap = ngp_dev_knob_u32("NGP_HIP_ENC_BLOCKS", blocks_default), items_per_claim = ngp_dev_knob_u32("NGP_HIP_ENC_CLAIM", claim_default);
	const uint32_t blocks = items < blocks_cap ? items : blocks_cap;
	static const uint32_t cost_model = ngp_dev_knob_u32("NGP_HIP_ENC_COST", 0u);   // dev: A / B of the queue cut
	if (n_dims == 2) hipLaunchKernelGGL(encode_planes_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, stride, n, n_pad, planes, queues, cost_model, grid_off, items_per_claim);
	else hipLaunchKernelGGL(encode_planes_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, stride, n, n_pad, planes, queues, cost_model, grid_off, items_per_claim);
	NGP_LAUNCH_CHECK("encode_planes_kernel");
	*planes_out = planes; *n_pad_out = n_pad;
	return 0;
}

int ngp_hip_nerf_inference_ws(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                              uint32_t n, uint16_t* out, uint32_t out_stride, void* workspace, uint64_t workspace_bytes, const NgpNetVariant* variant) {
	if (n == 0) return 0;
	if (variant_scalar(variant)) return ngp_hip_nerf_inference(stream, desc_dev, params, coords, coord_stride_floats, n, out, out_stride, variant);   // the scalar checker has one path
	if (out_stride < 4 || (out_stride & 3)) { set_last_error("ngp_hip_nerf_inference_ws: out_stride must be a multiple of 4", hipErrorInvalidValue); return -1; }
	const h2* planes; uint32_t n_pad;
	if (variant_is_generic(variant)) {   // same two kernels, the grid behind the variant's MLP parameters, the variant's MLP kernel over the planes
		if (variant_check(variant, "ngp_hip_nerf_inference_ws: at most 16 extra dims and 3 hidden colour layers")) return -1;
		if (launch_encode(stream, desc_dev, params, coords, coord_stride_floats, n, workspace, workspace_bytes, &planes, &n_pad, "ngp_hip_nerf_inference_ws: workspace too small", nullptr, ngp_hip_net_mlp_params_host(variant))) return -1;
		return nx_forward(stream, 0, variant, desc_dev, params, coords, coord_stride_floats, n, out, out_stride, nullptr, planes, n_pad);
	}
	if (launch_encode(stream, desc_dev, params, coords, coord_stride_floats, n, workspace, workspace_bytes, &planes, &n_pad, "ngp_hip_nerf_inference_ws: workspace too small")) return -1;
	hipLaunchKernelGGL((nerf_forward_kernel<0, 1>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, coord_stride_floats, n, (half_t*)out, out_stride, (half_t*)nullptr, planes, n_pad);
	NGP_LAUNCH_CHECK("nerf_forward_kernel<0, pre>");
	return 0;
}

int ngp_hip_nerf_density_ws(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n, uint16_t* out0,
                            void* workspace, uint64_t workspace_bytes, const NgpNetVariant* variant) {
	if (n == 0) return 0;
	if (variant_scalar(variant)) return ngp_hip_nerf_density(stream, desc_dev, params, pos, pos_stride_floats, n, out0, variant);
	const h2* planes; uint32_t n_pad;
	if (variant_is_generic(variant)) {
		if (launch_encode(stream, desc_dev, params, pos, pos_stride_floats, n, workspace, workspace_bytes, &planes, &n_pad, "ngp_hip_nerf_density_ws: workspace too small", nullptr, ngp_hip_net_mlp_params_host(variant))) return -1;
		return nx_forward(stream, 1, variant, desc_dev, params, pos, pos_stride_floats, n, out0, 1, nullptr, planes, n_pad);
	}
	if (launch_encode(stream, desc_dev, params, pos, pos_stride_floats, n, workspace, workspace_bytes, &planes, &n_pad, "ngp_hip_nerf_density_ws: workspace too small")) return -1;
	hipLaunchKernelGGL((nerf_forward_kernel<1, 1>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, pos, pos_stride_floats, n, (half_t*)out0, 1u, (half_t*)nullptr, planes, n_pad);
	NGP_LAUNCH_CHECK("nerf_forward_kernel<1, pre>");
	return 0;
}

int ngp_hip_nerf_forward_ws(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                            uint32_t n, uint16_t* out, uint32_t out_stride, uint16_t* x_saved, void* workspace, uint64_t workspace_bytes, const NgpNetVariant* variant) {
	if (n == 0) return 0;
	if (variant_is_generic(variant)) return ngp_hip_nerf_forward(stream, desc_dev, params, coords, coord_stride_floats, n, out, out_stride, x_saved, variant);
	if (out_stride < 4 || (out_stride & 3)) { set_last_error("ngp_hip_nerf_forward_ws: out_stride must be a multiple of 4", hipErrorInvalidValue); return -1; }
	const h2* planes; uint32_t n_pad;
	if (launch_encode(stream, desc_dev, params, coords, coord_stride_floats, n, workspace, workspace_bytes, &planes, &n_pad, "ngp_hip_nerf_forward_ws: workspace too small")) return -1;
	hipLaunchKernelGGL((nerf_forward_kernel<2, 1>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, coords, coord_stride_floats, n, (half_t*)out, out_stride, (half_t*)x_saved, planes, n_pad);
	NGP_LAUNCH_CHECK("nerf_forward_kernel<2, pre>");
	return 0;
}

// scratch layout: [weight-gradient partials 512 x 10240 fp32][dL/dx planes 16 x n half2][grid partials 16 x 4 MiB][binned path: counters, item lists, run sums]
constexpr uint32_t FB_MAX_WORKGROUPS = 512;   // two resident workgroups per CU
static uint64_t scratch_off_dx(uint32_t) { return (uint64_t)FB_MAX_WORKGROUPS * NGP_MLP_N_PARAMS * 4u; }
static uint64_t scratch_off_gb(uint32_t n) { return scratch_off_dx(n) + (uint64_t)16 * n * 4u; }
// the binned path's part: counters, the dense levels' per-workgroup bin counts, the levels' record spaces (gb_level_records) — for a given level table, or (no table) for any
static uint64_t gb_fx_bytes_for(const NgpNetDesc* desc_host, uint32_t n_dims, uint32_t n) {
	return GB_FX_COUNTER_BYTES + gb_wg_hist_bytes(n) + (n_dims == 2 ? gb_records_bytes<2>(desc_host, n) : gb_records_bytes<3>(desc_host, n));
}
static uint64_t gb_fx_bytes(uint32_t n) { return gb_fx_bytes_for(nullptr, 3, n); }
static uint64_t scratch_off_fx_for(const NgpNetDesc* desc_host, uint32_t n) { return scratch_off_gb(n) + gb_partials_bytes<3>(desc_host); }
static uint64_t scratch_off_fx(uint32_t n) { return scratch_off_fx_for(nullptr, n); }
uint64_t ngp_hip_nerf_backward_scratch_bytes(uint32_t n) { return scratch_off_fx(n) + gb_fx_bytes(n); }
uint64_t ngp_hip_nerf_backward_scratch_bytes_for(const NgpNetDesc* desc_host, uint32_t n) { return scratch_off_fx_for(desc_host, n) + gb_fx_bytes_for(desc_host, 3, n); }

static int nerf_backward_impl(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords,
                              uint32_t coord_stride_floats, uint32_t n, const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride,
                              uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* mlp_done_event, void* grid_gradients_event, float* dL_dinput, LiveSamples live, const float* grid_coords = nullptr);

static int gen_backward(void* stream, const NgpNetVariant* v, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords, uint32_t stride, uint32_t n, const uint16_t* x_saved,
                        const uint16_t* dL_dout, uint32_t dl_stride, uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* grid_gradients_event) {
	if (n == 0 || (n % 256) != 0) { set_last_error("ngp_hip_nerf_backward: n must be a positive multiple of 256", hipErrorInvalidValue); return -1; }
	if (scratch_bytes < ngp_hip_nerf_backward_scratch_bytes_for(desc_host, n)) { set_last_error("ngp_hip_nerf_backward: scratch too small (ngp_hip_nerf_backward_scratch_bytes_for(desc_host, n), or ngp_hip_nerf_backward_scratch_bytes(n) without a host level table)", hipErrorInvalidValue); return -1; }
	hipStream_t st = (hipStream_t)stream;
	const GenLayout L = gen_layout(v->n_extra_dims, v->n_rgb_hidden_layers);
	GenExtra ex; ex.extra_dims = v->extra_dims; ex.sample_slot = v->sample_slot;
	float* partials = (float*)scratch;                                    // [grid][n_mlp] fp32 (<= 256 x 15 360 x 4 B: inside the fused path's 512 x 10 240 x 4 B)
	h2* dx_planes = (h2*)((char*)scratch + scratch_off_dx(n));
	h2* gb_partials = (h2*)((char*)scratch + scratch_off_gb(n));
	static bool attr_set = false;
	if (!attr_set) { NGP_HIP_TRY(hipFuncSetAttribute((const void*)gen_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEN_BWD_SMEM)); attr_set = true; }
	const uint32_t n_iters = n / 16;
	const uint32_t grid = n_iters < 256u ? n_iters : 256u;                // one workgroup per CU (87 KiB of LDS)
	static_assert((uint64_t)256 * GEN_MAX_MLP <= (uint64_t)FB_MAX_WORKGROUPS * NGP_MLP_N_PARAMS, "partials of the generic backward fit the fused path's scratch");
	hipLaunchKernelGGL(gen_backward_kernel, dim3(grid), dim3(256), GEN_BWD_SMEM, st, L, desc_dev, (const half_t*)params, coords, stride, n, (const half_t*)x_saved, (const half_t*)dL_dout, dl_stride,
	                   dx_planes, partials, ex, v->dL_dextra);
	NGP_LAUNCH_CHECK("gen_backward_kernel");
	hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(div_up(L.n_mlp, 64)), dim3(64 * WR_WAVES), 0, st, (const float*)partials, grid, (half_t*)grads, L.n_mlp);
	NGP_LAUNCH_CHECK("wgrad_reduce_kernel");
	if (launch_grid_backward<3>(st, desc_dev, coords, stride, n, (const h2*)dx_planes, gb_partials, (char*)scratch + scratch_off_fx_for(desc_host, n), (h2*)(grads + L.n_mlp), false, true, WgradJob{nullptr, 0u, nullptr, 0u}, desc_host)) return -1;
	if (grid_gradients_event) NGP_HIP_TRY(hipEventRecord((hipEvent_t)grid_gradients_event, st));
	return 0;
}

// recompute + dgrad + weight gradients of a network variant on the MFMA kernels (one or two launches of nx_backward_kernel), then the hash-grid backward
static int nx_backward(void* stream, const NgpNetVariant* v, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords, uint32_t stride, uint32_t n, const uint16_t* x_saved,
                       const uint16_t* dL_dout, uint32_t dl_stride, uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* mlp_done_event, void* grid_gradients_event, float* dL_dinput) {
	if (n == 0 || (n % 256) != 0) { set_last_error("ngp_hip_nerf_backward: n must be a positive multiple of 256", hipErrorInvalidValue); return -1; }
	if (scratch_bytes < ngp_hip_nerf_backward_scratch_bytes_for(desc_host, n)) { set_last_error("ngp_hip_nerf_backward: scratch too small (ngp_hip_nerf_backward_scratch_bytes_for(desc_host, n), or ngp_hip_nerf_backward_scratch_bytes(n) without a host level table)", hipErrorInvalidValue); return -1; }
	hipStream_t st = (hipStream_t)stream;
	const GenLayout L = gen_layout(v->n_extra_dims, v->n_rgb_hidden_layers);
	float* partials = (float*)scratch;                                    // [grid][n_mlp] fp32 (<= 256 x 15 360 x 4 B: inside the fused path's 512 x 10 240 x 4 B)
	h2* dx_planes = (h2*)((char*)scratch + scratch_off_dx(n));
	h2* gb_partials = (h2*)((char*)scratch + scratch_off_gb(n));
	uint32_t* zero_words = (uint32_t*)((char*)scratch + scratch_off_fx_for(desc_host, n));
	const uint32_t n_quads = n / 128;
	const uint32_t grid = n_quads < 256u ? n_quads : 256u;                // one workgroup per CU (up to 100 KiB of LDS)
	NxBwdArgs a{desc_dev, (const half_t*)params, coords, stride, n, (const half_t*)x_saved, (const half_t*)dL_dout, dl_stride, dx_planes, partials, zero_words, (uint32_t)(sizeof(GbFxCounters) / 4),
	            v->extra_dims, v->sample_slot, v->n_extra_dims, v->dL_dextra, dL_dinput};
	if (nx_bwd_dispatch(v->n_rgb_hidden_layers, v->n_extra_dims != 0, dim3(grid), st, a)) { set_last_error("nx_backward: launch set-up failed", hipErrorInvalidValue); return -1; }
	NGP_LAUNCH_CHECK("nx_backward_kernel");
	if (dL_dinput) {
		hipLaunchKernelGGL(nerf_input_pos_gradient_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, desc_dev, (const half_t*)params, coords, stride, n, (const h2*)dx_planes, dL_dinput, L.n_mlp);
		NGP_LAUNCH_CHECK("nerf_input_pos_gradient_kernel");
	}
	if (mlp_done_event) NGP_HIP_TRY(hipEventRecord((hipEvent_t)mlp_done_event, st));
	hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(div_up(L.n_mlp, 64)), dim3(64 * WR_WAVES), 0, st, (const float*)partials, grid, (half_t*)grads, L.n_mlp);
	NGP_LAUNCH_CHECK("wgrad_reduce_kernel");
	if (launch_grid_backward<3>(st, desc_dev, coords, stride, n, (const h2*)dx_planes, gb_partials, (char*)scratch + scratch_off_fx_for(desc_host, n), (h2*)(grads + L.n_mlp), true, true, WgradJob{nullptr, 0u, nullptr, 0u}, desc_host)) return -1;
	if (grid_gradients_event) NGP_HIP_TRY(hipEventRecord((hipEvent_t)grid_gradients_event, st));
	return 0;
}

// desc_host lays out the host's side of the backward pass (the levels' record offsets, the owners' launch grid), desc_dev is what the kernels read: the two must be the same
// table.  The first call with a new (desc_dev, contents of desc_host) pair reads desc_dev back once — stream-ordered, then the host waits for it — and compares; later calls
// with the same pair cost a hash of 400 bytes.  (a mismatch used to leave gradient entries unwritten; with the packed record space it would write out of bounds.)
static int verify_desc_pair(hipStream_t st, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host) {
	struct Seen { const void* dev; uint64_t hash; };
	static std::mutex mu;
	static Seen seen[8] = {};
	static unsigned next = 0;
	uint64_t h = 1469598103934665603ull;
	const unsigned char* b = (const unsigned char*)desc_host;
	for (size_t i = 0; i < sizeof(NgpNetDesc); ++i) { h ^= b[i]; h *= 1099511628211ull; }
	{
		std::lock_guard<std::mutex> lock(mu);
		for (const Seen& e : seen) if (e.dev == (const void*)desc_dev && e.hash == h) return 0;
	}
	NgpNetDesc on_device;
	NGP_HIP_TRY(hipMemcpyAsync(&on_device, desc_dev, sizeof(NgpNetDesc), hipMemcpyDeviceToHost, st));
	NGP_HIP_TRY(hipStreamSynchronize(st));
	if (memcmp(&on_device, desc_host, sizeof(NgpNetDesc)) != 0) { set_last_error("ngp_hip_nerf_backward: desc_host is not the level table desc_dev points to (pass the host copy of the SAME NgpNetDesc, or NULL)", hipErrorInvalidValue); return -1; }
	std::lock_guard<std::mutex> lock(mu);
	seen[next++ % 8u] = Seen{(const void*)desc_dev, h};
	return 0;
}

int ngp_hip_nerf_backward(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords,
                          uint32_t coord_stride_floats, uint32_t n, const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride,
                          uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* mlp_done_event, void* grid_gradients_event, float* dL_dinput, const NgpNetVariant* variant,
                          const uint32_t* x_row_index) {
	if (variant_check(variant, "ngp_hip_nerf_backward: at most 16 extra dims and 3 hidden colour layers")) return -1;
	if (desc_host && verify_desc_pair((hipStream_t)stream, desc_dev, desc_host)) return -1;
	if (variant_is_generic(variant)) {
		if (x_row_index) { set_last_error("ngp_hip_nerf_backward: x_row_index (encoding rows left in the uncompacted batch) is built for the base network family only", hipErrorNotSupported); return -1; }
		if (!variant_scalar(variant)) return nx_backward(stream, variant, desc_dev, desc_host, params, coords, coord_stride_floats, n, x_saved, dL_dout, dl_stride, grads, scratch, scratch_bytes, mlp_done_event, grid_gradients_event, dL_dinput);
		if (dL_dinput) { set_last_error("ngp_hip_nerf_backward: dL_dinput (camera-side trainables) is not built into the scalar checker kernels (NGP_NETX_SCALAR)", hipErrorNotSupported); return -1; }
		return gen_backward(stream, variant, desc_dev, desc_host, params, coords, coord_stride_floats, n, x_saved, dL_dout, dl_stride, grads, scratch, scratch_bytes, grid_gradients_event);
	}
	return nerf_backward_impl(stream, desc_dev, desc_host, params, coords, coord_stride_floats, n, x_saved, dL_dout, dl_stride, grads, scratch, scratch_bytes, mlp_done_event, grid_gradients_event, dL_dinput, LiveSamples{nullptr, nullptr, nullptr, x_row_index});
}

// The backward pass over the LIVE samples of a batch (ngp_hip_compact_live_samples): slot k < *n_live_dev stands for row live_index[k] of coords / x_saved / dL_dout; the
// hash-grid backward reads the live samples' positions from coords_live (their rows, packed).  Base network family, no input gradient.
int ngp_hip_nerf_backward_live(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords,
                               uint32_t coord_stride_floats, uint32_t n, const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride,
                               uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* mlp_done_event, void* grid_gradients_event,
                               const uint32_t* live_index, const float* coords_live, const uint32_t* n_live_dev, uint32_t* zero_word_dev, const uint32_t* x_row_index) {
	if (!n_live_dev || !live_index || !coords_live) { set_last_error("ngp_hip_nerf_backward_live: live_index / coords_live / n_live_dev is NULL (ngp_hip_nerf_backward is the entry without a live list)", hipErrorInvalidValue); return -1; }
	if (desc_host && verify_desc_pair((hipStream_t)stream, desc_dev, desc_host)) return -1;
	return nerf_backward_impl(stream, desc_dev, desc_host, params, coords, coord_stride_floats, n, x_saved, dL_dout, dl_stride, grads, scratch, scratch_bytes, mlp_done_event, grid_gradients_event, nullptr,
	                          LiveSamples{n_live_dev, zero_word_dev, live_index, x_row_index}, coords_live);
}

// Samples of a training batch whose loss gradient is zero in all four channels — in fp16, after the roll-over: the tails of the rays, 30-45 % of a batch — contribute exact
// zeros to every sum of the backward pass.  This pass lists the others (the 2048 rows of a workgroup keep their order, the workgroups take their ranges in arrival order), copies
// their coordinate rows next to each other and counts them; ngp_hip_nerf_backward_live then runs its MFMA kernel and the binning of the hash-grid backward over them only.
constexpr uint32_t CL_ROWS = 2048;   // rows per workgroup: one atomic on the counter per 2048 rows (a workgroup per 256 rows spent 15 us of 17 queueing on that one word)
__global__ void __launch_bounds__(256) compact_live_samples_kernel(uint32_t n, const half_t* __restrict__ dl, uint32_t dl_stride, const float* __restrict__ coords, uint32_t coord_stride,
                                                                   uint32_t* __restrict__ live_index, float* __restrict__ coords_out, uint32_t* __restrict__ n_live) {
	constexpr uint32_t PER = CL_ROWS / 256;
	__shared__ uint32_t s_cnt[PER * 4 + 1], s_base, s_src[CL_ROWS];
	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	unsigned long long masks[PER];
#pragma unroll
	for (uint32_t u = 0; u < PER; ++u) {
		const uint32_t s = blockIdx.x * CL_ROWS + u * 256u + threadIdx.x;
		uint2 d = {0u, 0u};
		if (s < n) d = *(const uint2*)(dl + (size_t)s * dl_stride);
		masks[u] = __ballot(((d.x | d.y) & 0x7fff7fffu) != 0u);   // +-0 in all four channels: nothing to propagate
		if (lane == 0) s_cnt[u * 4 + w] = (uint32_t)__popcll(masks[u]);
	}
	__syncthreads();
	if (threadIdx.x == 0) {   // exclusive prefix over the 32 (pass, wave) counts, in row order
		uint32_t run = 0;
		for (uint32_t k = 0; k < PER * 4; ++k) { const uint32_t c = s_cnt[k]; s_cnt[k] = run; run += c; }
		s_cnt[PER * 4] = run;
		s_base = run ? atomicAdd(n_live, run) : 0u;
	}
	__syncthreads();
#pragma unroll
	for (uint32_t u = 0; u < PER; ++u)
		if ((masks[u] >> lane) & 1ull) s_src[s_cnt[u * 4 + w] + (uint32_t)__popcll(masks[u] & ((1ull << lane) - 1ull))] = blockIdx.x * CL_ROWS + u * 256u + threadIdx.x;
	__syncthreads();
	const uint32_t total = s_cnt[PER * 4], base = s_base;
	for (uint32_t k = threadIdx.x; k < total; k += 256u) live_index[base + k] = s_src[k];
	// the rows leave as one contiguous run of total * coord_stride floats
	for (uint32_t e = threadIdx.x; e < total * coord_stride; e += 256u) {
		const uint32_t k = e / coord_stride, f = e - k * coord_stride;
		coords_out[(size_t)base * coord_stride + e] = coords[(size_t)s_src[k] * coord_stride + f];
	}
}
int ngp_hip_compact_live_samples(void* stream, uint32_t n, const uint16_t* dL_dout, uint32_t dl_stride, const float* coords, uint32_t coord_stride_floats,
                                 uint32_t* live_index_out, float* coords_out, uint32_t* n_live_dev) {
	if (n == 0) return 0;
	if (dl_stride < 4 || (dl_stride & 3) || coord_stride_floats < 3) { set_last_error("ngp_hip_compact_live_samples: dl_stride must be a multiple of 4 halves (8-byte rows), coords rows start with the position", hipErrorInvalidValue); return -1; }
	if (!n_live_dev || !live_index_out || !coords_out) { set_last_error("ngp_hip_compact_live_samples: NULL output", hipErrorInvalidValue); return -1; }
	hipLaunchKernelGGL(compact_live_samples_kernel, dim3(div_up(n, CL_ROWS)), dim3(256), 0, (hipStream_t)stream, n, (const half_t*)dL_dout, dl_stride, coords, coord_stride_floats, live_index_out, coords_out, n_live_dev);
	NGP_LAUNCH_CHECK("compact_live_samples_kernel");
	return 0;
}

// per-image extra dims of a training step (src/testbed_nerf.cu:1136, 1246: every sample of a ray carries its image's row; :1710-1746: their gradient)
int ngp_hip_ray_images(void* stream, uint32_t n_rays_capacity, const uint32_t* rays_counter, const uint32_t* ray_indices, uint32_t n_rays_global, uint32_t n_training_images, const float* cdf_img,
                       uint32_t* ray_image) {
	if (!n_rays_capacity) return 0;
	hipLaunchKernelGGL(ray_images_kernel, dim3(div_up(n_rays_capacity, 128)), dim3(128), 0, (hipStream_t)stream, n_rays_capacity, rays_counter, ray_indices, n_rays_global, n_training_images, cdf_img, ray_image);
	NGP_LAUNCH_CHECK("ray_images_kernel");
	return 0;
}
int ngp_hip_expand_ray_slots(void* stream, uint32_t n_rays_capacity, const uint32_t* rays_counter, const uint32_t* ray_image, const uint32_t* numsteps, uint32_t n_samples_capacity, uint32_t* sample_slot) {
	if (!n_rays_capacity) return 0;
	hipLaunchKernelGGL(expand_ray_slots_kernel, dim3(div_up(n_rays_capacity, 128)), dim3(128), 0, (hipStream_t)stream, n_rays_capacity, rays_counter, ray_image, numsteps, n_samples_capacity, sample_slot);
	NGP_LAUNCH_CHECK("expand_ray_slots_kernel");
	return 0;
}
int ngp_hip_rollover_slots(void* stream, uint32_t n_elements, const uint32_t* n_input_elements, uint32_t* sample_slot) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(rollover_slots_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, n_input_elements, sample_slot);
	NGP_LAUNCH_CHECK("rollover_slots_kernel");
	return 0;
}
int ngp_hip_extra_dims_gradient(void* stream, uint32_t n_rays_capacity, const uint32_t* rays_counter, const uint32_t* ray_image, const uint32_t* numsteps, const float* dL_dextra, uint32_t n_extra_dims,
                                float* gradient) {
	if (!n_rays_capacity || !n_extra_dims) return 0;
	hipLaunchKernelGGL(extra_dims_gradient_kernel, dim3(div_up(n_rays_capacity, 128)), dim3(128), 0, (hipStream_t)stream, n_rays_capacity, rays_counter, ray_image, numsteps, dL_dextra, n_extra_dims, gradient);
	NGP_LAUNCH_CHECK("extra_dims_gradient_kernel");
	return 0;
}

static int nerf_backward_impl(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords,
                              uint32_t coord_stride_floats, uint32_t n, const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride,
                              uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* mlp_done_event, void* grid_gradients_event, float* dL_dinput, LiveSamples live, const float* grid_coords /* positions for the hash-grid backward (NULL: coords) */) {
	if (n == 0 || (n % 256) != 0) { set_last_error("ngp_hip_nerf_backward: n must be a positive multiple of 256", hipErrorInvalidValue); return -1; }
	if (scratch_bytes < ngp_hip_nerf_backward_scratch_bytes_for(desc_host, n)) { set_last_error("ngp_hip_nerf_backward: scratch too small (ngp_hip_nerf_backward_scratch_bytes_for(desc_host, n), or ngp_hip_nerf_backward_scratch_bytes(n) without a host level table)", hipErrorInvalidValue); return -1; }
	hipStream_t st = (hipStream_t)stream;
	float* partials = (float*)scratch;
	h2* dx_planes = (h2*)((char*)scratch + scratch_off_dx(n));
	h2* gb_partials = (h2*)((char*)scratch + scratch_off_gb(n));
	const uint32_t n_quads = n / 128;
	const uint32_t grid = n_quads < FB_MAX_WORKGROUPS ? n_quads : FB_MAX_WORKGROUPS;
	if (dL_dinput) hipLaunchKernelGGL(nerf_backward_fused_kernel<true>, dim3(grid), dim3(256), 0, st, desc_dev, (const half_t*)params, coords, coord_stride_floats, n, (const half_t*)x_saved, (const half_t*)dL_dout, dl_stride,
	                                  dx_planes, partials, (uint32_t*)((char*)scratch + scratch_off_fx_for(desc_host, n)), (uint32_t)(sizeof(GbFxCounters) / 4), dL_dinput, live);
	else hipLaunchKernelGGL(nerf_backward_fused_kernel<false>, dim3(grid), dim3(256), 0, st, desc_dev, (const half_t*)params, coords, coord_stride_floats, n, (const half_t*)x_saved, (const half_t*)dL_dout, dl_stride,
	                        dx_planes, partials, (uint32_t*)((char*)scratch + scratch_off_fx_for(desc_host, n)), (uint32_t)(sizeof(GbFxCounters) / 4), (float*)nullptr, live);
	NGP_LAUNCH_CHECK("nerf_backward_fused_kernel");
	if (dL_dinput) {
		hipLaunchKernelGGL(nerf_input_pos_gradient_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, desc_dev, (const half_t*)params, coords, coord_stride_floats, n, (const h2*)dx_planes, dL_dinput, (uint32_t)GRID_OFF);
		NGP_LAUNCH_CHECK("nerf_input_pos_gradient_kernel");
	}
	if (mlp_done_event) NGP_HIP_TRY(hipEventRecord((hipEvent_t)mlp_done_event, st));
	// EGradientMode::Overwrite: every table entry is written exactly once (no memset, no global float atomics).  The weight-gradient partials are summed by extra
	// rows of the hash-grid backward's first launch.
	if (launch_grid_backward<3>(st, desc_dev, grid_coords ? grid_coords : coords, coord_stride_floats, n, (const h2*)dx_planes, gb_partials, (char*)scratch + scratch_off_fx_for(desc_host, n), (h2*)(grads + NGP_MLP_N_PARAMS), true, true,
	                            WgradJob{(const float*)partials, grid, (half_t*)grads, (uint32_t)NGP_MLP_N_PARAMS}, desc_host, live.n_live)) return -1;
	if (grid_gradients_event) NGP_HIP_TRY(hipEventRecord((hipEvent_t)grid_gradients_event, st));
	return 0;
}

// ---- renderer visualisation passes: input gradient (Normals), activation visualisation (EncodingVis / Slice)
static uint64_t ig_off_dl(uint32_t n) { return ngp_hip_nerf_backward_scratch_bytes(n); }
static uint64_t ig_off_x(uint32_t n) { return ig_off_dl(n) + (uint64_t)n * 4 * 2; }
static uint64_t ig_off_out(uint32_t n) { return ig_off_x(n) + (uint64_t)n * 32 * 2; }
static uint64_t ig_off_din(uint32_t n) { return ig_off_out(n) + (uint64_t)n * 4 * 2; }
uint64_t ngp_hip_nerf_input_gradient_scratch_bytes(uint32_t n) { return ig_off_din(n) + (uint64_t)n * 6 * 4; }

int ngp_hip_nerf_input_gradient(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, uint32_t dim, float* coords_inout,
                                uint32_t coord_stride_floats, uint32_t n, void* scratch, uint64_t scratch_bytes, const NgpNetVariant* variant) {
	if (variant_check(variant, "ngp_hip_nerf_input_gradient: at most 16 extra dims and 3 hidden colour layers")) return -1;
	if (variant_scalar(variant)) { set_last_error("ngp_hip_nerf_input_gradient: not built into the scalar checker kernels (NGP_NETX_SCALAR)", hipErrorNotSupported); return -1; }
	if (n == 0 || (n % 256) != 0) { set_last_error("ngp_hip_nerf_input_gradient: n must be a positive multiple of 256", hipErrorInvalidValue); return -1; }
	if (dim >= 4) { set_last_error("ngp_hip_nerf_input_gradient: dim must be 0..3 (the padded outputs 4..15 carry nothing)", hipErrorInvalidValue); return -1; }
	if (coord_stride_floats < 7) { set_last_error("ngp_hip_nerf_input_gradient: coords are NgpCoord records (>= 7 floats)", hipErrorInvalidValue); return -1; }
	if (scratch_bytes < ngp_hip_nerf_input_gradient_scratch_bytes(n)) { set_last_error("ngp_hip_nerf_input_gradient: scratch too small", hipErrorInvalidValue); return -1; }
	hipStream_t st = (hipStream_t)stream;
	const float backprop_scale = 128.0f;   // [tcnn] input_gradient's default: keeps the fp16 backward out of the denormals
	half_t* dl = (half_t*)((char*)scratch + ig_off_dl(n));
	uint16_t* x_saved = (uint16_t*)((char*)scratch + ig_off_x(n));
	uint16_t* out = (uint16_t*)((char*)scratch + ig_off_out(n));
	float* din = (float*)((char*)scratch + ig_off_din(n));
	hipLaunchKernelGGL(one_hot_dl_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, n, dim, backprop_scale, dl);
	NGP_LAUNCH_CHECK("one_hot_dl_kernel");
	if (ngp_hip_nerf_forward(stream, desc_dev, params, coords_inout, coord_stride_floats, n, out, 4, x_saved, variant)) return -1;
	// the fused backward kernel with the direction gradient, then dL/dpos through the hash encoding; no parameter gradients (EGradientMode::Ignore):
	// the per-workgroup weight-gradient partials land in scratch and are dropped, the hash-grid backward does not run
	float* partials = (float*)scratch;
	h2* dx_planes = (h2*)((char*)scratch + scratch_off_dx(n));
	const uint32_t n_quads = n / 128;
	const uint32_t grid = n_quads < FB_MAX_WORKGROUPS ? n_quads : FB_MAX_WORKGROUPS;
	(void)desc_host;
	if (variant_is_generic(variant)) {   // the same three steps on the variant's kernels: first launch only (the second one of a deep network only adds weight-gradient tiles)
		NxBwdArgs a{desc_dev, (const half_t*)params, (const float*)coords_inout, coord_stride_floats, n, (const half_t*)x_saved, (const half_t*)dl, 4u, dx_planes, partials,
		            (uint32_t*)((char*)scratch + scratch_off_fx(n)), (uint32_t)(sizeof(GbFxCounters) / 4), variant->extra_dims, variant->sample_slot, variant->n_extra_dims, nullptr, din};
		if (nx_bwd_dispatch(variant->n_rgb_hidden_layers, variant->n_extra_dims != 0, dim3(n_quads < 256u ? n_quads : 256u), st, a, true)) { set_last_error("ngp_hip_nerf_input_gradient: launch set-up failed", hipErrorInvalidValue); return -1; }
		NGP_LAUNCH_CHECK("nx_backward_kernel (input gradient)");
		hipLaunchKernelGGL(nerf_input_pos_gradient_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, desc_dev, (const half_t*)params, (const float*)coords_inout, coord_stride_floats, n, (const h2*)dx_planes, din, ngp_hip_net_mlp_params_host(variant));
		NGP_LAUNCH_CHECK("nerf_input_pos_gradient_kernel");
		hipLaunchKernelGGL(input_gradient_writeback_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, n, 1.0f / backprop_scale, (const float*)din, coords_inout, coord_stride_floats);
		NGP_LAUNCH_CHECK("input_gradient_writeback_kernel");
		return 0;
	}
	hipLaunchKernelGGL(nerf_backward_fused_kernel<true>, dim3(grid), dim3(256), 0, st, desc_dev, (const half_t*)params, (const float*)coords_inout, coord_stride_floats, n, (const half_t*)x_saved, (const half_t*)dl, 4u,
	                   dx_planes, partials, (uint32_t*)((char*)scratch + scratch_off_fx(n)), (uint32_t)(sizeof(GbFxCounters) / 4), din, LiveSamples{nullptr, nullptr, nullptr, nullptr});
	NGP_LAUNCH_CHECK("nerf_backward_fused_kernel (input gradient)");
	hipLaunchKernelGGL(nerf_input_pos_gradient_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, desc_dev, (const half_t*)params, (const float*)coords_inout, coord_stride_floats, n, (const h2*)dx_planes, din, (uint32_t)GRID_OFF);
	NGP_LAUNCH_CHECK("nerf_input_pos_gradient_kernel");
	hipLaunchKernelGGL(input_gradient_writeback_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, n, 1.0f / backprop_scale, (const float*)din, coords_inout, coord_stride_floats);
	NGP_LAUNCH_CHECK("input_gradient_writeback_kernel");
	return 0;
}

int ngp_hip_nerf_visualize_activation(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, uint32_t layer, uint32_t dimension, const float* coords,
                                      uint32_t coord_stride_floats, uint32_t n, float* out, uint32_t out_stride_floats, const NgpNetVariant* variant) {
	if (variant_check(variant, "ngp_hip_nerf_visualize_activation: at most 16 extra dims and 3 hidden colour layers")) return -1;
	if (variant_is_generic(variant)) {
		const GenLayout L = gen_layout(variant->n_extra_dims, variant->n_rgb_hidden_layers);
		const uint32_t n_layers = 3u + variant->n_rgb_hidden_layers;   // encoding, density hidden, colour input, colour hidden layers
		const uint32_t width = layer == 0 ? 32u : layer == 1 ? 64u : layer == 2 ? L.rgb_in : 64u;
		if (layer >= n_layers || dimension >= width) { set_last_error("ngp_hip_nerf_visualize_activation: layer / dimension outside the variant's activations (32, 64, colour input width, 64 per hidden colour layer)", hipErrorInvalidValue); return -1; }
		if (!n) return 0;
		hipLaunchKernelGGL(netx_visualize_activation_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, L, desc_dev, (const half_t*)params, layer, dimension, coords, coord_stride_floats, n, out, out_stride_floats,
		                   variant->extra_dims, variant->sample_slot);
		NGP_LAUNCH_CHECK("netx_visualize_activation_kernel");
		return 0;
	}
	static const uint32_t widths[5] = {32, 64, 32, 64, 64};   // NerfNetwork::width(layer) (nerf_network.h:474-484)
	if (layer >= 5 || dimension >= widths[layer]) { set_last_error("ngp_hip_nerf_visualize_activation: layer 0..4, dimension below the layer's width (32, 64, 32, 64, 64)", hipErrorInvalidValue); return -1; }
	if (!n) return 0;
	hipLaunchKernelGGL(nerf_visualize_activation_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, layer, dimension, coords, coord_stride_floats, n, out, out_stride_floats);
	NGP_LAUNCH_CHECK("nerf_visualize_activation_kernel");
	return 0;
}

// ---- hash-grid backward on its own (tcnn kernel_grid_backward): scratch = [partials 16 x 4 MiB][binned path]
uint64_t ngp_hip_grid_backward_scratch_bytes(uint32_t n) { return (uint64_t)16 * GB_PARTIAL_LEVEL_BYTES + gb_fx_bytes(n); }
static int grid_backward_entry(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                               const uint16_t* dL_dx_planes, uint16_t* grid_grad, void* scratch, uint64_t scratch_bytes, bool ordered) {
	if (n == 0 || (n % 256) != 0) { set_last_error("ngp_hip_grid_backward: n must be a positive multiple of 256", hipErrorInvalidValue); return -1; }
	if (n_dims != 2 && n_dims != 3) { set_last_error("ngp_hip_grid_backward: n_dims must be 2 or 3", hipErrorInvalidValue); return -1; }
	if (scratch_bytes < ngp_hip_grid_backward_scratch_bytes(n)) { set_last_error("ngp_hip_grid_backward: scratch too small", hipErrorInvalidValue); return -1; }
	hipStream_t st = (hipStream_t)stream;
	h2* gb_partials = (h2*)scratch;
	void* fx = (char*)scratch + (uint64_t)16 * GB_PARTIAL_LEVEL_BYTES;
	if (n_dims == 2) return launch_grid_backward<2>(st, desc_dev, pos, pos_stride_floats, n, (const h2*)dL_dx_planes, gb_partials, fx, (h2*)grid_grad, false, ordered);
	return launch_grid_backward<3>(st, desc_dev, pos, pos_stride_floats, n, (const h2*)dL_dx_planes, gb_partials, fx, (h2*)grid_grad, false, ordered);
}
int ngp_hip_grid_backward(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                          const uint16_t* dL_dx_planes, uint16_t* grid_grad, void* scratch, uint64_t scratch_bytes) {
	return grid_backward_entry(stream, n_dims, desc_dev, pos, pos_stride_floats, n, dL_dx_planes, grid_grad, scratch, scratch_bytes, true);
}
int ngp_hip_grid_backward_unordered(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                                    const uint16_t* dL_dx_planes, uint16_t* grid_grad, void* scratch, uint64_t scratch_bytes) {
	return grid_backward_entry(stream, n_dims, desc_dev, pos, pos_stride_floats, n, dL_dx_planes, grid_grad, scratch, scratch_bytes, false);
}

// ---- plumbing configs: grid encoding -> one MLP (P1 image, P2 sdf)
int ngp_hip_gridmlp_make_desc_host(uint32_t n_dims, uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, NgpNetDesc* d) {
	if (n_levels != 16 || !d || (n_dims != 2 && n_dims != 3)) { set_last_error("ngp_hip_gridmlp_make_desc_host: n_levels must be 16 and n_dims 2 or 3", hipErrorInvalidValue); return -1; }
	uint32_t offset = 0;
	const float log2_pls = log2f(per_level_scale);
	for (uint32_t l = 0; l < n_levels; ++l) {
		const float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
		const uint32_t res = (uint32_t)ceilf(scale) + 1u;
		uint64_t dense = 1;
		for (uint32_t k = 0; k < n_dims; ++k) dense *= res;
		const uint32_t max_params = 0xffffffffu / 2u;
		uint32_t cnt = dense > max_params ? max_params : (uint32_t)dense;
		cnt = (cnt + 7u) / 8u * 8u;
		const uint32_t hashmap = 1u << log2_hashmap_size;
		const uint32_t size = cnt < hashmap ? cnt : hashmap;
		d->levels[l].scale = scale; d->levels[l].resolution = res; d->levels[l].offset = offset; d->levels[l].size = size;
		// the hash-grid backward's owners cover at most 256 slices of 16384 entries per level (log2_hashmap_size <= 22); refused here rather than trained wrongly
		if (size > GB_FX_MAX_SLICES * GB_SLICE) { set_last_error("make_desc: a level above 2^22 entries (log2_hashmap_size > 22) is outside what the hash-grid backward covers", hipErrorInvalidValue); return -1; }
		offset += size;
	}
	d->n_levels = n_levels;
	d->n_grid_entries = offset;
	return 0;
}

uint32_t ngp_hip_gridmlp_n_params_host(const NgpNetDesc* d) { return NGP_GRIDMLP_N_PARAMS + 2u * d->n_grid_entries; }

int ngp_hip_gridmlp_init_params(void* stream, const NgpNetDesc* desc_host, uint64_t seed, float* master, uint16_t* params, uint16_t* inference_params) {
	uint64_t state = 0u, inc = (1ull << 1u) | 1u;
	state = state * 0x5851f42d4c957f2dULL + inc;
	state += seed;
	state = state * 0x5851f42d4c957f2dULL + inc;
	const uint32_t n = ngp_hip_gridmlp_n_params_host(desc_host);
	hipLaunchKernelGGL(gridmlp_init_params_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, state, inc, master, (half_t*)params, (half_t*)inference_params);
	NGP_LAUNCH_CHECK("gridmlp_init_params_kernel");
	return 0;
}

int ngp_hip_gridmlp_forward(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                            uint16_t* out, uint32_t out_stride, uint16_t* x_saved) {
	if (n == 0) return 0;
	if (out_stride < 4 || (out_stride & 3)) { set_last_error("ngp_hip_gridmlp_forward: out_stride must be a multiple of 4", hipErrorInvalidValue); return -1; }
	if (n_dims == 2) hipLaunchKernelGGL((gridmlp_forward_kernel<2, 0>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, pos, pos_stride_floats, n, (half_t*)out, out_stride, (half_t*)x_saved, (const h2*)nullptr, 0u);
	else if (n_dims == 3) hipLaunchKernelGGL((gridmlp_forward_kernel<3, 0>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, pos, pos_stride_floats, n, (half_t*)out, out_stride, (half_t*)x_saved, (const h2*)nullptr, 0u);
	else { set_last_error("ngp_hip_gridmlp_forward: n_dims must be 2 or 3", hipErrorInvalidValue); return -1; }
	NGP_LAUNCH_CHECK("gridmlp_forward_kernel");
	return 0;
}

// the two-kernel organisation of the same pass (the NeRF path's ngp_hip_nerf_forward_ws): XCD-affine encode into level planes, then the MLP kernel at twice the
// occupancy.  Same bits as ngp_hip_gridmlp_forward (tests/test_gridmlp_gpu.py); faster where the positions have no order (SDF batches), not where they run along
// a dense level's x axis (stratified image batches) — the host measures and chooses (Testbed::network_pass).
int ngp_hip_gridmlp_forward_ws(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                               uint16_t* out, uint32_t out_stride, uint16_t* x_saved, void* workspace, uint64_t workspace_bytes) {
	if (n == 0) return 0;
	if (out_stride < 4 || (out_stride & 3)) { set_last_error("ngp_hip_gridmlp_forward_ws: out_stride must be a multiple of 4", hipErrorInvalidValue); return -1; }
	if (n_dims != 2 && n_dims != 3) { set_last_error("ngp_hip_gridmlp_forward_ws: n_dims must be 2 or 3", hipErrorInvalidValue); return -1; }
	const h2* planes; uint32_t n_pad;
	if (launch_encode(stream, desc_dev, params, pos, pos_stride_floats, n, workspace, workspace_bytes, &planes, &n_pad, "ngp_hip_gridmlp_forward_ws: workspace too small (ngp_hip_nerf_encode_workspace_bytes)", nullptr, GM_GRID_OFF, n_dims)) return -1;
	if (n_dims == 2) hipLaunchKernelGGL((gridmlp_forward_kernel<2, 1>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, pos, pos_stride_floats, n, (half_t*)out, out_stride, (half_t*)x_saved, planes, n_pad);
	else hipLaunchKernelGGL((gridmlp_forward_kernel<3, 1>), dim3(fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, desc_dev, (const half_t*)params, pos, pos_stride_floats, n, (half_t*)out, out_stride, (half_t*)x_saved, planes, n_pad);
	NGP_LAUNCH_CHECK("gridmlp_forward_kernel<pre>");
	return 0;
}

// scratch layout: [weight-gradient partials 512 x 7168 fp32][dL/dx planes 16 x n half2][grid partials 16 x 4 MiB][binned path: counters, item lists, run sums]
static uint64_t gm_scratch_off_dx(uint32_t) { return (uint64_t)FB_MAX_WORKGROUPS * NGP_GRIDMLP_N_PARAMS * 4u; }
static uint64_t gm_scratch_off_gb(uint32_t n) { return gm_scratch_off_dx(n) + (uint64_t)16 * n * 4u; }
static uint64_t gm_scratch_off_fx(uint32_t n) { return gm_scratch_off_gb(n) + (uint64_t)16 * GB_PARTIAL_LEVEL_BYTES; }
uint64_t ngp_hip_gridmlp_backward_scratch_bytes(uint32_t n) { return gm_scratch_off_fx(n) + gb_fx_bytes(n); }

int ngp_hip_gridmlp_backward(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                             const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride, uint16_t* grads, void* scratch, uint64_t scratch_bytes) {
	if (n == 0 || (n % 256) != 0) { set_last_error("ngp_hip_gridmlp_backward: n must be a positive multiple of 256", hipErrorInvalidValue); return -1; }
	if (n_dims != 2 && n_dims != 3) { set_last_error("ngp_hip_gridmlp_backward: n_dims must be 2 or 3", hipErrorInvalidValue); return -1; }
	if (scratch_bytes < ngp_hip_gridmlp_backward_scratch_bytes(n)) { set_last_error("ngp_hip_gridmlp_backward: scratch too small", hipErrorInvalidValue); return -1; }
	hipStream_t st = (hipStream_t)stream;
	float* partials = (float*)scratch;
	h2* dx_planes = (h2*)((char*)scratch + gm_scratch_off_dx(n));
	h2* gb_partials = (h2*)((char*)scratch + gm_scratch_off_gb(n));
	const uint32_t n_quads = n / 128;
	const uint32_t grid = n_quads < FB_MAX_WORKGROUPS ? n_quads : FB_MAX_WORKGROUPS;
	hipLaunchKernelGGL(gridmlp_backward_fused_kernel, dim3(grid), dim3(256), 0, st, (const half_t*)params, n, (const half_t*)x_saved, (const half_t*)dL_dout, dl_stride, dx_planes, partials,
	                   (uint32_t*)((char*)scratch + gm_scratch_off_fx(n)), (uint32_t)(sizeof(GbFxCounters) / 4));
	NGP_LAUNCH_CHECK("gridmlp_backward_fused_kernel");
	hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(div_up(NGP_GRIDMLP_N_PARAMS, 64)), dim3(64 * WR_WAVES), 0, st, (const float*)partials, grid, (half_t*)grads, (uint32_t)NGP_GRIDMLP_N_PARAMS);
	NGP_LAUNCH_CHECK("wgrad_reduce_kernel");
	if (n_dims == 2) { if (launch_grid_backward<2>(st, desc_dev, pos, pos_stride_floats, n, (const h2*)dx_planes, gb_partials, (char*)scratch + gm_scratch_off_fx(n), (h2*)(grads + NGP_GRIDMLP_N_PARAMS), true, true)) return -1; }   // image fitting: the stratified batch (testbed_image.cu:220-291) runs along x — consecutive samples share coarse cells, the merging walk pays (0.25 vs 0.36 ms per step measured)
	else {   // SDF: samples in no spatial order — pair records (backward group 345 -> 232 us)
	 if (launch_grid_backward<3>(st, desc_dev, pos, pos_stride_floats, n, (const h2*)dx_planes, gb_partials, (char*)scratch + gm_scratch_off_fx(n), (h2*)(grads + NGP_GRIDMLP_N_PARAMS), true, false)) return -1; }
	return 0;
}

// fp16 <-> fp32 copies of the gradient vector around the reduce-scatter of the sharded optimizer step (the sum over the ranks is taken in fp32)
namespace ngp {
__global__ void __launch_bounds__(256) f16_to_f32_kernel(uint32_t n, uint32_t n_padded, const half_t* __restrict__ src, float* __restrict__ dst) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_padded) dst[i] = i < n ? (float)src[i] : 0.0f;
}
__global__ void __launch_bounds__(256) f32_to_f16_kernel(uint32_t n, const float* __restrict__ src, half_t* __restrict__ dst) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = (half_t)src[i];
}
}
int ngp_hip_f16_to_f32(void* stream, uint32_t n, uint32_t n_padded, const uint16_t* src, float* dst) {
	if (n_padded < n) { set_last_error("ngp_hip_f16_to_f32: n_padded < n", hipErrorInvalidValue); return -1; }
	if (!n_padded) return 0;
	hipLaunchKernelGGL(f16_to_f32_kernel, dim3(div_up(n_padded, 256)), dim3(256), 0, (hipStream_t)stream, n, n_padded, (const half_t*)src, dst);
	NGP_LAUNCH_CHECK("f16_to_f32_kernel");
	return 0;
}
int ngp_hip_f32_to_f16(void* stream, uint32_t n, const float* src, uint16_t* dst) {
	if (!n) return 0;
	hipLaunchKernelGGL(f32_to_f16_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, src, (half_t*)dst);
	NGP_LAUNCH_CHECK("f32_to_f16_kernel");
	return 0;
}

int ngp_hip_optimizer_step(void* stream, uint32_t n_params, uint32_t n_matrix_params, uint32_t step, float learning_rate, float beta1, float beta2,
                                  float epsilon, float l2_reg, float loss_scale, float ema_decay, const uint16_t* grads, float* master, uint16_t* params,
                                  float* first_moments, float* second_moments, float* ema, uint16_t* inference_params, uint32_t optimize_mask) {
	const float lr = learning_rate * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
	const float ema_debias_old = 1.0f - powf(ema_decay, (float)(step - 1));
	const float ema_debias_new = 1.0f / (1.0f - powf(ema_decay, (float)step));
	const uintptr_t align_or = (uintptr_t)grads | (uintptr_t)master | (uintptr_t)params | (uintptr_t)first_moments | (uintptr_t)second_moments | (uintptr_t)ema | (uintptr_t)inference_params;
	const uint32_t n_vec = (align_or & 15u) ? 0u : (n_params & ~3u);   // 16-byte aligned arrays: groups of four, then a scalar tail
	if (n_vec) {
		hipLaunchKernelGGL(adam_ema_vec4_kernel, dim3(div_up(n_vec / 4, 256)), dim3(256), 0, (hipStream_t)stream, n_vec / 4, n_matrix_params, lr, beta1, beta2, epsilon, l2_reg,
		                   loss_scale, ema_decay, ema_debias_old, ema_debias_new, optimize_mask, (const half4_t*)grads, (float4*)master, (half4_t*)params, (float4*)first_moments, (float4*)second_moments,
		                   (float4*)ema, (half4_t*)inference_params);
		NGP_LAUNCH_CHECK("adam_ema_vec4_kernel");
	}
	if (n_vec < n_params) {
		const uint32_t r = n_params - n_vec, nm = n_matrix_params > n_vec ? n_matrix_params - n_vec : 0u;
		hipLaunchKernelGGL(adam_ema_kernel, dim3(div_up(r, 256)), dim3(256), 0, (hipStream_t)stream, r, nm, lr, beta1, beta2, epsilon, l2_reg,
		                   loss_scale, ema_decay, ema_debias_old, ema_debias_new, optimize_mask, (const half_t*)grads + n_vec, master + n_vec, (half_t*)params + n_vec, first_moments + n_vec,
		                   second_moments + n_vec, ema + n_vec, (half_t*)inference_params + n_vec);
		NGP_LAUNCH_CHECK("adam_ema_kernel");
	}
	return 0;
}

} // extern "C"
