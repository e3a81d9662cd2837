// network_netx_mfma.cuh — the NerfNetwork VARIANTS on the matrix cores (included by network.hip inside namespace ngp, after network_generic.cuh):
//   * per-image extra dims behind the direction encoding (latent codes, light directions): the colour network's first layer takes one more K-block
//     (nerf_network.h:81-98, 452-460: [density out 16 | SH 16 | extra dims, padded to 16]);
//   * rgb_network.n_hidden_layers 0 / 1 / 3 (configs/nerf/base_{0,1,3}layer.json; src/testbed.cu:2337-2363).
// Same scheme as the base family's kernels in network.hip — one wave owns 32 samples end to end, the f32 D tile of a layer packed in registers IS the B operand of the next
// one, weights in LDS pre-permuted into A-operand tiles, the hash gather feeding the first MFMA from registers; backward = recompute + dgrad chain + weight gradients in one
// kernel, operands transposed through LDS one layer at a time — with the layer list a template parameter instead of written out.  The 32 x 32 output
// tiles of the weight gradients are dealt round-robin to the workgroup's 4 waves, three accumulators per wave; networks with more than 12 tiles (three hidden colour
// layers, or two + extra dims) take a second launch of the same kernel for the remaining tiles (the chain is recomputed, nothing else is written twice).
// The scalar kernels of network_generic.cuh stay as the bit-compatible checker (NGP_NETX_SCALAR in NgpNetVariant::flags): fp32 sums in the oracle's order there,
// MFMA accumulation order here — compared at the base kernels' tolerances (tests/test_netx_gpu.py).
#pragma once

constexpr int MAP_XTRA = 4;   // K-slot -> feature map of the extra-dims K-block: lane (j, g) holds extra dims 8g .. 8g+7 of its sample (slot_feature: 32 + 8 g + e)

template <int NH, int XK>
struct NxL {
	static_assert(NH >= 0 && NH <= 3 && (XK == 0 || XK == 1), "0..3 hidden colour layers, extra dims or not");
	static constexpr int RIN = 32 + 16 * XK;                 // colour network input: density out 16 | SH 16 | extra 16
	static constexpr int C0_OUT = NH == 0 ? 16 : 64;
	static constexpr int C0_MT = NH == 0 ? 1 : 2, C0_KB = 2 + XK;
	// parameters (gen_layout's order)
	static constexpr int OFF_W1 = 0, OFF_W2 = 64 * 32, OFF_C0 = OFF_W2 + 16 * 64;
	static constexpr int OFF_CH1 = OFF_C0 + C0_OUT * RIN;     // hidden matrix m = 1 .. NH-1 at OFF_CH1 + (m - 1) * 4096
	static constexpr int OFF_OUT = OFF_CH1 + (NH > 1 ? (NH - 1) * 4096 : 0);
	static constexpr int N_MLP = NH == 0 ? OFF_CH1 : OFF_OUT + 16 * 64;
	// forward tiles
	static constexpr int T_W1 = 0, T_W2 = 4, T_C0 = 8;        // C0: mt * C0_KB + kb
	static constexpr int T_CH1 = T_C0 + C0_MT * C0_KB;        // hidden m: + (m - 1) * 8 + mt * 4 + kb
	static constexpr int T_OUT = T_CH1 + (NH > 1 ? (NH - 1) * 8 : 0);
	static constexpr int N_FWD = NH == 0 ? T_CH1 : T_OUT + 4;
	// backward (transposed) tiles
	static constexpr int T_OUTT = N_FWD;                      // + mt (2): A[i = last hidden feat][slot = channel]            (NH >= 1)
	static constexpr int T_CH1T = T_OUTT + (NH >= 1 ? 2 : 0); // hidden m: + (m - 1) * 8 + mt * 4 + kb: A[i = in feat][slot = out feat]
	static constexpr int C0T_MT = 1 + XK, C0T_KB = NH == 0 ? 1 : 4;
	static constexpr int T_C0T = T_CH1T + (NH > 1 ? (NH - 1) * 8 : 0);   // + mt * C0T_KB + kb: A[i = colour input feat][slot = C0 out feat]
	static constexpr int T_W2T = T_C0T + C0T_MT * C0T_KB;     // + mt (2)
	static constexpr int T_W1T = T_W2T + 2;                   // + kb (4)
	static constexpr int N_ALL = T_W1T + 4;
	// weight-gradient tiles in the order the backward kernel meets the layers: OUT, CH(NH-1) .. CH(1), C0, W2, W1
	static constexpr int G_OUT = 0, G_CHLAST = NH >= 1 ? 2 : 0;             // hidden m at G_CHLAST + (NH - 1 - m) * 4
	static constexpr int G_C0 = G_CHLAST + (NH > 1 ? (NH - 1) * 4 : 0);
	static constexpr int C0_NT = 1 + XK;
	static constexpr int G_W2 = G_C0 + C0_MT * C0_NT, G_W1 = G_W2 + 2, N_GT = G_W1 + 2;
	static constexpr int N_PASS = (N_GT + 11) / 12;
};

// one A-operand tile of the variant's parameter block (runtime tile id; staging only)
template <int NH, int XK>
__device__ __forceinline__ h8 nx_gather_tile(const half_t* P, int tile, int lane) {
	typedef NxL<NH, XK> L;
	if (tile < L::T_W2) return gather_tile_spec(P, L::OFF_W1, 64, 32, (tile - L::T_W1) >> 1, (tile - L::T_W1) & 1, MAP_ENC, false, lane);
	if (tile < L::T_C0) return gather_tile_spec(P, L::OFF_W2, 16, 64, 0, tile - L::T_W2, MAP_HID, false, lane);
	if (tile < L::T_CH1) {
		const int q = tile - L::T_C0, mt = q / L::C0_KB, kb = q % L::C0_KB;
		return gather_tile_spec(P, L::OFF_C0, L::C0_OUT, L::RIN, mt, kb, kb == 2 ? MAP_XTRA : MAP_RGBIN, false, lane);
	}
	if (tile < L::T_OUT) { const int q = tile - L::T_CH1; return gather_tile_spec(P, L::OFF_CH1 + (q >> 3) * 4096, 64, 64, (q >> 2) & 1, q & 3, MAP_HID, false, lane); }
	if (tile < L::N_FWD) return gather_tile_spec(P, L::OFF_OUT, 16, 64, 0, tile - L::T_OUT, MAP_HID, false, lane);
	if (tile < L::T_CH1T) return gather_tile_spec(P, L::OFF_OUT, 16, 64, tile - L::T_OUTT, 0, MAP_CH, true, lane);
	if (tile < L::T_C0T) { const int q = tile - L::T_CH1T; return gather_tile_spec(P, L::OFF_CH1 + (q >> 3) * 4096, 64, 64, (q >> 2) & 1, q & 3, MAP_HID, true, lane); }
	if (tile < L::T_W2T) { const int q = tile - L::T_C0T; return gather_tile_spec(P, L::OFF_C0, L::C0_OUT, L::RIN, q / L::C0T_KB, q % L::C0T_KB, NH == 0 ? MAP_CH : MAP_HID, true, lane); }
	if (tile < L::T_W1T) return gather_tile_spec(P, L::OFF_W2, 16, 64, tile - L::T_W2T, 0, MAP_RGBIN, true, lane);
	return gather_tile_spec(P, L::OFF_W1, 64, 32, 0, tile - L::T_W1T, MAP_HID, true, lane);
}

// activations of one 32-sample tile in B-operand form, kept for the backward recompute
template <int NH, int XK> struct NxActs { h8 h1[4]; h8 rin[2 + XK]; h8 hc[NH > 0 ? NH : 1][4]; };

// the extra-dims K-block of lane (j, g): [tcnn] Identity encoding = the input cast to the network precision, rows beyond n_extra are padding
__device__ __forceinline__ h8 nx_extra_block(const float* __restrict__ extra_dims, const uint32_t* __restrict__ sample_slot, uint32_t n_extra, uint32_t s, int g) {
	h8 r = {};
	if (!extra_dims) return r;
	const float* row = extra_dims + (size_t)(sample_slot ? sample_slot[s] : 0u) * n_extra;
#pragma unroll
	for (int e = 0; e < 8; ++e) if ((uint32_t)(8 * g + e) < n_extra) r[e] = (half_t)row[8 * g + e];
	return r;
}

template <int NH, int XK, bool KEEP>
__device__ __forceinline__ void nx_mlp_forward(const h8* __restrict__ lt, int lane, const h8& x0, const h8& x1, const h8& sh, const h8& xt, f32x16& dd, f32x16& oo, NxActs<NH, XK>* acts) {
	typedef NxL<NH, XK> L;
	const f32x16 zero = {};
	f32x16 a0 = NGP_MFMA(lt[(L::T_W1 + 0) * 64 + lane], x0, zero);
	a0 = NGP_MFMA(lt[(L::T_W1 + 1) * 64 + lane], x1, a0);
	f32x16 a1 = NGP_MFMA(lt[(L::T_W1 + 2) * 64 + lane], x0, zero);
	a1 = NGP_MFMA(lt[(L::T_W1 + 3) * 64 + lane], x1, a1);
	h8 h[4];
	d_to_b<true>(a0, h[0], h[1]);
	d_to_b<true>(a1, h[2], h[3]);
	if (KEEP) { for (int k = 0; k < 4; ++k) acts->h1[k] = h[k]; }
	dd = zero;
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) dd = NGP_MFMA(lt[(L::T_W2 + kb) * 64 + lane], h[kb], dd);
	h8 r0, junk;
	d_to_b<false>(dd, r0, junk);
	if (KEEP) { acts->rin[0] = r0; acts->rin[1] = sh; if (XK) acts->rin[1 + XK] = xt; }
	if (NH == 0) {   // one linear map [16][RIN]: rows 0..2 are the colour
		oo = NGP_MFMA(lt[(L::T_C0 + 0) * 64 + lane], r0, zero);
		oo = NGP_MFMA(lt[(L::T_C0 + 1) * 64 + lane], sh, oo);
		if (XK) oo = NGP_MFMA(lt[(L::T_C0 + 2) * 64 + lane], xt, oo);
		return;
	}
	a0 = NGP_MFMA(lt[(L::T_C0 + 0) * 64 + lane], r0, zero);
	a0 = NGP_MFMA(lt[(L::T_C0 + 1) * 64 + lane], sh, a0);
	if (XK) a0 = NGP_MFMA(lt[(L::T_C0 + 2) * 64 + lane], xt, a0);
	a1 = NGP_MFMA(lt[(L::T_C0 + L::C0_KB + 0) * 64 + lane], r0, zero);
	a1 = NGP_MFMA(lt[(L::T_C0 + L::C0_KB + 1) * 64 + lane], sh, a1);
	if (XK) a1 = NGP_MFMA(lt[(L::T_C0 + L::C0_KB + 2) * 64 + lane], xt, a1);
	d_to_b<true>(a0, h[0], h[1]);
	d_to_b<true>(a1, h[2], h[3]);
	if (KEEP) { for (int k = 0; k < 4; ++k) acts->hc[0][k] = h[k]; }
#pragma unroll
	for (int m = 1; m < NH; ++m) {
		a0 = zero; a1 = zero;
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) {
			a0 = NGP_MFMA(lt[(L::T_CH1 + (m - 1) * 8 + kb) * 64 + lane], h[kb], a0);
			a1 = NGP_MFMA(lt[(L::T_CH1 + (m - 1) * 8 + 4 + kb) * 64 + lane], h[kb], a1);
		}
		d_to_b<true>(a0, h[0], h[1]);
		d_to_b<true>(a1, h[2], h[3]);
		if (KEEP) { for (int k = 0; k < 4; ++k) acts->hc[m][k] = h[k]; }
	}
	if (KEEP) return;
	oo = zero;
#pragma unroll
	for (int kb = 0; kb < 4; ++kb) oo = NGP_MFMA(lt[(L::T_OUT + kb) * 64 + lane], h[kb], oo);
}

// MODE 0 inference (rgb sigma), 1 density only (the density network is the base family's; only the grid sits behind a different number of MLP parameters: any NH / XK
// instantiation serves), 2 training forward (also stores the encoded features).  PRE = 1: the features come from encode_planes_kernel (x_planes[level][n_pad]).
template <int MODE, int PRE, int NH, int XK>
__global__ void __launch_bounds__(256, 2) nx_forward_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ coords, uint32_t coord_stride, uint32_t n,
                                                            half_t* __restrict__ out, uint32_t out_stride, half_t* __restrict__ x_saved, const h2* __restrict__ x_planes, uint32_t n_pad,
                                                            const float* __restrict__ extra_dims, const uint32_t* __restrict__ sample_slot, uint32_t n_extra, uint32_t grid_off) {
	typedef NxL<NH, XK> L;
	__shared__ __attribute__((aligned(16))) h8 lds_tiles[(MODE == 1 ? 8 : L::N_FWD) * 64];
	stage_tiles<(MODE == 1 ? L::OFF_C0 : L::N_MLP), (MODE == 1 ? 8 : L::N_FWD)>(lds_tiles, params, nx_gather_tile<NH, XK>);
	const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
	const uint32_t n_tiles = (n + 31) / 32;
	const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
	const h2* __restrict__ grid = (const h2*)(params + grid_off);
	for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
		const uint32_t s = tile * 32 + j;
		const bool valid = s < n;
		const uint32_t sc = valid ? s : 0;
		const float* c = coords + (size_t)sc * coord_stride;
		h8 x0, x1;
		if (PRE == 1) {
			const h2* xp = x_planes + (size_t)(8 * g) * n_pad + sc;
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
		f32x16 dd, oo;
		uint32_t lt_off = 0;
		asm volatile("" : "+s"(lt_off)); // keep the LDS weight reads inside the loop
		if (MODE == 1) {   // (position-only records: nothing behind c[2] is read)
			const h8 none = {};
			mlp_forward<true, false>(lds_tiles + lt_off, lane, x0, x1, none, dd, oo, nullptr);   // tiles 0..7 are W1, W2 in either numbering
			if (valid && g == 0) out[s] = (half_t)dd[0];
			continue;
		}
		const h8 sh = sh4_half(g, c[4], c[5], c[6]);
		h8 xt = {};
		if (XK) xt = nx_extra_block(extra_dims, sample_slot, n_extra, sc, g);
		nx_mlp_forward<NH, XK, false>(lds_tiles + lt_off, lane, x0, x1, sh, xt, dd, oo, nullptr);
		if (valid && g == 0) {
			typedef _Float16 h4 __attribute__((ext_vector_type(4)));
			h4 o; o[0] = (half_t)oo[0]; o[1] = (half_t)oo[1]; o[2] = (half_t)oo[2]; o[3] = (half_t)dd[0];
			*(h4*)(out + (size_t)s * out_stride) = o;
		}
	}
}

// ---- backward -----------------------------------------------------------------------------------------------------------------------------------------------
// weight-gradient tile job: dY rows dy_row0 + mt * 32 .. (rows at or beyond dy_valid are zero), H rows h_row0 + nt * 32 .., all 8 K-blocks of the 128 staged samples
__device__ __forceinline__ void nx_job(const char* __restrict__ stage, int dy_row0, int dy_valid, int h_row0, int mt, int nt, int lane, f32x16& acc) {
	const int r32 = lane & 31, g = lane >> 5;
	const h8 zero = {};
	const bool a_ok = mt * 32 + r32 < dy_valid;
#pragma unroll
	for (int kbs = 0; kbs < 8; ++kbs) {
		h8 a = fb_get(stage, dy_row0 + (a_ok ? mt * 32 + r32 : 0), kbs, g);
		if (!a_ok) a = zero;
		const h8 b = fb_get(stage, h_row0 + nt * 32 + r32, kbs, g);
		acc = NGP_MFMA(a, b, acc);
	}
}
// the jobs of one layer: its N_MT x N_NT tiles, global ids G0 .. — tile t belongs to wave t % 4, accumulator (t / 4) % 3 of launch (t / 4) / 3
template <int G0, int N_MT, int N_NT, int PASS>
__device__ __forceinline__ void nx_layer_jobs(const char* __restrict__ stage, int dy_valid, int w, int lane, f32x16 (&acc)[3]) {
#pragma unroll
	for (int mt = 0; mt < N_MT; ++mt) {
#pragma unroll
		for (int nt = 0; nt < N_NT; ++nt) {
			const int t = G0 + mt * N_NT + nt;
			if ((t / 4) / 3 == PASS && w == (t & 3)) nx_job(stage, 0, dy_valid, 64, mt, nt, lane, acc[(t / 4) % 3]);
		}
	}
}
// where weight-gradient tile t lands in the parameter block
template <int NH, int XK>
__device__ __forceinline__ bool nx_tile_dst(int t, int& off, int& n_out, int& n_in, int& mt, int& nt) {
	typedef NxL<NH, XK> L;
	if (t >= L::N_GT) return false;
	if (NH >= 1 && t < L::G_CHLAST) { off = L::OFF_OUT; n_out = 16; n_in = 64; mt = 0; nt = t - L::G_OUT; return true; }
	if (t < L::G_C0) { const int q = t - L::G_CHLAST, m = NH - 1 - (q >> 2); off = L::OFF_CH1 + (m - 1) * 4096; n_out = 64; n_in = 64; mt = (q >> 1) & 1; nt = q & 1; return true; }
	if (t < L::G_W2) { const int q = t - L::G_C0; off = L::OFF_C0; n_out = L::C0_OUT; n_in = L::RIN; mt = q / L::C0_NT; nt = q % L::C0_NT; return true; }
	if (t < L::G_W1) { off = L::OFF_W2; n_out = 16; n_in = 64; mt = 0; nt = t - L::G_W2; return true; }
	off = L::OFF_W1; n_out = 64; n_in = 32; mt = t - L::G_W1; nt = 0;
	return true;
}

template <int NH, int XK, int PASS>
__global__ void __launch_bounds__(256) nx_backward_kernel(const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ coords, uint32_t coord_stride, uint32_t n,
                                                          const half_t* __restrict__ x_saved, const half_t* __restrict__ dL_dout, uint32_t dl_stride, h2* __restrict__ dx_planes,
                                                          float* __restrict__ partials /* [gridDim.x][N_MLP] */, uint32_t* __restrict__ zero_words, uint32_t n_zero_words,
                                                          const float* __restrict__ extra_dims, const uint32_t* __restrict__ sample_slot, uint32_t n_extra, float* __restrict__ dL_dextra,
                                                          float* __restrict__ dL_dinput /* [n][6] or NULL */) {
	typedef NxL<NH, XK> L;
	NGP_RAISE_CHAIN_PRIORITY();
	extern __shared__ __attribute__((aligned(16))) char nx_smem[];
	h8* lds_tiles = (h8*)nx_smem;
	char* stage = nx_smem + (size_t)L::N_ALL * 1024;
	if (PASS == 0) for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_zero_words; k += gridDim.x * blockDim.x) zero_words[k] = 0u;   // the hash-grid backward's counters
	stage_tiles<L::N_MLP, L::N_ALL>(lds_tiles, params, nx_gather_tile<NH, XK>);

	const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t n_quads = n / 128;
	const f32x16 zero = {};
	f32x16 acc[3] = {zero, zero, zero};
	const int col = w * 32 + j;

	for (uint32_t quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
		const uint32_t s = (quad * 4 + w) * 32 + j;
		const float* c = coords + (size_t)s * coord_stride;
		const h8* xs = (const h8*)(x_saved + (size_t)s * 32 + 16 * g);
		const h8 x0 = xs[0], x1 = xs[1];
		const h8 sh = sh4_half(g, c[4], c[5], c[6]);
		h8 xt = {};
		if (XK) xt = nx_extra_block(extra_dims, sample_slot, n_extra, s, g);
		NxActs<NH, XK> a;
		f32x16 dd, oo;
		uint32_t lt_off = 0;
		asm volatile("" : "+s"(lt_off));
		const h8* lt = lds_tiles + lt_off;
		nx_mlp_forward<NH, XK, true>(lt, lane, x0, x1, sh, xt, dd, oo, &a);

		const half_t* dl = dL_dout + (size_t)s * dl_stride;
		h8 dout = {};
		if (g == 0) { dout[0] = dl[0]; dout[1] = dl[1]; dout[2] = dl[2]; }
		const half_t dsigma = dl[3];
		h8 dh[4];
		f32x16 t0, t1;

		if constexpr (NH >= 1) {
			// ---- output layer [16][64]: dY = dout (16 rows), H = the last hidden activation
			fb_put(stage, 0, MAP_CH, 0, g, col, dout);
#pragma unroll
			for (int kb = 0; kb < 4; ++kb) fb_put(stage, 64, MAP_HID, kb, g, col, a.hc[NH - 1][kb]);
			__syncthreads();
			nx_layer_jobs<L::G_OUT, 1, 2, PASS>(stage, 16, w, lane, acc);
			t0 = NGP_MFMA(lt[(L::T_OUTT + 0) * 64 + lane], dout, zero);
			t1 = NGP_MFMA(lt[(L::T_OUTT + 1) * 64 + lane], dout, zero);
			dh[0] = mask_delta(t0, 0, a.hc[NH - 1][0]); dh[1] = mask_delta(t0, 1, a.hc[NH - 1][1]);
			dh[2] = mask_delta(t1, 0, a.hc[NH - 1][2]); dh[3] = mask_delta(t1, 1, a.hc[NH - 1][3]);
			__syncthreads();
			// ---- hidden layers m = NH-1 .. 1: dY = d(hc[m]), H = hc[m-1]
#pragma unroll
			for (int m = NH - 1; m >= 1; --m) {
#pragma unroll
				for (int kb = 0; kb < 4; ++kb) { fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]); fb_put(stage, 64, MAP_HID, kb, g, col, a.hc[m - 1][kb]); }
				__syncthreads();
				if (m == NH - 1) nx_layer_jobs<L::G_CHLAST, 2, 2, PASS>(stage, 64, w, lane, acc);
				else nx_layer_jobs<L::G_CHLAST + 4, 2, 2, PASS>(stage, 64, w, lane, acc);   // (NH = 3, m = 1)
				t0 = zero; t1 = zero;
#pragma unroll
				for (int kb = 0; kb < 4; ++kb) {
					t0 = NGP_MFMA(lt[(L::T_CH1T + (m - 1) * 8 + kb) * 64 + lane], dh[kb], t0);
					t1 = NGP_MFMA(lt[(L::T_CH1T + (m - 1) * 8 + 4 + kb) * 64 + lane], dh[kb], t1);
				}
				dh[0] = mask_delta(t0, 0, a.hc[m - 1][0]); dh[1] = mask_delta(t0, 1, a.hc[m - 1][1]);
				dh[2] = mask_delta(t1, 0, a.hc[m - 1][2]); dh[3] = mask_delta(t1, 1, a.hc[m - 1][3]);
				__syncthreads();
			}
			// ---- first colour layer [64][RIN]: dY = d(hc[0]), H = [density out | SH | extra]
#pragma unroll
			for (int kb = 0; kb < 4; ++kb) fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]);
		} else {
			// ---- the single colour matrix [16][RIN]: dY = dout
			fb_put(stage, 0, MAP_CH, 0, g, col, dout);
		}
		fb_put(stage, 64, MAP_RGBIN, 0, g, col, a.rin[0]);
		fb_put(stage, 64, MAP_RGBIN, 1, g, col, a.rin[1]);
		if (XK) fb_put(stage, 64, MAP_XTRA, 2, g, col, a.rin[1 + XK]);
		__syncthreads();
		nx_layer_jobs<L::G_C0, L::C0_MT, L::C0_NT, PASS>(stage, L::C0_OUT, w, lane, acc);
		// d_in = C0^T dY: rows 0..15 the density-net output gradient, 16..31 dL/d(SH), 32..47 dL/d(extra dims)
		t0 = zero; t1 = zero;
		if constexpr (NH >= 1) {
#pragma unroll
			for (int kb = 0; kb < 4; ++kb) {
				t0 = NGP_MFMA(lt[(L::T_C0T + kb) * 64 + lane], dh[kb], t0);
				if (XK) t1 = NGP_MFMA(lt[(L::T_C0T + L::C0T_KB + kb) * 64 + lane], dh[kb], t1);
			}
		} else {
			t0 = NGP_MFMA(lt[(L::T_C0T + 0) * 64 + lane], dout, zero);
			if (XK) t1 = NGP_MFMA(lt[(L::T_C0T + 1) * 64 + lane], dout, zero);
		}
		h8 dden;
#pragma unroll
		for (int e = 0; e < 8; ++e) dden[e] = (half_t)t0[e];
		if (g == 0) dden[0] = (half_t)((float)dden[0] + (float)dsigma); // add_density_gradient (nerf_network.h:63-74): fp16 += fp16
		if (PASS == 0 && XK && dL_dextra) {   // [tcnn] Identity backward: the fp16 rows 32.. of the colour network's dL_dinput (testbed_nerf.cu:1741)
#pragma unroll
			for (int r = 0; r < 8; ++r) {
				const uint32_t row = (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * g);
				if (row < n_extra) dL_dextra[(size_t)s * n_extra + row] = (float)(half_t)t1[r];
			}
		}
		if (PASS == 0 && dL_dinput) {   // dL/d(direction) through the SH basis, as nerf_backward_fused_kernel<true>
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
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 64, MAP_HID, kb, g, col, a.h1[kb]);
		__syncthreads();
		nx_layer_jobs<L::G_W2, 1, 2, PASS>(stage, 16, w, lane, acc);
		t0 = NGP_MFMA(lt[(L::T_W2T + 0) * 64 + lane], dden, zero);
		t1 = NGP_MFMA(lt[(L::T_W2T + 1) * 64 + lane], dden, zero);
		dh[0] = mask_delta(t0, 0, a.h1[0]); dh[1] = mask_delta(t0, 1, a.h1[1]);
		dh[2] = mask_delta(t1, 0, a.h1[2]); dh[3] = mask_delta(t1, 1, a.h1[3]);
		__syncthreads();

		// ---- W1: dY = d_h1, H = x
#pragma unroll
		for (int kb = 0; kb < 4; ++kb) fb_put(stage, 0, MAP_HID, kb, g, col, dh[kb]);
		fb_put(stage, 64, MAP_ENC, 0, g, col, x0);
		fb_put(stage, 64, MAP_ENC, 1, g, col, x1);
		__syncthreads();
		nx_layer_jobs<L::G_W1, 2, 1, PASS>(stage, 64, w, lane, acc);
		if (PASS == 0) {
			t0 = zero;
#pragma unroll
			for (int kb = 0; kb < 4; ++kb) t0 = NGP_MFMA(lt[(L::T_W1T + kb) * 64 + lane], dh[kb], t0);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int lvl = 4 * q + 2 * g;
				h2 u, v;
				u[0] = (half_t)t0[4 * q + 0]; u[1] = (half_t)t0[4 * q + 1];
				v[0] = (half_t)t0[4 * q + 2]; v[1] = (half_t)t0[4 * q + 3];
				dx_planes[(size_t)lvl * n + s] = u;
				dx_planes[(size_t)(lvl + 1) * n + s] = v;
			}
		}
		__syncthreads();
	}
	// ---- this workgroup's partial weight gradients: accumulator q of wave w is tile (PASS * 3 + q) * 4 + w
	float* __restrict__ dst = partials + (size_t)blockIdx.x * L::N_MLP;
	float* red = (float*)stage;   // 4 x 16 x 64 floats = 16 KiB
#pragma unroll
	for (int q = 0; q < 3; ++q) {
#pragma unroll
		for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[q][r];
		__syncthreads();
		for (int idx = threadIdx.x; idx < 4 * 16 * 64; idx += 256) {
			const int tw = idx >> 10, r = (idx >> 6) & 15, l = idx & 63;
			int off, n_out, n_in, mt, nt;
			if (!nx_tile_dst<NH, XK>((PASS * 3 + q) * 4 + tw, off, n_out, n_in, mt, nt)) continue;
			const int o = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), i = nt * 32 + (l & 31);
			if (o < n_out && i < n_in) dst[off + o * n_in + i] = red[idx];
		}
		__syncthreads();
	}
}
template <int NH, int XK> constexpr size_t nx_bwd_smem() { return (size_t)NxL<NH, XK>::N_ALL * 1024 + FB_STAGE_BYTES; }

// ---- host-side launchers (templates: C++ linkage, so they live here rather than inside network.hip's extern "C" block)
struct NxFwdArgs { const NgpNetDesc* desc; const half_t* params; const float* coords; uint32_t stride, n; half_t* out; uint32_t out_stride; half_t* x_saved; const h2* planes; uint32_t n_pad;
                   const float* extra_dims; const uint32_t* sample_slot; uint32_t n_extra, grid_off; };
template <int MODE, int PRE, int NH, int XK>
static void nx_fwd_launch(dim3 grid, hipStream_t st, const NxFwdArgs& a) {
	nx_forward_kernel<MODE, PRE, NH, XK><<<grid, dim3(256), 0, st>>>(a.desc, a.params, a.coords, a.stride, a.n, a.out, a.out_stride, a.x_saved, a.planes, a.n_pad, a.extra_dims, a.sample_slot, a.n_extra, a.grid_off);
}
template <int MODE, int PRE>
static int nx_fwd_dispatch(uint32_t n_hidden, bool extra, dim3 grid, hipStream_t st, const NxFwdArgs& a) {
	switch (n_hidden * 2u + (extra ? 1u : 0u)) {
		case 0: nx_fwd_launch<MODE, PRE, 0, 0>(grid, st, a); return 0;
		case 1: nx_fwd_launch<MODE, PRE, 0, 1>(grid, st, a); return 0;
		case 2: nx_fwd_launch<MODE, PRE, 1, 0>(grid, st, a); return 0;
		case 3: nx_fwd_launch<MODE, PRE, 1, 1>(grid, st, a); return 0;
		case 5: nx_fwd_launch<MODE, PRE, 2, 1>(grid, st, a); return 0;
		case 6: nx_fwd_launch<MODE, PRE, 3, 0>(grid, st, a); return 0;
		case 7: nx_fwd_launch<MODE, PRE, 3, 1>(grid, st, a); return 0;
		default: return -1;   // (2 hidden layers without extra dims is the base family: network.hip's own kernels)
	}
}

struct NxBwdArgs { const NgpNetDesc* desc; const half_t* params; const float* coords; uint32_t stride, n; const half_t* x_saved; const half_t* dL_dout; uint32_t dl_stride; h2* dx_planes; float* partials;
                   uint32_t* zero_words; uint32_t n_zero_words; const float* extra_dims; const uint32_t* sample_slot; uint32_t n_extra; float* dL_dextra; float* dL_dinput; };
template <int NH, int XK, int PASS>
static int nx_bwd_launch(dim3 grid, hipStream_t st, const NxBwdArgs& a) {
	static bool attr_set = false;
	if (!attr_set) {
		if (hipFuncSetAttribute((const void*)nx_backward_kernel<NH, XK, PASS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nx_bwd_smem<NH, XK>()) != hipSuccess) return -1;
		attr_set = true;
	}
	nx_backward_kernel<NH, XK, PASS><<<grid, dim3(256), nx_bwd_smem<NH, XK>(), st>>>(a.desc, a.params, a.coords, a.stride, a.n, a.x_saved, a.dL_dout, a.dl_stride, a.dx_planes, a.partials, a.zero_words, a.n_zero_words,
	                                                                                a.extra_dims, a.sample_slot, a.n_extra, a.dL_dextra, a.dL_dinput);
	return 0;
}
template <int NH, int XK>
static int nx_bwd_passes(dim3 grid, hipStream_t st, const NxBwdArgs& a, bool first_only) {
	if (nx_bwd_launch<NH, XK, 0>(grid, st, a)) return -1;
	if constexpr (NxL<NH, XK>::N_PASS > 1) { if (!first_only && nx_bwd_launch<NH, XK, 1>(grid, st, a)) return -1; }
	static_assert(NxL<NH, XK>::N_PASS <= 2, "two launches cover 24 weight-gradient tiles");
	return 0;
}
static int nx_bwd_dispatch(uint32_t n_hidden, bool extra, dim3 grid, hipStream_t st, const NxBwdArgs& a, bool first_only = false) {
	switch (n_hidden * 2u + (extra ? 1u : 0u)) {
		case 0: return nx_bwd_passes<0, 0>(grid, st, a, first_only);
		case 1: return nx_bwd_passes<0, 1>(grid, st, a, first_only);
		case 2: return nx_bwd_passes<1, 0>(grid, st, a, first_only);
		case 3: return nx_bwd_passes<1, 1>(grid, st, a, first_only);
		case 5: return nx_bwd_passes<2, 1>(grid, st, a, first_only);
		case 6: return nx_bwd_passes<3, 0>(grid, st, a, first_only);
		case 7: return nx_bwd_passes<3, 1>(grid, st, a, first_only);
		default: return -1;
	}
}
