// network_generic.cuh — NerfNetwork for the configurations the fused MFMA kernels of network.hip are not specialised for (included by network.hip, inside namespace ngp):
//   * per-image extra dims behind the direction encoding — latent codes (`n_extra_learnable_dims`) and light directions (`driver_parameters`):
//     include/neural-graphics-primitives/nerf_network.h:81-84 (the dir encoding takes n_dir_dims + n_extra_dims inputs), configs/nerf/base.json:37-51 (Composite:
//     SphericalHarmonics on 3 dims, Identity on the rest), src/testbed_nerf.cu:1710-1746, 2297-2338, 3029-3054;
//   * rgb_network.n_hidden_layers 0 / 1 / 3 (configs/nerf/base_{0,1,3}layer.json; src/testbed.cu:2337-2363 builds whatever the JSON says).
// Default-off features and foreign snapshots: the design goal here is "every such network runs, bit-compatible with the oracle's restatement", not the roofline —
// the base family (no extra dims, two hidden colour layers) never comes here.  One WAVE walks 4 samples at a time through the layers: lane = output neuron,
// activations of the 4 samples in LDS as [feature][4] fp16 (one broadcast 8-byte read feeds 4 FMAs), weights in LDS transposed ([in][out]: conflict-free).
// Sums run sequentially over the inputs in fp32 with separate multiply and add — the order and roundings of oracle/orc_netx.c, so outputs are bit-identical to it.
// Backward: per-sample dgrad the same way ([out][in] weights in LDS), weight gradients accumulated in registers (lane = input neuron, one register per output
// neuron; the workgroup's 4 waves split the matrices and each sums over all 16 samples of the iteration), one fp32 partial per workgroup for wgrad_reduce_kernel,
// dL/dx as level planes for the binned hash-grid backward of network.hip, dL/d(extra dims) per sample for the latent-code optimiser.
#pragma once

struct GenLayout {
	uint32_t n_extra, n_hidden, rgb_in, n_mats, n_mlp;
	uint32_t n_out[6], n_in[6], off[6];
	uint32_t relu_mask;   // bit m: ReLU on the output of matrix m
};
__host__ __device__ inline GenLayout gen_layout(uint32_t n_extra, uint32_t n_hidden) {
	GenLayout L;
	L.n_extra = n_extra; L.n_hidden = n_hidden;
	L.rgb_in = 16u + ((16u + n_extra + 15u) / 16u) * 16u;
	uint32_t m = 0;
	L.n_out[m] = 64; L.n_in[m] = 32; ++m;          // density hidden layer, ReLU
	L.n_out[m] = 16; L.n_in[m] = 64; ++m;          // density output
	L.relu_mask = 1u;
	if (n_hidden == 0) { L.n_out[m] = 16; L.n_in[m] = L.rgb_in; ++m; }
	else {
		L.relu_mask |= 1u << m; L.n_out[m] = 64; L.n_in[m] = L.rgb_in; ++m;
		for (uint32_t k = 1; k < n_hidden; ++k) { L.relu_mask |= 1u << m; L.n_out[m] = 64; L.n_in[m] = 64; ++m; }
		L.n_out[m] = 16; L.n_in[m] = 64; ++m;
	}
	L.n_mats = m;
	uint32_t off = 0;
	for (uint32_t k = 0; k < m; ++k) { L.off[k] = off; off += L.n_out[k] * L.n_in[k]; }
	L.n_mlp = off;
	return L;
}
constexpr uint32_t GEN_MAX_MLP = 64 * 32 + 16 * 64 + 64 * 48 + 2 * 64 * 64 + 16 * 64;   // 15 360: three hidden colour layers + extra dims
constexpr int GEN_SG = 4;                       // samples a wave carries through the layers at once
constexpr int GEN_ROWS_FWD = 32 + 64 + 64 + 64 + 64;        // X, H1, RIN, two ping-pong buffers
constexpr int GEN_ROWS_BWD = 32 + 64 + 64 + 4 * 64 /* colour activations */ + 16 /* d_out */ + 3 * 64 /* colour deltas */ + 64 /* d_in */ + 64 /* d_h1 */ + 32 /* d_x */;

typedef _Float16 gen_h4 __attribute__((ext_vector_type(4)));
// fp32 -> fp16 of a value that was ROUNDED to fp32 first: without the barrier the compiler folds the producing add / multiply into v_fma_mixlo_f16, which rounds the exact
// result once — a double-rounding difference from the oracle's (and tcnn's) "fp32 result, then __float2half" in one of a few thousand values
__device__ __forceinline__ half_t gen_to_half(float v) { asm volatile("" : "+v"(v)); return (half_t)v; }

// SH degree 4 in the oracle's order (oracle/orc_network.c orc_sh4)
__device__ __forceinline__ void gen_sh4(float dx, float dy, float dz, float* out) {
#pragma clang fp contract(off)
	const float x = dx * 2.0f - 1.0f, y = dy * 2.0f - 1.0f, z = dz * 2.0f - 1.0f;
	const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	out[0] = 0.28209479177387814f;
	out[1] = -0.48860251190291987f * y;
	out[2] = 0.48860251190291987f * z;
	out[3] = -0.48860251190291987f * x;
	out[4] = 1.0925484305920792f * xy;
	out[5] = -1.0925484305920792f * yz;
	out[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
	out[7] = -1.0925484305920792f * xz;
	out[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
	out[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
	out[10] = 2.8906114426405538f * xy * z;
	out[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
	out[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
	out[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
	out[14] = 1.4453057213202769f * z * (x2 - y2);
	out[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// y[o][k] = act(sum_i W[o][i] x[i][k]) for the wave's 4 samples; wt: the matrix transposed ([in][out]) in LDS; in / out: [feature][4] fp16 in LDS
__device__ __forceinline__ void gen_dense(const half_t* __restrict__ wt, uint32_t n_out, uint32_t n_in, bool relu, const half_t* __restrict__ in, half_t* __restrict__ out, int lane) {
#pragma clang fp contract(off)
	if ((uint32_t)lane < n_out) {
		float acc[GEN_SG] = {0.f, 0.f, 0.f, 0.f};
		for (uint32_t i = 0; i < n_in; ++i) {
			const float w = (float)wt[i * n_out + lane];
			const gen_h4 x4 = *(const gen_h4*)(in + i * GEN_SG);
#pragma unroll
			for (int k = 0; k < GEN_SG; ++k) { const float p = w * (float)x4[k]; acc[k] = acc[k] + p; }
		}
		gen_h4 y;
#pragma unroll
		for (int k = 0; k < GEN_SG; ++k) { float v = acc[k]; if (relu && v < 0.0f) v = 0.0f; y[k] = gen_to_half(v); }
		*(gen_h4*)(out + lane * GEN_SG) = y;
	}
}
// dx[i][k] = relu'(act[i][k]) * sum_o W[o][i] dy[o][k]; w: the matrix as stored ([out][in]) in LDS
__device__ __forceinline__ void gen_dense_bwd(const half_t* __restrict__ w, uint32_t n_out, uint32_t n_in, const half_t* __restrict__ dy, const half_t* __restrict__ fwd_act /* or NULL */,
                                              half_t* __restrict__ dx, int lane) {
#pragma clang fp contract(off)
	if ((uint32_t)lane < n_in) {
		float acc[GEN_SG] = {0.f, 0.f, 0.f, 0.f};
		for (uint32_t o = 0; o < n_out; ++o) {
			const float wv = (float)w[o * n_in + lane];
			const gen_h4 d4 = *(const gen_h4*)(dy + o * GEN_SG);
#pragma unroll
			for (int k = 0; k < GEN_SG; ++k) { const float p = wv * (float)d4[k]; acc[k] = acc[k] + p; }
		}
		gen_h4 r;
#pragma unroll
		for (int k = 0; k < GEN_SG; ++k) {
			float v = acc[k];
			if (fwd_act && !((float)fwd_act[lane * GEN_SG + k] > 0.0f)) v = 0.0f;
			r[k] = gen_to_half(v);
		}
		*(gen_h4*)(dx + lane * GEN_SG) = r;
	}
}

struct GenExtra { const float* extra_dims; const uint32_t* sample_slot; };

// one level of the hash encoding with every product and sum rounded separately (oracle/orc_network.c orc_grid_encode_one): network.hip's encode_level lives under
// `fp contract(fast)` and may differ from that in the last fp16 bit
__device__ __forceinline__ void gen_encode_level(const NgpGridLevel lv, const h2* __restrict__ grid, float px, float py, float pz, half_t& o0, half_t& o1) {
#pragma clang fp contract(off)
	const LevelPos p = level_pos(lv, px, py, pz);
	float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		float w = 1.0f;
		w = w * ((c & 1) ? p.fx : (1.0f - p.fx));
		w = w * (((c >> 1) & 1) ? p.fy : (1.0f - p.fy));
		w = w * (((c >> 2) & 1) ? p.fz : (1.0f - p.fz));
		const h2 v = grid[lv.offset + grid_index(lv, p.gx + (c & 1), p.gy + ((c >> 1) & 1), p.gz + ((c >> 2) & 1))];
		const float t0 = w * (float)v[0], t1 = w * (float)v[1];
		r0 = r0 + t0; r1 = r1 + t1;
	}
	o0 = gen_to_half(r0); o1 = gen_to_half(r1);
}

// the wave's 4 samples up to the colour network's input: X (encoding), H1, RIN = [density out 16 | SH 16 | extra dims, zero padded]
__device__ __forceinline__ void gen_front(const GenLayout& L, const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const half_t* __restrict__ wt,
                                          const float* __restrict__ coords, uint32_t coord_stride, uint32_t n, uint32_t s0, const half_t* __restrict__ x_rows /* saved encodings or NULL */,
                                          GenExtra ex, half_t* __restrict__ X, half_t* __restrict__ H1, half_t* __restrict__ RIN, int lane, bool density_only = false) {
	const int k = lane >> 4, q = lane & 15;
	const uint32_t s = s0 + k < n ? s0 + k : n - 1;
	const float* c = coords + (size_t)s * coord_stride;
	{   // lane (k, level q): one level of sample k
		half_t a, b;
		if (x_rows) { const h2 v = *(const h2*)(x_rows + (size_t)s * 32 + 2 * q); a = v[0]; b = v[1]; }
		else gen_encode_level(desc->levels[q], (const h2*)(params + L.n_mlp), c[0], c[1], c[2], a, b);
		X[(2 * q) * GEN_SG + k] = a; X[(2 * q + 1) * GEN_SG + k] = b;
	}
	if (q == 0 && !density_only) {   // (density(): positions only, the records may be 3 floats long)
		float sh[16];
		gen_sh4(c[4], c[5], c[6], sh);
#pragma unroll
		for (int i = 0; i < 16; ++i) RIN[(16 + i) * GEN_SG + k] = gen_to_half(sh[i]);
	}
	if (L.rgb_in > 32 && !density_only) {   // [tcnn] Identity encoding: the input cast to the network precision; rows beyond n_extra are padding
		float v = 0.0f;
		if ((uint32_t)q < L.n_extra && ex.extra_dims) v = ex.extra_dims[(size_t)(ex.sample_slot ? ex.sample_slot[s] : 0u) * L.n_extra + q];
		RIN[(32 + q) * GEN_SG + k] = (half_t)v;
	}
	gen_dense(wt + L.off[0], 64, 32, true, X, H1, lane);
	gen_dense(wt + L.off[1], 16, 64, false, H1, RIN, lane);   // the density network's 16 outputs are rows 0..15 of the colour network's input (nerf_network.h:108, 160)
}

__device__ __forceinline__ void gen_stage_weights(const GenLayout& L, const half_t* __restrict__ params, half_t* __restrict__ w_lds /* or NULL */, half_t* __restrict__ wt_lds) {
	for (uint32_t m = 0; m < L.n_mats; ++m) {
		const uint32_t cnt = L.n_out[m] * L.n_in[m];
		for (uint32_t e = threadIdx.x; e < cnt; e += blockDim.x) {
			const half_t v = params[L.off[m] + e];
			const uint32_t o = e / L.n_in[m], i = e - o * L.n_in[m];
			if (w_lds) w_lds[L.off[m] + e] = v;
			wt_lds[L.off[m] + i * L.n_out[m] + o] = v;
		}
	}
	__syncthreads();
}

// MODE 0: inference (rgb sigma).  MODE 1: density only (out[s] = channel 0 of the density network, NerfNetwork::density).  MODE 2: training forward, also saves the encoding rows.
template <int MODE>
__global__ void __launch_bounds__(256) gen_forward_kernel(GenLayout L, const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ coords,
                                                          uint32_t coord_stride, uint32_t n, half_t* __restrict__ out, uint32_t out_stride, half_t* __restrict__ x_saved, GenExtra ex) {
	__shared__ __attribute__((aligned(16))) half_t wt[GEN_MAX_MLP];
	__shared__ __attribute__((aligned(16))) half_t acts[4][GEN_ROWS_FWD * GEN_SG];
	gen_stage_weights(L, params, nullptr, wt);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	half_t* X = acts[w]; half_t* H1 = X + 32 * GEN_SG; half_t* RIN = H1 + 64 * GEN_SG; half_t* A0 = RIN + 64 * GEN_SG; half_t* A1 = A0 + 64 * GEN_SG;
	const uint32_t n_groups = (n + GEN_SG - 1) / GEN_SG;
	for (uint32_t grp = blockIdx.x * 4 + w; grp < n_groups; grp += gridDim.x * 4) {
		const uint32_t s0 = grp * GEN_SG;
		gen_front(L, desc, params, wt, coords, coord_stride, n, s0, nullptr, ex, X, H1, RIN, lane, MODE == 1);
		if (MODE == 1) {
			if (lane < GEN_SG && s0 + lane < n) out[s0 + lane] = RIN[lane];
			continue;
		}
		const half_t* cur = RIN; half_t* nxt = A0;
		for (uint32_t m = 2; m < L.n_mats; ++m) {
			gen_dense(wt + L.off[m], L.n_out[m], L.n_in[m], (L.relu_mask >> m) & 1u, cur, nxt, lane);
			cur = nxt; nxt = nxt == A0 ? A1 : A0;
		}
		const int k = lane >> 4, q = lane & 15;
		const uint32_t s = s0 + k;
		if (s < n) {
			if (q < 3) out[(size_t)s * out_stride + q] = cur[q * GEN_SG + k];
			if (q == 3) out[(size_t)s * out_stride + 3] = RIN[k];   // extract_density (nerf_network.h:32-43)
			if (MODE == 2) { h2 v; v[0] = X[(2 * q) * GEN_SG + k]; v[1] = X[(2 * q + 1) * GEN_SG + k]; *(h2*)(x_saved + (size_t)s * 32 + 2 * q) = v; }
		}
	}
}

// forward (recomputed from the saved encodings) + backward of 16 samples per workgroup iteration.  n must be a multiple of 16.
__global__ void __launch_bounds__(256) gen_backward_kernel(GenLayout L, const NgpNetDesc* __restrict__ desc, const half_t* __restrict__ params, const float* __restrict__ coords,
                                                           uint32_t coord_stride, uint32_t n, const half_t* __restrict__ x_saved, const half_t* __restrict__ dL_dout, uint32_t dl_stride,
                                                           h2* __restrict__ dx_planes, float* __restrict__ partials /* [gridDim.x][L.n_mlp] */, GenExtra ex, float* __restrict__ dL_dextra /* [n][n_extra] or NULL */) {
#pragma clang fp contract(off)
	extern __shared__ __attribute__((aligned(16))) char gen_smem[];
	half_t* w_lds = (half_t*)gen_smem;
	half_t* wt = w_lds + GEN_MAX_MLP;
	half_t* acts_all = wt + GEN_MAX_MLP;
	gen_stage_weights(L, params, w_lds, wt);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	auto wave_buf = [&](int wv) { return acts_all + (size_t)wv * GEN_ROWS_BWD * GEN_SG; };
	// row offsets inside a wave's buffer
	constexpr int R_X = 0, R_H1 = 32, R_RIN = 96, R_ACT = 160 /* + 64 m' */, R_DOUT = 416, R_DACT = 432 /* + 64 m' */, R_DIN = 624, R_DH1 = 688, R_DX = 752;
	static_assert(R_DX + 32 == GEN_ROWS_BWD, "rows");
	const uint32_t n_col = L.n_mats - 2;   // colour matrices
	// weight-gradient accumulators: wave w owns matrices {ma, mb} (mb may be none): registers acc[o] for its lane's input neuron
	int ma, mb;
	{
		// wave 0: W1 + W2; wave 1: first colour matrix + the last one; waves 2, 3: the hidden-to-hidden matrices, if any
		const int first = 2, last = (int)L.n_mats - 1;
		if (w == 0) { ma = 0; mb = 1; }
		else if (w == 1) { ma = first; mb = last != first ? last : -1; }
		else { const int mid = first + (w - 1); ma = mid < last ? mid : -1; mb = -1; }
	}
	float acc_a[64], acc_b[16];
#pragma unroll
	for (int o = 0; o < 64; ++o) acc_a[o] = 0.0f;
#pragma unroll
	for (int o = 0; o < 16; ++o) acc_b[o] = 0.0f;

	const uint32_t n_iters = n / 16;
	for (uint32_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
		half_t* B = wave_buf(w);
		const uint32_t s0 = it * 16 + w * GEN_SG;
		gen_front(L, desc, params, wt, coords, coord_stride, n, s0, x_saved, ex, B + R_X * GEN_SG, B + R_H1 * GEN_SG, B + R_RIN * GEN_SG, lane);
		const half_t* cur = B + R_RIN * GEN_SG;
		for (uint32_t m = 0; m < n_col; ++m) {
			half_t* o = B + (R_ACT + 64 * (int)m) * GEN_SG;
			gen_dense(wt + L.off[2 + m], L.n_out[2 + m], L.n_in[2 + m], (L.relu_mask >> (2 + m)) & 1u, cur, o, lane);
			cur = o;
		}
		// d_out: rgb gradients in rows 0..2, the padded outputs carry none
		{
			const int k = lane >> 4, q = lane & 15;
			half_t v = (half_t)0.0f;
			if (q < 3) v = dL_dout[(size_t)(s0 + k) * dl_stride + q];
			B[(R_DOUT + q) * GEN_SG + k] = v;
		}
		// colour network, last matrix to first: dy of matrix m lives in R_DOUT (last) or R_DACT + 64 m (the delta of activation m = input of matrix m + 1)
		for (int m = (int)n_col - 1; m >= 0; --m) {
			const half_t* dy = m == (int)n_col - 1 ? B + R_DOUT * GEN_SG : B + (R_DACT + 64 * m) * GEN_SG;
			if (m == 0) gen_dense_bwd(w_lds + L.off[2], L.n_out[2], L.n_in[2], dy, nullptr, B + R_DIN * GEN_SG, lane);
			else gen_dense_bwd(w_lds + L.off[2 + m], L.n_out[2 + m], L.n_in[2 + m], dy, B + (R_ACT + 64 * (m - 1)) * GEN_SG, B + (R_DACT + 64 * (m - 1)) * GEN_SG, lane);
		}
		{
			const int k = lane >> 4, q = lane & 15;
			const uint32_t s = s0 + k;
			if (dL_dextra && (uint32_t)q < L.n_extra) dL_dextra[(size_t)s * L.n_extra + q] = (float)B[(R_DIN + 32 + q) * GEN_SG + k];   // [tcnn] Identity backward
			if (q == 0) {   // add_density_gradient (nerf_network.h:63-74): fp16 += fp16
				const float d = (float)B[R_DIN * GEN_SG + k] + (float)dL_dout[(size_t)s * dl_stride + 3];
				B[R_DIN * GEN_SG + k] = gen_to_half(d);
			}
		}
		gen_dense_bwd(w_lds + L.off[1], 16, 64, B + R_DIN * GEN_SG, B + R_H1 * GEN_SG, B + R_DH1 * GEN_SG, lane);
		gen_dense_bwd(w_lds + L.off[0], 64, 32, B + R_DH1 * GEN_SG, nullptr, B + R_DX * GEN_SG, lane);
		{   // dL/dx as level planes [level][sample] half2 for the hash-grid backward
			const int k = lane >> 4, q = lane & 15;
			h2 v; v[0] = B[(R_DX + 2 * q) * GEN_SG + k]; v[1] = B[(R_DX + 2 * q + 1) * GEN_SG + k];
			dx_planes[(size_t)q * n + s0 + k] = v;
		}
		__syncthreads();
		// ---- weight gradients of this iteration's 16 samples: dW[o][i] += dy[o] * h[i]
		auto dy_rows = [&](int m) -> int { return m == 0 ? R_DH1 : m == 1 ? R_DIN : (m == (int)L.n_mats - 1 ? R_DOUT : R_DACT + 64 * (m - 2)); };
		auto in_rows = [&](int m) -> int { return m == 0 ? R_X : m == 1 ? R_H1 : (m == 2 ? R_RIN : R_ACT + 64 * (m - 3)); };
		for (int wv = 0; wv < 4; ++wv) {
			const half_t* Bv = wave_buf(wv);
			if (ma >= 0 && (uint32_t)lane < L.n_in[ma]) {
				const gen_h4 h4 = *(const gen_h4*)(Bv + (in_rows(ma) + lane) * GEN_SG);
				const half_t* dyp = Bv + dy_rows(ma) * GEN_SG;
				const uint32_t no = L.n_out[ma];
#pragma unroll
				for (int o = 0; o < 64; ++o) {
					if ((uint32_t)o < no) {
						const gen_h4 d4 = *(const gen_h4*)(dyp + o * GEN_SG);
#pragma unroll
						for (int k = 0; k < GEN_SG; ++k) { const float p = (float)d4[k] * (float)h4[k]; acc_a[o] = acc_a[o] + p; }
					}
				}
			}
			if (mb >= 0 && (uint32_t)lane < L.n_in[mb]) {
				const gen_h4 h4 = *(const gen_h4*)(Bv + (in_rows(mb) + lane) * GEN_SG);
				const half_t* dyp = Bv + dy_rows(mb) * GEN_SG;
#pragma unroll
				for (int o = 0; o < 16; ++o) {
					const gen_h4 d4 = *(const gen_h4*)(dyp + o * GEN_SG);
#pragma unroll
					for (int k = 0; k < GEN_SG; ++k) { const float p = (float)d4[k] * (float)h4[k]; acc_b[o] = acc_b[o] + p; }
				}
			}
		}
		__syncthreads();
	}
	float* __restrict__ dst = partials + (size_t)blockIdx.x * L.n_mlp;
	if (ma >= 0 && (uint32_t)lane < L.n_in[ma]) {
#pragma unroll
		for (int o = 0; o < 64; ++o) if ((uint32_t)o < L.n_out[ma]) dst[L.off[ma] + o * L.n_in[ma] + lane] = acc_a[o];
	}
	if (mb >= 0 && (uint32_t)lane < L.n_in[mb]) {
#pragma unroll
		for (int o = 0; o < 16; ++o) dst[L.off[mb] + o * L.n_in[mb] + lane] = acc_b[o];
	}
}
constexpr size_t GEN_BWD_SMEM = (size_t)(2 * GEN_MAX_MLP + 4 * GEN_ROWS_BWD * GEN_SG) * sizeof(half_t);

// parameter init: [tcnn] Xavier-uniform per matrix, U(-1e-4, 1e-4) for the grid; element k <- k-th draw of pcg32(seed) (orc_nerf_init_params_x)
__global__ void gen_init_params_kernel(GenLayout L, uint32_t n_params, uint64_t seed_state, uint64_t seed_inc, float* __restrict__ master, half_t* __restrict__ params, half_t* __restrict__ inference) {
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_params) return;
	Pcg32 rng; rng.state = seed_state; rng.inc = seed_inc;
	rng.advance(k);
	float scale = 1e-4f;
	for (uint32_t m = 0; m < L.n_mats; ++m) if (k >= L.off[m] && k < L.off[m] + L.n_out[m] * L.n_in[m]) scale = sqrtf(6.0f / (float)(L.n_out[m] + L.n_in[m]));
	float v;
	{
#pragma clang fp contract(off)
		v = rng.next_float() * (scale - (-scale)) + (-scale);
	}
	master[k] = v; params[k] = (half_t)v; inference[k] = (half_t)v;
}

// compute_extra_dims_gradient_train_nerf (src/testbed_nerf.cu:1710-1746): one thread per kept ray; the extra-dim gradients of its compacted samples into its image's row
__global__ void __launch_bounds__(128) extra_dims_gradient_kernel(uint32_t n_rays_capacity, const uint32_t* __restrict__ rays_counter, const uint32_t* __restrict__ ray_image,
                                                                  const uint32_t* __restrict__ numsteps, const float* __restrict__ dL_dextra, uint32_t n_extra, float* __restrict__ gradient) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_rays_capacity || i >= *rays_counter) return;
	const uint32_t cnt = numsteps[2 * i], base = numsteps[2 * i + 1];
	if (cnt == 0) return;
	float* g = gradient + (size_t)ray_image[i] * n_extra;
	for (uint32_t k = 0; k < n_extra; ++k) {
		float sum = 0.0f;
		for (uint32_t j = 0; j < cnt; ++j) sum += dL_dextra[(size_t)(base + j) * n_extra + k];
		atomicAdd(&g[k], sum);
	}
}

// ray_image[i] = image_idx of kept ray i (src/testbed_nerf.cu:1062-1083, 1131-1136, 1736), once per step for the three kernels below
__global__ void __launch_bounds__(128) ray_images_kernel(uint32_t n_rays_capacity, const uint32_t* __restrict__ rays_counter, const uint32_t* __restrict__ ray_indices, uint32_t n_rays_global,
                                                         uint32_t n_training_images, const float* __restrict__ cdf_img, uint32_t* __restrict__ ray_image) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_rays_capacity || i >= *rays_counter) return;
	ray_image[i] = image_idx(ray_indices[i], n_rays_global, n_training_images, cdf_img, nullptr);
}

// per-sample row index into the extra-dims table: sample_slot[base + j] = ray_image[ray] for every kept ray's run; then (compacted batch only) the roll-over rule
// slot[k] = slot[k % n_kept] for the padded tail, like fill_rollover (src/testbed_nerf.cu:3314-3322)
__global__ void __launch_bounds__(128) expand_ray_slots_kernel(uint32_t n_rays_capacity, const uint32_t* __restrict__ rays_counter, const uint32_t* __restrict__ ray_image,
                                                               const uint32_t* __restrict__ numsteps, uint32_t n_samples_capacity, uint32_t* __restrict__ sample_slot) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_rays_capacity || i >= *rays_counter) return;
	const uint32_t cnt = numsteps[2 * i], base = numsteps[2 * i + 1];
	const uint32_t img = ray_image[i];
	for (uint32_t j = 0; j < cnt && base + j < n_samples_capacity; ++j) sample_slot[base + j] = img;
}
__global__ void __launch_bounds__(256) rollover_slots_kernel(uint32_t n_elements, const uint32_t* __restrict__ n_input_elements, uint32_t* __restrict__ sample_slot) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t n_in = *n_input_elements;
	if (i < n_in || i >= n_elements || n_in == 0) return;
	sample_slot[i] = sample_slot[i % n_in];
}
