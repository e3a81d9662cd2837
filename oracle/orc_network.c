/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_core.h header).  PARITY UNPINNED.
 *
 * NerfNetwork = HashGrid(L=16,F=2) -> density MLP 32->64->16 -> [16 | SH4(dir) 16] -> rgb MLP 32->64->64->16(3).
 * Composition follows include/neural-graphics-primitives/nerf_network.h:76-548
 * (inference 103-137, forward 143-185, backward 187-266, param order 361-394, init 396-441).
 * The arithmetic of GridEncoding / FullyFusedMLP / SphericalHarmonics / Adam / Ema lives in
 * NVlabs/tiny-cuda-nn (submodule, .gitmodules:16-18, UNPINNED commit, sources absent) and is restated
 * from that library's published algorithm [tcnn]:
 *   grid:  scale_l = exp2(l*log2(b))*N_min - 1; res_l = ceil(scale_l)+1; pos = fma(scale, x, 0.5) (one rounding); floor/fract;
 *          dense index = x + y*res + z*res^2 (stride stops growing once > level size), hashed index =
 *          x*1 ^ y*2654435761 ^ z*805459861; index %= level size; trilinear weights; fp16 features.
 *   mlp:   row-major [out][in] fp16 weights, fp32 accumulate, ReLU on hidden layers, fp16 activations.
 *   sh4:   16 real SH basis functions of 2*d-1.
 * Rounding points chosen here (and mirrored by the HIP kernels): fp32 accumulation everywhere, one
 * round-to-nearest-even conversion to fp16 per stored activation / gradient.
 */
#include "ngp_oracle.h"
#include <stdlib.h>

/* [tcnn] grid.h: grid_scale / grid_resolution / offset table; testbed.cu:2313-2325 gives per_level_scale */
void orc_net_make_levels(uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale,
                         orc_grid_level* levels, uint32_t* n_grid_entries) {
	uint32_t offset = 0;
	float log2_pls = log2f(per_level_scale);
	for (uint32_t l = 0; l < n_levels; ++l) {
		float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
		uint32_t res = (uint32_t)ceilf(scale) + 1u;
		uint64_t dense = (uint64_t)res * res * res;
		uint32_t max_params = 0xffffffffu / 2u;
		uint32_t n = dense > max_params ? max_params : (uint32_t)dense;
		n = (n + 7u) / 8u * 8u;
		uint32_t hashmap = 1u << log2_hashmap_size;
		uint32_t size = n < hashmap ? n : hashmap;
		levels[l].scale = scale;
		levels[l].resolution = res;
		levels[l].offset = offset;
		levels[l].size = size;
		offset += size;
	}
	*n_grid_entries = offset;
}

/* testbed.cu:2313-2325 */
float orc_per_level_scale(uint32_t n_levels, uint32_t base_resolution, float desired_resolution, uint32_t aabb_scale) {
	return expf(logf(desired_resolution * (float)aabb_scale / (float)base_resolution) / (float)(n_levels - 1));
}

static inline uint32_t orc_grid_index(const orc_grid_level* lv, uint32_t x, uint32_t y, uint32_t z) {
	uint32_t stride = 1, index = 0;
	const uint32_t p[3] = {x, y, z};
	for (int d = 0; d < 3 && stride <= lv->size; ++d) {
		index += p[d] * stride;
		stride *= lv->resolution;
	}
	if (lv->size < stride) {
		index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
	}
	return index % lv->size;
}

/* [tcnn] kernel_grid forward, one sample: pos in [0,1]^3 -> 2*L fp16 features (returned as fp16 bits) */
uint32_t orc_grid_index_export(const orc_grid_level* lv, uint32_t x, uint32_t y, uint32_t z) { return orc_grid_index(lv, x, y, z); }

void orc_grid_encode_one(const orc_net* net, const uint16_t* grid /* fp16 [entries][2] */, const float pos_in[3], uint16_t* out) {
	for (uint32_t l = 0; l < net->n_levels; ++l) {
		const orc_grid_level* lv = &net->levels[l];
		float pos[3]; uint32_t pg[3];
		for (int d = 0; d < 3; ++d) {
			float p = fmaf(lv->scale, pos_in[d], 0.5f);   /* [tcnn] pos_fract: fmaf(scale, input, 0.5f) */
			float fl = floorf(p);
			pg[d] = (uint32_t)(int)fl;
			pos[d] = p - fl;
		}
		float r0 = 0.0f, r1 = 0.0f;
		for (uint32_t idx = 0; idx < 8; ++idx) {
			float w = 1.0f; uint32_t c[3];
			for (int d = 0; d < 3; ++d) {
				if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; c[d] = pg[d]; }
				else { w *= pos[d]; c[d] = pg[d] + 1; }
			}
			uint32_t gi = orc_grid_index(lv, c[0], c[1], c[2]);
			const uint16_t* v = grid + 2u * ((size_t)lv->offset + gi);
			r0 += w * orc_h2f(v[0]);
			r1 += w * orc_h2f(v[1]);
		}
		out[2 * l + 0] = orc_f2h(r0);
		out[2 * l + 1] = orc_f2h(r1);
	}
}

/* [tcnn] SphericalHarmonics degree 4: x,y,z = 2*d-1 */
void orc_sh4(const float dir[3], float out[16]) {
	float x = dir[0] * 2.0f - 1.0f, y = dir[1] * 2.0f - 1.0f, z = dir[2] * 2.0f - 1.0f;
	float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
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

/* y[out] = act(W[out][in] . x[in]); fp16 in, fp32 acc, fp16 out */
static void orc_dense(const uint16_t* W, uint32_t n_out, uint32_t n_in, const uint16_t* x, uint16_t* y, int relu) {
	for (uint32_t o = 0; o < n_out; ++o) {
		float acc = 0.0f;
		for (uint32_t i = 0; i < n_in; ++i) acc += orc_h2f(W[o * n_in + i]) * orc_h2f(x[i]);
		if (relu && acc < 0.0f) acc = 0.0f;
		y[o] = orc_f2h(acc);
	}
}
/* dx[in] = W^T . dy ; fp16 in, fp32 acc, optional relu mask from forward activation, fp16 out */
static void orc_dense_bwd_input(const uint16_t* W, uint32_t n_out, uint32_t n_in, const uint16_t* dy, const uint16_t* fwd_act /* [in] or NULL */, uint16_t* dx) {
	for (uint32_t i = 0; i < n_in; ++i) {
		float acc = 0.0f;
		for (uint32_t o = 0; o < n_out; ++o) acc += orc_h2f(W[o * n_in + i]) * orc_h2f(dy[o]);
		if (fwd_act && !(orc_h2f(fwd_act[i]) > 0.0f)) acc = 0.0f;
		dx[i] = orc_f2h(acc);
	}
}

/* offsets into the parameter vector: nerf_network.h:361-394 (density MLP, rgb MLP, pos grid, dir enc[0 params]) */
#define W1_OFF 0u                 /* [64][32] */
#define W2_OFF (64u * 32u)        /* [16][64] */
#define W3_OFF (W2_OFF + 16u * 64u) /* [64][32] */
#define W4_OFF (W3_OFF + 64u * 32u) /* [64][64] */
#define W5_OFF (W4_OFF + 64u * 64u) /* [16][64] */
#define GRID_OFF (W5_OFF + 16u * 64u) /* = 10240 */

uint32_t orc_net_n_params(const orc_net* net) { return GRID_OFF + 2u * net->n_grid_entries; }
uint32_t orc_net_mlp_params(void) { return GRID_OFF; }

/* One sample through the whole network keeping activations (all fp16 bits).
 * act layout (orc_act): x[32] h1[64] in_rgb[32] h2[64] h3[64] out[16] */
void orc_nerf_forward_one(const orc_net* net, const uint16_t* params, const float coord[7], orc_act* a) {
	orc_grid_encode_one(net, params + GRID_OFF, coord, a->x);
	orc_dense(params + W1_OFF, 64, 32, a->x, a->h1, 1);
	orc_dense(params + W2_OFF, 16, 64, a->h1, a->in_rgb, 0);  /* density net output -> first 16 rows of rgb net input (nerf_network.h:108, 160) */
	float sh[16];
	orc_sh4(coord + 4, sh);                                    /* dir at float offset 4 (testbed.cu:2358 dir_offset = n_pos+1) */
	for (int i = 0; i < 16; ++i) a->in_rgb[16 + i] = orc_f2h(sh[i]);
	orc_dense(params + W3_OFF, 64, 32, a->in_rgb, a->h2, 1);
	orc_dense(params + W4_OFF, 64, 64, a->h2, a->h3, 1);
	orc_dense(params + W5_OFF, 16, 64, a->h3, a->out, 0);
	a->out[3] = a->in_rgb[0];                                  /* extract_density: nerf_network.h:32-43, 130-136 */
}

/* [tcnn] Network::visualize_activation: forward pass, then extract_dimension_pos_neg_kernel on forward_activations(layer) with NerfNetwork's layer
 * map (nerf_network.h:474-501): 0 = encoding, 1 = density hidden layer, 2 = colour network input, 3.. = colour hidden layers.  Output row 0 =
 * max(-v, 0), row 1 = max(v, 0), row 2 = 0, further rows = 1; a single output row gets v itself.  `out` may alias `coords`. */
void orc_nerf_visualize_activation(const orc_net* net, const uint16_t* params, uint32_t layer, uint32_t dimension, const float* coords, uint32_t coord_stride_floats,
                                   uint32_t n, float* out, uint32_t out_stride_floats) {
	#pragma omp parallel for schedule(static) if (n >= 512)
	for (uint32_t i = 0; i < n; ++i) {
		orc_act a;
		orc_nerf_forward_one(net, params, coords + (size_t)i * coord_stride_floats, &a);
		const uint16_t* src = layer == 0 ? a.x : layer == 1 ? a.h1 : layer == 2 ? a.in_rgb : layer == 3 ? a.h2 : a.h3;
		const float v = orc_h2f(src[dimension]);
		float* o = out + (size_t)i * out_stride_floats;
		if (out_stride_floats == 1) { o[0] = v; continue; }
		for (uint32_t k = 0; k < out_stride_floats; ++k) o[k] = k == 0 ? fmaxf(-v, 0.0f) : k == 1 ? fmaxf(v, 0.0f) : k == 2 ? 0.0f : 1.0f;
	}
}

/* nerf_network.h:103-137: N samples -> rgbsigma fp16, `out_stride` halves per sample (>= 4), channels 0..3 written.
 * With out_stride >= 16 all 16 padded rgb-net outputs are written like the reference's AoS matrix. */
void orc_nerf_inference(const orc_net* net, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                        uint32_t n, uint16_t* out, uint32_t out_stride) {
	#pragma omp parallel for schedule(static) if (n >= 512)
	for (uint32_t i = 0; i < n; ++i) {
		orc_act a;
		orc_nerf_forward_one(net, params, coords + (size_t)i * coord_stride_floats, &a);
		uint32_t w = out_stride >= 16 ? 16 : 4;
		for (uint32_t c = 0; c < w; ++c) out[(size_t)i * out_stride + c] = a.out[c];
	}
}

/* nerf_network.h:268-284 density(): pos (3 floats, stride given) -> 16 fp16 density-net outputs; channel 0 is the density logit */
void orc_nerf_density(const orc_net* net, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n, uint16_t* out0) {
	#pragma omp parallel for schedule(static) if (n >= 512)
	for (uint32_t i = 0; i < n; ++i) {
		uint16_t x[32], h1[64], d[16];
		orc_grid_encode_one(net, params + GRID_OFF, pos + (size_t)i * pos_stride_floats, x);
		orc_dense(params + W1_OFF, 64, 32, x, h1, 1);
		orc_dense(params + W2_OFF, 16, 64, h1, d, 0);
		out0[i] = d[0];
	}
}

/* nerf_network.h:143-266 forward + backward over a batch.
 * dL_dout: fp16 [n][4] (rgb, sigma) — the reference matrix is 16 wide with only 0..3 consumed (extract_rgb 46-60, add_density_gradient 63-74).
 * grads_out: double [n_params] (MLP weight grads + grid grads), accumulated here in double; the device keeps fp32/fp16
 * partial sums, tests compare with a tolerance.  Per-sample fp16 rounding points follow the header comment. */
void orc_nerf_forward_backward(const orc_net* net, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                               uint32_t n, const uint16_t* dL_dout, uint16_t* out_rgbsigma /* [n][4] or NULL */,
                               double* grads_out, uint16_t* dL_dx_out /* [n][32] or NULL */) {
	const uint32_t np = orc_net_n_params(net);
	for (uint32_t k = 0; k < np; ++k) grads_out[k] = 0.0;
	/* samples are independent; MLP weight gradients are accumulated per thread and merged, grid gradients with atomics
	 * (double accumulation: the summation order does not matter at the tolerances the tests use) */
	#pragma omp parallel
	{
	double* mlp_acc = (double*)calloc(GRID_OFF, sizeof(double));
	#pragma omp for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		const float* coord = coords + (size_t)i * coord_stride_floats;
		orc_act a;
		orc_nerf_forward_one(net, params, coord, &a);
		if (out_rgbsigma) for (int c = 0; c < 4; ++c) out_rgbsigma[(size_t)i * 4 + c] = a.out[c];

		uint16_t d_out[16]; memset(d_out, 0, sizeof(d_out));
		for (int c = 0; c < 3; ++c) d_out[c] = dL_dout[(size_t)i * 4 + c];
		uint16_t d_h3[64], d_h2[64], d_in[32], d_h1[64], d_x[32];
		orc_dense_bwd_input(params + W5_OFF, 16, 64, d_out, a.h3, d_h3);
		orc_dense_bwd_input(params + W4_OFF, 64, 64, d_h3, a.h2, d_h2);
		orc_dense_bwd_input(params + W3_OFF, 64, 32, d_h2, NULL, d_in);
		/* add_density_gradient (nerf_network.h:63-74): fp16 += fp16 */
		d_in[0] = orc_f2h(orc_h2f(d_in[0]) + orc_h2f(dL_dout[(size_t)i * 4 + 3]));
		orc_dense_bwd_input(params + W2_OFF, 16, 64, d_in /* first 16 */, a.h1, d_h1);
		orc_dense_bwd_input(params + W1_OFF, 64, 32, d_h1, NULL, d_x);
		if (dL_dx_out) memcpy(dL_dx_out + (size_t)i * 32, d_x, sizeof(d_x));

		/* weight gradients dW[o][i] += dy[o] * x[i] */
#define WGRAD(OFF, NOUT, NIN, DY, X) \
		for (uint32_t o = 0; o < (NOUT); ++o) { float dy_ = orc_h2f((DY)[o]); if (dy_ != 0.0f) for (uint32_t k = 0; k < (NIN); ++k) mlp_acc[(OFF) + o * (NIN) + k] += (double)(dy_ * orc_h2f((X)[k])); }
		WGRAD(W5_OFF, 16, 64, d_out, a.h3)
		WGRAD(W4_OFF, 64, 64, d_h3, a.h2)
		WGRAD(W3_OFF, 64, 32, d_h2, a.in_rgb)
		WGRAD(W2_OFF, 16, 64, d_in, a.h1)
		WGRAD(W1_OFF, 64, 32, d_h1, a.x)
#undef WGRAD

		/* [tcnn] kernel_grid_backward: grad[idx][f] += fp16(w * dL/dx[2l+f]) */
		for (uint32_t l = 0; l < net->n_levels; ++l) {
			const orc_grid_level* lv = &net->levels[l];
			float pos[3]; uint32_t pg[3];
			for (int d = 0; d < 3; ++d) {
				float p = fmaf(lv->scale, coord[d], 0.5f);   /* [tcnn] pos_fract: fmaf(scale, input, 0.5f) */
				float fl = floorf(p);
				pg[d] = (uint32_t)(int)fl;
				pos[d] = p - fl;
			}
			float g0 = orc_h2f(d_x[2 * l]), g1 = orc_h2f(d_x[2 * l + 1]);
			for (uint32_t idx = 0; idx < 8; ++idx) {
				float w = 1.0f; uint32_t c[3];
				for (int d = 0; d < 3; ++d) {
					if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; c[d] = pg[d]; }
					else { w *= pos[d]; c[d] = pg[d] + 1; }
				}
				uint32_t gi = orc_grid_index(lv, c[0], c[1], c[2]);
				size_t k = GRID_OFF + 2u * ((size_t)lv->offset + gi);
				const double a0 = (double)orc_rh(w * g0), a1 = (double)orc_rh(w * g1);
				#pragma omp atomic
				grads_out[k + 0] += a0;
				#pragma omp atomic
				grads_out[k + 1] += a1;
			}
		}
	}
	#pragma omp critical
	for (uint32_t k = 0; k < GRID_OFF; ++k) grads_out[k] += mlp_acc[k];
	free(mlp_acc);
	}
}

/* dL/d(network input) — NerfNetwork::backward_impl with a dL_dinput matrix (nerf_network.h:187-266), requested by the training step when camera
 * parameters train (prepare_input_gradients, testbed_nerf.cu:3324-3346).  [tcnn] GridEncoding: the forward pass keeps dy/dx of the trilinear
 * interpolation per level, feature and dimension in fp32 — weight = scale * prod over the OTHER dimensions of (1 - w | w), times (value at the
 * +1 corner - value at the 0 corner) — and backward_input sums dL/dy * dy/dx over levels and features in fp32.  [tcnn] SphericalHarmonics: the
 * analytic derivative of the 16 polynomials in (2 d - 1), times 2.  dL/dy are the fp16 values the MLP backward produced (d_x, d_in[16..31]).
 * Output: [n][6] fp32 = d/d(pos x, y, z), d/d(dir x, y, z) in the WARPED input coordinates the network sees. */
static void orc_sh4_grad(const float dir[3], const float g[16], float out[3]) {
	const float x = dir[0] * 2.0f - 1.0f, y = dir[1] * 2.0f - 1.0f, z = dir[2] * 2.0f - 1.0f;
	const float x2 = x * x, y2 = y * y, z2 = z * z;
	const float A = 0.48860251190291987f, B = 1.0925484305920792f, C = 0.94617469575755997f, E = 0.54627421529603959f, F = 0.59004358992664352f,
	            G = 2.8906114426405538f, Hh = 0.45704579946446572f, K = 0.3731763325901154f, M = 1.4453057213202769f;
	float dx = 0.0f, dy = 0.0f, dz = 0.0f;
	dy += g[1] * -A;
	dz += g[2] * A;
	dx += g[3] * -A;
	dx += g[4] * (B * y);            dy += g[4] * (B * x);
	dy += g[5] * (-B * z);           dz += g[5] * (-B * y);
	dz += g[6] * (2.0f * C * z);
	dx += g[7] * (-B * z);           dz += g[7] * (-B * x);
	dx += g[8] * (2.0f * E * x);     dy += g[8] * (-2.0f * E * y);
	dx += g[9] * (-6.0f * F * x * y); dy += g[9] * (F * (-3.0f * x2 + 3.0f * y2));
	dx += g[10] * (G * y * z);       dy += g[10] * (G * x * z);        dz += g[10] * (G * x * y);
	dy += g[11] * (Hh * (1.0f - 5.0f * z2)); dz += g[11] * (-10.0f * Hh * y * z);
	dz += g[12] * (K * (15.0f * z2 - 3.0f));
	dx += g[13] * (Hh * (1.0f - 5.0f * z2)); dz += g[13] * (-10.0f * Hh * x * z);
	dx += g[14] * (2.0f * M * x * z); dy += g[14] * (-2.0f * M * y * z); dz += g[14] * (M * (x2 - y2));
	dx += g[15] * (F * (-3.0f * x2 + 3.0f * y2)); dy += g[15] * (6.0f * F * x * y);
	out[0] = 2.0f * dx; out[1] = 2.0f * dy; out[2] = 2.0f * dz;
}

void orc_nerf_input_gradient(const orc_net* net, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, const uint16_t* dL_dout,
                             float* dL_dinput /* [n][6] */) {
	const uint16_t* grid = params + GRID_OFF;
	#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		const float* coord = coords + (size_t)i * coord_stride_floats;
		orc_act a;
		orc_nerf_forward_one(net, params, coord, &a);
		uint16_t d_out[16]; memset(d_out, 0, sizeof(d_out));
		for (int c = 0; c < 3; ++c) d_out[c] = dL_dout[(size_t)i * 4 + c];
		uint16_t d_h3[64], d_h2[64], d_in[32], d_h1[64], d_x[32];
		orc_dense_bwd_input(params + W5_OFF, 16, 64, d_out, a.h3, d_h3);
		orc_dense_bwd_input(params + W4_OFF, 64, 64, d_h3, a.h2, d_h2);
		orc_dense_bwd_input(params + W3_OFF, 64, 32, d_h2, NULL, d_in);
		d_in[0] = orc_f2h(orc_h2f(d_in[0]) + orc_h2f(dL_dout[(size_t)i * 4 + 3]));
		orc_dense_bwd_input(params + W2_OFF, 16, 64, d_in, a.h1, d_h1);
		orc_dense_bwd_input(params + W1_OFF, 64, 32, d_h1, NULL, d_x);
		float gp[3] = {0.0f, 0.0f, 0.0f};
		for (uint32_t l = 0; l < net->n_levels; ++l) {
			const orc_grid_level* lv = &net->levels[l];
			float w[3]; uint32_t pg[3];
			for (int d = 0; d < 3; ++d) {
				float p = fmaf(lv->scale, coord[d], 0.5f);
				float fl = floorf(p);
				pg[d] = (uint32_t)(int)fl;
				w[d] = p - fl;
			}
			const float g0 = orc_h2f(d_x[2 * l]), g1 = orc_h2f(d_x[2 * l + 1]);
			for (int gd = 0; gd < 3; ++gd) {
				float dy0 = 0.0f, dy1 = 0.0f;   /* dy/dx[gd] of the level's two features */
				for (uint32_t idx = 0; idx < 4; ++idx) {
					float weight = lv->scale; uint32_t c[3];
					for (int nd = 0; nd < 2; ++nd) {
						const int dim = nd >= gd ? nd + 1 : nd;
						if ((idx & (1u << nd)) == 0) { weight *= 1.0f - w[dim]; c[dim] = pg[dim]; }
						else { weight *= w[dim]; c[dim] = pg[dim] + 1; }
					}
					c[gd] = pg[gd];
					const size_t lo = 2u * ((size_t)lv->offset + orc_grid_index(lv, c[0], c[1], c[2]));
					c[gd] = pg[gd] + 1;
					const size_t hi = 2u * ((size_t)lv->offset + orc_grid_index(lv, c[0], c[1], c[2]));
					dy0 += weight * (orc_h2f(grid[hi]) - orc_h2f(grid[lo]));
					dy1 += weight * (orc_h2f(grid[hi + 1]) - orc_h2f(grid[lo + 1]));
				}
				gp[gd] += g0 * dy0;
				gp[gd] += g1 * dy1;
			}
		}
		float gsh[16], gd3[3];
		for (int k = 0; k < 16; ++k) gsh[k] = orc_h2f(d_in[16 + k]);
		orc_sh4_grad(coord + 4, gsh, gd3);
		float* o = dL_dinput + (size_t)i * 6;
		o[0] = gp[0]; o[1] = gp[1]; o[2] = gp[2]; o[3] = gd3[0]; o[4] = gd3[1]; o[5] = gd3[2];
	}
}

/* nerf_network.h:396-441 initialize_params order: density MLP, rgb MLP, pos grid (dir enc has none).
 * [tcnn] Xavier-uniform for matrices (scale sqrt(6/(fan_in+fan_out))), U(-1e-4,1e-4) for the grid, drawn from the
 * Trainer's pcg32(seed) (testbed.cu:2445).  tcnn's generator kernel interleaves streams across threads in a way
 * that cannot be verified here; this restatement draws element k from the k-th next_float() of one stream. */
void orc_nerf_init_params(const orc_net* net, uint64_t seed, float* params_fp32) {
	orc_pcg32 rng = orc_pcg32_make(seed);
	const uint32_t dims[5][2] = {{64, 32}, {16, 64}, {64, 32}, {64, 64}, {16, 64}};
	uint32_t off = 0;
	for (int m = 0; m < 5; ++m) {
		float scale = sqrtf(6.0f / (float)(dims[m][0] + dims[m][1]));
		uint32_t cnt = dims[m][0] * dims[m][1];
		for (uint32_t k = 0; k < cnt; ++k) params_fp32[off + k] = orc_pcg32_next_float(&rng) * (scale - (-scale)) + (-scale);
		off += cnt;
	}
	uint32_t ng = 2u * net->n_grid_entries;
	for (uint32_t k = 0; k < ng; ++k) params_fp32[off + k] = orc_pcg32_next_float(&rng) * (1e-4f - (-1e-4f)) + (-1e-4f);
}

void orc_f32_to_f16(const float* in, uint16_t* out, uint32_t n) { for (uint32_t i = 0; i < n; ++i) out[i] = orc_f2h(in[i]); }
void orc_f16_to_f32(const uint16_t* in, float* out, uint32_t n) { for (uint32_t i = 0; i < n; ++i) out[i] = orc_h2f(in[i]); }

/* [tcnn] optimizers/adam.h + exponential_decay.h + ema.h as configured by configs/nerf/base.json:5-22,
 * driven by Trainer::optimizer_step(stream, loss_scale=128) (testbed_nerf.cu:2950).
 *   g = grad/loss_scale; encoding params with g == 0 are skipped; matrix params get g += l2_reg*w;
 *   m = b1*m+(1-b1)g; v = b2*v+(1-b2)g^2; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps);
 *   ExponentialDecay: lr *= decay_base whenever step >= decay_start and (step - decay_start) % decay_interval == 0
 *   Ema: ema = (ema*decay*(1-decay^(t-1)) + w_fp16*(1-decay)) / (1-decay^t), inference params = fp16(ema).
 * `step` is the 1-based count of optimizer steps including this one. */
void orc_adam_ema_step(uint32_t n_params, uint32_t n_matrix_params, uint32_t step, float base_lr_after_decay, float beta1, float beta2,
                       float epsilon, float l2_reg, float loss_scale, float ema_decay, const uint16_t* grads_fp16, float* master,
                       uint16_t* params_fp16, float* m1, float* m2, float* ema_fp32, uint16_t* inference_fp16) {
	float lr = base_lr_after_decay * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
	float ema_debias_old = 1.0f - powf(ema_decay, (float)(step - 1));
	float ema_debias_new = 1.0f / (1.0f - powf(ema_decay, (float)step));
	for (uint32_t i = 0; i < n_params; ++i) {
		float g = orc_h2f(grads_fp16[i]) / loss_scale;
		int skip = (i >= n_matrix_params) && g == 0.0f;
		if (!skip) {
			float w = master[i];
			if (i < n_matrix_params) g += l2_reg * w;
			float gsq = g * g;
			float fm = m1[i] = beta1 * m1[i] + (1.0f - beta1) * g;
			float sm = m2[i] = beta2 * m2[i] + (1.0f - beta2) * gsq;
			float eff = lr / (sqrtf(sm) + epsilon);
			float nw = w - eff * fm;
			master[i] = nw;
			params_fp16[i] = orc_f2h(nw);
		}
		float filtered = (ema_fp32[i] * ema_decay * ema_debias_old + orc_h2f(params_fp16[i]) * (1.0f - ema_decay)) * ema_debias_new;
		ema_fp32[i] = filtered;
		inference_fp16[i] = orc_f2h(filtered);
	}
}

/* ================================================================================================================================
 * Plumbing configs P1 (2-D image) / P2 (SDF): tcnn NetworkWithInputEncoding = HashGrid(n_dims in {2,3}, L=16, F=2) -> FullyFusedMLP
 * 32 -> 64 -> 64 -> 16 (ReLU, no output activation), as Testbed::reset_network builds it for Image / Sdf mode (src/testbed.cu:2397-2445,
 * configs/image/base.json, configs/sdf/base.json).  Parameter order [tcnn NetworkWithInputEncoding::set_params]: network, then encoding.
 * Training = Trainer::training_step (forward, loss, backward) + optimizer_step(128) (src/testbed_image.cu:277-288, src/testbed_sdf.cu:1229-1252). */
#define GM_L0_OFF 0u
#define GM_L1_OFF (64u * 32u)
#define GM_L2_OFF (GM_L1_OFF + 64u * 64u)
#define GM_GRID_OFF (GM_L2_OFF + 16u * 64u)   /* 7168 */

void orc_gridmlp_make_levels(uint32_t n_dims, uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale,
                             orc_grid_level* levels, uint32_t* n_grid_entries) {
	uint32_t offset = 0;
	float log2_pls = log2f(per_level_scale);
	for (uint32_t l = 0; l < n_levels; ++l) {
		float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
		uint32_t res = (uint32_t)ceilf(scale) + 1u;
		uint64_t dense = 1;
		for (uint32_t d = 0; d < n_dims; ++d) dense *= res;
		uint32_t max_params = 0xffffffffu / 2u;
		uint32_t n = dense > max_params ? max_params : (uint32_t)dense;
		n = (n + 7u) / 8u * 8u;
		uint32_t hashmap = 1u << log2_hashmap_size;
		uint32_t size = n < hashmap ? n : hashmap;
		levels[l].scale = scale; levels[l].resolution = res; levels[l].offset = offset; levels[l].size = size;
		offset += size;
	}
	*n_grid_entries = offset;
}
uint32_t orc_gridmlp_n_params(const orc_net* net) { return GM_GRID_OFF + 2u * net->n_grid_entries; }

/* [tcnn] grid_index<N_DIMS> with the primes {1, 2654435761, 805459861} */
static inline uint32_t orc_grid_index_nd(uint32_t n_dims, const orc_grid_level* lv, const uint32_t p[3]) {
	static const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
	uint32_t stride = 1, index = 0;
	for (uint32_t d = 0; d < n_dims && stride <= lv->size; ++d) { index += p[d] * stride; stride *= lv->resolution; }
	if (lv->size < stride) { index = 0; for (uint32_t d = 0; d < n_dims; ++d) index ^= p[d] * primes[d]; }
	return index % lv->size;
}

void orc_grid_encode_nd(uint32_t n_dims, const orc_net* net, const uint16_t* grid, const float* pos_in, uint16_t* out) {
	for (uint32_t l = 0; l < net->n_levels; ++l) {
		const orc_grid_level* lv = &net->levels[l];
		float pos[3] = {0, 0, 0}; uint32_t pg[3] = {0, 0, 0};
		for (uint32_t d = 0; d < n_dims; ++d) {
			float p = fmaf(lv->scale, pos_in[d], 0.5f);   /* [tcnn] pos_fract: fmaf(scale, input, 0.5f) */
			float fl = floorf(p);
			pg[d] = (uint32_t)(int)fl;
			pos[d] = p - fl;
		}
		float r0 = 0.0f, r1 = 0.0f;
		for (uint32_t idx = 0; idx < (1u << n_dims); ++idx) {
			float w = 1.0f; uint32_t c[3] = {0, 0, 0};
			for (uint32_t d = 0; d < n_dims; ++d) {
				if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; c[d] = pg[d]; }
				else { w *= pos[d]; c[d] = pg[d] + 1; }
			}
			const uint16_t* v = grid + 2u * ((size_t)lv->offset + orc_grid_index_nd(n_dims, lv, c));
			r0 += w * orc_h2f(v[0]);
			r1 += w * orc_h2f(v[1]);
		}
		out[2 * l + 0] = orc_f2h(r0);
		out[2 * l + 1] = orc_f2h(r1);
	}
}

typedef struct { uint16_t x[32], h1[64], h2[64], out[16]; } orc_gm_act;
static void orc_gridmlp_forward_one(uint32_t n_dims, const orc_net* net, const uint16_t* params, const float* pos, orc_gm_act* a) {
	orc_grid_encode_nd(n_dims, net, params + GM_GRID_OFF, pos, a->x);
	orc_dense(params + GM_L0_OFF, 64, 32, a->x, a->h1, 1);
	orc_dense(params + GM_L1_OFF, 64, 64, a->h1, a->h2, 1);
	orc_dense(params + GM_L2_OFF, 16, 64, a->h2, a->out, 0);
}

void orc_gridmlp_inference(uint32_t n_dims, const orc_net* net, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n, uint16_t* out, uint32_t out_stride) {
	#pragma omp parallel for schedule(static) if (n >= 512)
	for (uint32_t i = 0; i < n; ++i) {
		orc_gm_act a;
		orc_gridmlp_forward_one(n_dims, net, params, pos + (size_t)i * pos_stride_floats, &a);
		for (uint32_t c = 0; c < 4; ++c) out[(size_t)i * out_stride + c] = a.out[c];
	}
}

/* [tcnn] kernel_grid_backward (encodings/grid.h), restated with an exact sum: every term half(w * dL/dx) is a multiple of 2^-24 below 2^16,
 * so a 64-bit integer holds the sum of all terms of an entry exactly; it is rounded to fp16 once (nearest even).  tcnn itself rounds after
 * every atomicAdd(half2) in whatever order the GPU schedules them, which no restatement can pin; this is the order-independent answer those
 * sums scatter around.  Non-finite terms are dropped (the device does the same; such a step is skipped by the loss scaler anyway). */
static uint16_t orc_fixed24_to_half(long long a) {
	/* exact integer a * 2^-24 -> fp16: round to odd at 24 significant bits, then orc_f2h rounds nearest-even to 11 */
	unsigned long long m = a < 0 ? 0ull - (unsigned long long)a : (unsigned long long)a;
	float f;
	if (m < (1ull << 24)) f = (float)(uint32_t)m;
	else {
		int top_bit = 63; while (!((m >> top_bit) & 1ull)) --top_bit;
		const int sh = top_bit - 23;
		unsigned long long top = m >> sh;
		if (m & ((1ull << sh) - 1ull)) top |= 1ull;
		f = ldexpf((float)(uint32_t)top, sh);
	}
	f *= 1.0f / 16777216.0f;
	return orc_f2h(a < 0 ? -f : f);
}

void orc_grid_backward_exact(uint32_t n_dims, const orc_net* net, const float* pos_all, uint32_t pos_stride_floats, uint32_t n, const uint16_t* dL_dx_planes, uint16_t* grid_grad) {
	const orc_grid_level* last = &net->levels[net->n_levels - 1];
	const size_t n_entries = (size_t)last->offset + last->size;
	long long* acc = (long long*)calloc(2 * n_entries, sizeof(long long));
	for (uint32_t l = 0; l < net->n_levels; ++l) {
		const orc_grid_level* lv = &net->levels[l];
		for (uint32_t i = 0; i < n; ++i) {
			const float* pos_in = pos_all + (size_t)i * pos_stride_floats;
			float pos[3] = {0, 0, 0}; uint32_t pg[3] = {0, 0, 0};
			for (uint32_t d = 0; d < n_dims; ++d) {
				float p = fmaf(lv->scale, pos_in[d], 0.5f);   /* [tcnn] pos_fract: fmaf(scale, input, 0.5f) */
				float fl = floorf(p);
				pg[d] = (uint32_t)(int)fl;
				pos[d] = p - fl;
			}
			const float g0 = orc_h2f(dL_dx_planes[((size_t)l * n + i) * 2]), g1 = orc_h2f(dL_dx_planes[((size_t)l * n + i) * 2 + 1]);
			for (uint32_t idx = 0; idx < (1u << n_dims); ++idx) {
				float w = 1.0f; uint32_t c[3] = {0, 0, 0};
				for (uint32_t d = 0; d < n_dims; ++d) {
					if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; c[d] = pg[d]; }
					else { w *= pos[d]; c[d] = pg[d] + 1; }
				}
				const size_t k = 2u * ((size_t)lv->offset + orc_grid_index_nd(n_dims, lv, c));
				const float t0 = orc_rh(w * g0), t1 = orc_rh(w * g1);
				if (fabsf(t0) < 65520.0f) acc[k] += (long long)((double)t0 * 16777216.0);
				if (fabsf(t1) < 65520.0f) acc[k + 1] += (long long)((double)t1 * 16777216.0);
			}
		}
	}
	for (size_t k = 0; k < 2 * n_entries; ++k) grid_grad[k] = orc_fixed24_to_half(acc[k]);
	free(acc);
}

/* forward + backward over n samples; dL_dout fp16 [n][4]; grads_out double [n_params]; optional per-sample outputs / encodings / dL/dx */
void orc_gridmlp_forward_backward(uint32_t n_dims, const orc_net* net, const uint16_t* params, const float* pos_all, uint32_t pos_stride_floats, uint32_t n, const uint16_t* dL_dout,
                                  uint16_t* out4 /* [n][4] or NULL */, double* grads_out, uint16_t* dL_dx_out /* [n][32] or NULL */) {
	const uint32_t np = orc_gridmlp_n_params(net);
	for (uint32_t k = 0; k < np; ++k) grads_out[k] = 0.0;
	#pragma omp parallel
	{
	double* mlp_acc = (double*)calloc(GM_GRID_OFF, sizeof(double));
	#pragma omp for schedule(static)
	for (uint32_t i = 0; i < n; ++i) {
		const float* pos_in = pos_all + (size_t)i * pos_stride_floats;
		orc_gm_act a;
		orc_gridmlp_forward_one(n_dims, net, params, pos_in, &a);
		if (out4) for (int c = 0; c < 4; ++c) out4[(size_t)i * 4 + c] = a.out[c];
		uint16_t d_out[16]; memset(d_out, 0, sizeof(d_out));
		for (int c = 0; c < 4; ++c) d_out[c] = dL_dout[(size_t)i * 4 + c];
		uint16_t d_h2[64], d_h1[64], d_x[32];
		orc_dense_bwd_input(params + GM_L2_OFF, 16, 64, d_out, a.h2, d_h2);
		orc_dense_bwd_input(params + GM_L1_OFF, 64, 64, d_h2, a.h1, d_h1);
		orc_dense_bwd_input(params + GM_L0_OFF, 64, 32, d_h1, NULL, d_x);
		if (dL_dx_out) memcpy(dL_dx_out + (size_t)i * 32, d_x, sizeof(d_x));
#define WGRAD(OFF, NOUT, NIN, DY, X) \
		for (uint32_t o = 0; o < (NOUT); ++o) { float dy_ = orc_h2f((DY)[o]); if (dy_ != 0.0f) for (uint32_t k = 0; k < (NIN); ++k) mlp_acc[(OFF) + o * (NIN) + k] += (double)(dy_ * orc_h2f((X)[k])); }
		WGRAD(GM_L2_OFF, 16, 64, d_out, a.h2)
		WGRAD(GM_L1_OFF, 64, 64, d_h2, a.h1)
		WGRAD(GM_L0_OFF, 64, 32, d_h1, a.x)
#undef WGRAD
		for (uint32_t l = 0; l < net->n_levels; ++l) {
			const orc_grid_level* lv = &net->levels[l];
			float pos[3] = {0, 0, 0}; uint32_t pg[3] = {0, 0, 0};
			for (uint32_t d = 0; d < n_dims; ++d) {
				float p = fmaf(lv->scale, pos_in[d], 0.5f);   /* [tcnn] pos_fract: fmaf(scale, input, 0.5f) */
				float fl = floorf(p);
				pg[d] = (uint32_t)(int)fl;
				pos[d] = p - fl;
			}
			float g0 = orc_h2f(d_x[2 * l]), g1 = orc_h2f(d_x[2 * l + 1]);
			for (uint32_t idx = 0; idx < (1u << n_dims); ++idx) {
				float w = 1.0f; uint32_t c[3] = {0, 0, 0};
				for (uint32_t d = 0; d < n_dims; ++d) {
					if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; c[d] = pg[d]; }
					else { w *= pos[d]; c[d] = pg[d] + 1; }
				}
				size_t k = GM_GRID_OFF + 2u * ((size_t)lv->offset + orc_grid_index_nd(n_dims, lv, c));
				const double a0 = (double)orc_rh(w * g0), a1 = (double)orc_rh(w * g1);
				#pragma omp atomic
				grads_out[k + 0] += a0;
				#pragma omp atomic
				grads_out[k + 1] += a1;
			}
		}
	}
	#pragma omp critical
	for (uint32_t k = 0; k < GM_GRID_OFF; ++k) grads_out[k] += mlp_acc[k];
	free(mlp_acc);
	}
}

void orc_gridmlp_init_params(const orc_net* net, uint64_t seed, float* params_fp32) {
	orc_pcg32 rng = orc_pcg32_make(seed);
	const uint32_t dims[3][2] = {{64, 32}, {64, 64}, {16, 64}};
	uint32_t off = 0;
	for (int m = 0; m < 3; ++m) {
		float scale = sqrtf(6.0f / (float)(dims[m][0] + dims[m][1]));
		uint32_t cnt = dims[m][0] * dims[m][1];
		for (uint32_t k = 0; k < cnt; ++k) params_fp32[off + k] = orc_pcg32_next_float(&rng) * (scale - (-scale)) + (-scale);
		off += cnt;
	}
	uint32_t ng = 2u * net->n_grid_entries;
	for (uint32_t k = 0; k < ng; ++k) params_fp32[off + k] = orc_pcg32_next_float(&rng) * (1e-4f - (-1e-4f)) + (-1e-4f);
}

/* [tcnn] losses/{l2, relative_l2, l1, mape}.h (unverified recall): n_total = n * dims; value = f(d) / n_total; gradient = loss_scale * f'(d) / n_total */
void orc_tcnn_loss_and_gradient(int loss_type, uint32_t n, uint32_t dims, float loss_scale, const uint16_t* predictions, uint32_t pred_stride, const float* targets,
                           float* values, uint16_t* gradients, uint32_t grad_stride) {
	const float n_total = (float)(n * dims);
	for (uint32_t s = 0; s < n; ++s) for (uint32_t c = 0; c < 4; ++c) {
		if (c >= dims) { gradients[(size_t)s * grad_stride + c] = 0; continue; }
		float prediction = orc_h2f(predictions[(size_t)s * pred_stride + c]);
		float target = targets[(size_t)s * dims + c];
		float difference = prediction - target, value, gradient;
		switch (loss_type) {
			case ORC_LOSS_L2: value = difference * difference; gradient = 2.0f * difference; break;
			case ORC_LOSS_RELATIVE_L2: { float pn = prediction * prediction + 0.01f; value = difference * difference / pn; gradient = 2.0f * difference / pn; break; }
			case ORC_LOSS_L1: value = fabsf(difference); gradient = copysignf(1.0f, difference); break;
			default: { float sc = 1.0f / (fabsf(target) + 0.01f); value = fabsf(difference) * sc; gradient = copysignf(sc, difference); break; }
		}
		values[(size_t)s * dims + c] = value / n_total;
		gradients[(size_t)s * grad_stride + c] = orc_f2h(loss_scale * gradient / n_total);
	}
}

/* [tcnn] generate_random_uniform: element k = k-th next_float() */
void orc_generate_random_uniform(uint64_t rng_state, uint64_t rng_inc, uint32_t n_elements, float* out) {
	orc_pcg32 rng; rng.state = rng_state; rng.inc = rng_inc;
	for (uint32_t k = 0; k < n_elements; ++k) out[k] = orc_pcg32_next_float(&rng);
}

/* src/testbed_image.cu:62-77 */
void orc_image_stratify2(uint32_t n_elements, uint32_t log2_batch_size, float* inout_xy) {
	uint32_t log2_size = log2_batch_size / 2, size = 1u << log2_size;
	for (uint32_t i = 0; i < n_elements; ++i) {
		uint32_t in_batch_index = i & ((1u << log2_batch_size) - 1u);
		uint32_t x = in_batch_index & ((1u << log2_size) - 1u), y = in_batch_index >> log2_size;
		float vx = inout_xy[2 * i], vy = inout_xy[2 * i + 1];
		inout_xy[2 * i] = vx / (float)size + ((float)x / (float)size);
		inout_xy[2 * i + 1] = vy / (float)size + ((float)y / (float)size);
	}
}

/* src/testbed_image.cu:172-218 eval_image_kernel_and_snap<T, stride>; image_data_type 2 = fp16 RGBA, 3 = fp32 RGBA */
void orc_image_eval_and_snap(uint32_t n_elements, const void* texture, int image_data_type, float* positions_xy, const int32_t res[2], float* result, uint32_t stride,
                             int snap_to_pixel_centers, int linear_colors) {
	const int rx = res[0], ry = res[1];
	for (uint32_t i = 0; i < n_elements; ++i) {
		float px = positions_xy[2 * i], py = positions_xy[2 * i + 1];
		float val[4] = {0, 0, 0, 0};
		float texel[4][4];
		int coords[4][2], n_tex;
		float wts[4] = {1.0f, 0.0f, 0.0f, 0.0f};
		if (snap_to_pixel_centers) {
			int ix = (int)floorf(px * (float)rx), iy = (int)floorf(py * (float)ry);
			positions_xy[2 * i] = ((float)ix + 0.5f) / (float)rx; positions_xy[2 * i + 1] = ((float)iy + 0.5f) / (float)ry;
			ix = ix < 0 ? 0 : (ix > rx - 1 ? rx - 1 : ix); iy = iy < 0 ? 0 : (iy > ry - 1 ? ry - 1 : iy);
			coords[0][0] = ix; coords[0][1] = iy; wts[0] = 1.0f; n_tex = 1;
		} else {
			px = fminf(fmaxf(px * (float)rx - 0.5f, 0.0f), (float)rx - (1.0f + 1e-4f));
			py = fminf(fmaxf(py * (float)ry - 0.5f, 0.0f), (float)ry - (1.0f + 1e-4f));
			int ix = (int)px, iy = (int)py;
			float wx = px - (float)ix, wy = py - (float)iy;
			int x0 = ix < rx - 2 ? ix : rx - 2; if (x0 < 0) x0 = 0;
			int y0 = iy < ry - 2 ? iy : ry - 2; if (y0 < 0) y0 = 0;
			coords[0][0] = x0; coords[0][1] = y0; coords[1][0] = x0 + 1; coords[1][1] = y0; coords[2][0] = x0; coords[2][1] = y0 + 1; coords[3][0] = x0 + 1; coords[3][1] = y0 + 1;
			wts[0] = (1 - wx) * (1 - wy); wts[1] = wx * (1 - wy); wts[2] = (1 - wx) * wy; wts[3] = wx * wy;
			n_tex = 4;
		}
		for (int t = 0; t < n_tex; ++t) {
			size_t e = ((size_t)coords[t][1] * rx + coords[t][0]) * 4;
			for (int k = 0; k < 4; ++k) texel[t][k] = image_data_type == 2 ? orc_h2f(((const uint16_t*)texture)[e + k]) : ((const float*)texture)[e + k];
			if (!linear_colors) for (int k = 0; k < 3; ++k) texel[t][k] = orc_linear_to_srgb(texel[t][k]);
		}
		if (n_tex == 1) { for (int k = 0; k < 4; ++k) val[k] = texel[0][k]; }
		else for (int k = 0; k < 4; ++k) val[k] = ((wts[0] * texel[0][k] + wts[1] * texel[1][k]) + wts[2] * texel[2][k]) + wts[3] * texel[3][k];
		float* r = result + (size_t)i * stride;
		r[0] = val[0]; r[1] = val[1]; r[2] = val[2];
		for (uint32_t k = 3; k < stride; ++k) r[k] = 1.0f;
	}
}
