/* orc_netx.c — TEST INFRASTRUCTURE (CPU oracle; never linked into the product).  NerfNetwork as the reference BUILDS it for configs other than
 * configs/nerf/base.json: include/neural-graphics-primitives/nerf_network.h:76-101 (constructor), 103-266 (inference / forward / backward) with
 *   * n_extra_dims > 0 — per-image latent codes / light directions (src/testbed_nerf.cu:2297-2338, nerf_loader.h:94-99): the direction encoding is the
 *     Composite of configs/nerf/base.json:37-51, SphericalHarmonics(degree 4) on the 3 direction dims + Identity on the extra dims, so the colour network's
 *     input is [density output 16 | SH 16 | extra dims, zero-padded to 16] = 48 wide (next_multiple(16 + n_extra, 16) + 16, nerf_network.h:82-93);
 *   * rgb_network.n_hidden_layers in 0..3 (configs/nerf/base_{0,1,2,3}layer.json): 0 = one [16][in] matrix without activation ([tcnn] CutlassMLP with no
 *     hidden layer), h >= 1 = [64][in] -> (h - 1) x [64][64] -> [16][64], ReLU on the hidden layers ([tcnn] FullyFusedMLP).
 * Parameter order as in orc_network.c (nerf_network.h:361-394): density MLP, colour MLP (its matrices in layer order), grid.  Rounding points as there: fp16
 * storage of every activation / delta, fp32 accumulation, sequential sums.  parity unpinned for the same reason as orc_network.c (tiny-cuda-nn is absent). */
#include "ngp_oracle.h"
#include "orc_core.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint32_t x_rgb_in(const orc_netx* x) { return 16u + ((16u + x->n_extra_dims + 15u) / 16u) * 16u; }   /* 32, or 48 with 1..16 extra dims */

/* the colour network's matrices: out[m] x in[m], relu[m] on the matrix's output; returns their count */
static uint32_t x_rgb_layers(const orc_netx* x, uint32_t n_out[5], uint32_t n_in[5], int relu[5]) {
	const uint32_t in = x_rgb_in(x), h = x->n_rgb_hidden_layers;
	if (h == 0) { n_out[0] = 16; n_in[0] = in; relu[0] = 0; return 1; }
	uint32_t m = 0;
	n_out[m] = 64; n_in[m] = in; relu[m] = 1; ++m;
	for (uint32_t k = 1; k < h; ++k) { n_out[m] = 64; n_in[m] = 64; relu[m] = 1; ++m; }
	n_out[m] = 16; n_in[m] = 64; relu[m] = 0; ++m;
	return m;
}

uint32_t orc_netx_mlp_params(const orc_netx* x) {
	uint32_t no[5], ni[5]; int r[5];
	const uint32_t m = x_rgb_layers(x, no, ni, r);
	uint32_t n = 64u * 32u + 16u * 64u;
	for (uint32_t k = 0; k < m; ++k) n += no[k] * ni[k];
	return n;
}
uint32_t orc_netx_n_params(const orc_net* net, const orc_netx* x) { return orc_netx_mlp_params(x) + 2u * net->n_grid_entries; }

static void x_dense(const uint16_t* W, uint32_t n_out, uint32_t n_in, const uint16_t* in, uint16_t* out, int relu) {
	for (uint32_t o = 0; o < n_out; ++o) {
		float acc = 0.0f;
		for (uint32_t i = 0; i < n_in; ++i) acc += orc_h2f(W[o * n_in + i]) * orc_h2f(in[i]);
		if (relu && acc < 0.0f) acc = 0.0f;
		out[o] = orc_f2h(acc);
	}
}
static void x_dense_bwd(const uint16_t* W, uint32_t n_out, uint32_t n_in, const uint16_t* dy, const uint16_t* fwd_act, uint16_t* dx) {
	for (uint32_t i = 0; i < n_in; ++i) {
		float acc = 0.0f;
		for (uint32_t o = 0; o < n_out; ++o) acc += orc_h2f(W[o * n_in + i]) * orc_h2f(dy[o]);
		if (fwd_act && !(orc_h2f(fwd_act[i]) > 0.0f)) acc = 0.0f;
		dx[i] = orc_f2h(acc);
	}
}

typedef struct { uint16_t x[32], h1[64], rin[48], act[5][64]; } x_act;   /* act[m] = output of colour matrix m (the last one: 16 padded outputs) */

static const float* x_extra_of(const orc_netx* x, uint32_t sample) {
	if (!x->n_extra_dims || !x->extra_dims) return NULL;
	return x->extra_dims + (size_t)(x->sample_slot ? x->sample_slot[sample] : 0u) * x->n_extra_dims;
}

static void x_forward_one(const orc_net* net, const orc_netx* x, const uint16_t* params, const float* coord, const float* extra, x_act* a) {
	const uint32_t n_mlp = orc_netx_mlp_params(x);
	orc_grid_encode_one(net, params + n_mlp, coord, a->x);
	x_dense(params, 64, 32, a->x, a->h1, 1);
	x_dense(params + 64 * 32, 16, 64, a->h1, a->rin, 0);
	float sh[16];
	orc_sh4(coord + 4, sh);
	for (int i = 0; i < 16; ++i) a->rin[16 + i] = orc_f2h(sh[i]);
	const uint32_t in = x_rgb_in(x);
	for (uint32_t i = 32; i < in; ++i) a->rin[i] = (i - 32 < x->n_extra_dims && extra) ? orc_f2h(extra[i - 32]) : 0;   /* [tcnn] Identity: the input cast to the network precision; padding 0 */
	uint32_t no[5], ni[5]; int relu[5];
	const uint32_t m = x_rgb_layers(x, no, ni, relu);
	uint32_t off = 64 * 32 + 16 * 64;
	const uint16_t* cur = a->rin;
	for (uint32_t k = 0; k < m; ++k) { x_dense(params + off, no[k], ni[k], cur, a->act[k], relu[k]); cur = a->act[k]; off += no[k] * ni[k]; }
}

void orc_nerf_inference_x(const orc_net* net, const orc_netx* x, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, uint16_t* out, uint32_t out_stride) {
	uint32_t no[5], ni[5]; int relu[5];
	const uint32_t m = x_rgb_layers(x, no, ni, relu);
	#pragma omp parallel for schedule(static) if (n >= 512)
	for (uint32_t i = 0; i < n; ++i) {
		x_act a;
		x_forward_one(net, x, params, coords + (size_t)i * coord_stride_floats, x_extra_of(x, i), &a);
		for (uint32_t c = 0; c < 3; ++c) out[(size_t)i * out_stride + c] = a.act[m - 1][c];
		out[(size_t)i * out_stride + 3] = a.rin[0];   /* extract_density (nerf_network.h:32-43, 130-136) */
	}
}

/* forward + backward (nerf_network.h:143-266); grads_out: double [n_params]; dL_dx_out: fp16 [n][32] or NULL; dL_dextra_out: float [n][n_extra_dims] or NULL —
 * the extra-dim rows of the network's dL_dinput matrix ([tcnn] Identity backward: the fp16 gradient of its output, as float), what
 * compute_extra_dims_gradient_train_nerf sums per image (src/testbed_nerf.cu:1710-1746) */
void orc_nerf_forward_backward_x(const orc_net* net, const orc_netx* x, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, const uint16_t* dL_dout,
                                 uint16_t* out_rgbsigma, double* grads_out, uint16_t* dL_dx_out, float* dL_dextra_out) {
	const uint32_t n_mlp = orc_netx_mlp_params(x), np = orc_netx_n_params(net, x), in = x_rgb_in(x);
	uint32_t no[5], ni[5]; int relu[5];
	const uint32_t m = x_rgb_layers(x, no, ni, relu);
	for (uint32_t k = 0; k < np; ++k) grads_out[k] = 0.0;
	#pragma omp parallel
	{
	double* acc = (double*)calloc(n_mlp, sizeof(double));
	#pragma omp for schedule(static)
	for (uint32_t s = 0; s < n; ++s) {
		const float* coord = coords + (size_t)s * coord_stride_floats;
		x_act a;
		x_forward_one(net, x, params, coord, x_extra_of(x, s), &a);
		if (out_rgbsigma) { for (int c = 0; c < 3; ++c) out_rgbsigma[(size_t)s * 4 + c] = a.act[m - 1][c]; out_rgbsigma[(size_t)s * 4 + 3] = a.rin[0]; }
		/* colour network, last matrix to first */
		uint16_t dy[64], dprev[64];
		memset(dy, 0, sizeof(dy));
		for (int c = 0; c < 3; ++c) dy[c] = dL_dout[(size_t)s * 4 + c];
		uint32_t offs[5], off = 64 * 32 + 16 * 64;
		for (uint32_t k = 0; k < m; ++k) { offs[k] = off; off += no[k] * ni[k]; }
		uint16_t d_in[48];
		for (int k = (int)m - 1; k >= 0; --k) {
			const uint16_t* h_in = k == 0 ? a.rin : a.act[k - 1];
			for (uint32_t o = 0; o < no[k]; ++o) { const float d = orc_h2f(dy[o]); if (d != 0.0f) for (uint32_t i = 0; i < ni[k]; ++i) acc[offs[k] + o * ni[k] + i] += (double)(d * orc_h2f(h_in[i])); }
			if (k == 0) x_dense_bwd(params + offs[k], no[k], ni[k], dy, NULL, d_in);
			else { x_dense_bwd(params + offs[k], no[k], ni[k], dy, a.act[k - 1], dprev); memcpy(dy, dprev, sizeof(dy)); }
		}
		if (dL_dextra_out) for (uint32_t e = 0; e < x->n_extra_dims; ++e) dL_dextra_out[(size_t)s * x->n_extra_dims + e] = orc_h2f(d_in[32 + e]);
		d_in[0] = orc_f2h(orc_h2f(d_in[0]) + orc_h2f(dL_dout[(size_t)s * 4 + 3]));   /* add_density_gradient (nerf_network.h:63-74) */
		uint16_t d_h1[64], d_x[32];
		for (uint32_t o = 0; o < 16; ++o) { const float d = orc_h2f(d_in[o]); if (d != 0.0f) for (uint32_t i = 0; i < 64; ++i) acc[64 * 32 + o * 64 + i] += (double)(d * orc_h2f(a.h1[i])); }
		x_dense_bwd(params + 64 * 32, 16, 64, d_in, a.h1, d_h1);
		for (uint32_t o = 0; o < 64; ++o) { const float d = orc_h2f(d_h1[o]); if (d != 0.0f) for (uint32_t i = 0; i < 32; ++i) acc[o * 32 + i] += (double)(d * orc_h2f(a.x[i])); }
		x_dense_bwd(params, 64, 32, d_h1, NULL, d_x);
		if (dL_dx_out) memcpy(dL_dx_out + (size_t)s * 32, d_x, sizeof(d_x));
		(void)in;
		/* [tcnn] kernel_grid_backward: grad[idx][f] += fp16(w * dL/dx[2l + f]) */
		for (uint32_t l = 0; l < net->n_levels; ++l) {
			const orc_grid_level* lv = &net->levels[l];
			float pos[3]; uint32_t pg[3];
			for (int d = 0; d < 3; ++d) { const float p = fmaf(lv->scale, coord[d], 0.5f), fl = floorf(p); pg[d] = (uint32_t)(int)fl; pos[d] = p - fl; }
			const float g0 = orc_h2f(d_x[2 * l]), g1 = orc_h2f(d_x[2 * l + 1]);
			for (uint32_t idx = 0; idx < 8; ++idx) {
				float w = 1.0f; uint32_t c[3];
				for (int d = 0; d < 3; ++d) { if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; c[d] = pg[d]; } else { w *= pos[d]; c[d] = pg[d] + 1; } }
				const uint32_t gi = orc_grid_index_export(lv, c[0], c[1], c[2]);
				const size_t k = n_mlp + 2u * ((size_t)lv->offset + gi);
				const double a0 = (double)orc_rh(w * g0), a1 = (double)orc_rh(w * g1);
				#pragma omp atomic
				grads_out[k + 0] += a0;
				#pragma omp atomic
				grads_out[k + 1] += a1;
			}
		}
	}
	#pragma omp critical
	for (uint32_t k = 0; k < n_mlp; ++k) grads_out[k] += acc[k];
	free(acc);
	}
}

/* nerf_network.h:396-441 + [tcnn] Xavier-uniform per matrix / U(-1e-4, 1e-4) for the grid: element k <- k-th draw of pcg32(seed), like orc_nerf_init_params */
void orc_nerf_init_params_x(const orc_net* net, const orc_netx* x, uint64_t seed, float* params_fp32) {
	orc_pcg32 rng = orc_pcg32_make(seed);
	uint32_t no[7], ni[7]; int relu[5];
	no[0] = 64; ni[0] = 32; no[1] = 16; ni[1] = 64;
	const uint32_t m = x_rgb_layers(x, no + 2, ni + 2, relu);
	uint32_t off = 0;
	for (uint32_t k = 0; k < m + 2; ++k) {
		const float scale = sqrtf(6.0f / (float)(no[k] + ni[k]));
		const uint32_t cnt = no[k] * ni[k];
		for (uint32_t e = 0; e < cnt; ++e) params_fp32[off + e] = orc_pcg32_next_float(&rng) * (scale - (-scale)) + (-scale);
		off += cnt;
	}
	const uint32_t ng = 2u * net->n_grid_entries;
	for (uint32_t e = 0; e < ng; ++e) params_fp32[off + e] = orc_pcg32_next_float(&rng) * (1e-4f - (-1e-4f)) + (-1e-4f);
}

/* compute_extra_dims_gradient_train_nerf (src/testbed_nerf.cu:1710-1746): per kept ray, the extra-dim gradients of its compacted samples summed into its image's row */
void orc_compute_extra_dims_gradient(uint32_t n_rays_alive, const uint32_t* ray_image /* [n_rays_alive] */, const uint32_t* numsteps /* compacted (count, base) pairs */,
                                     const float* dL_dextra /* [samples][n_extra] */, uint32_t n_extra_dims, float* gradient /* [n_images][n_extra], accumulated */) {
	for (uint32_t i = 0; i < n_rays_alive; ++i) {
		const uint32_t cnt = numsteps[2 * i], base = numsteps[2 * i + 1];
		for (uint32_t j = 0; j < cnt; ++j) for (uint32_t k = 0; k < n_extra_dims; ++k) gradient[(size_t)ray_image[i] * n_extra_dims + k] += dL_dextra[(size_t)(base + j) * n_extra_dims + k];
	}
}
