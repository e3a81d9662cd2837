// loss.hip — per-ray compositing, loss, dL/d(rgb,sigma) and sample compaction for gfx950.
// Replaces src/testbed_nerf.cu:1280-1597 (compute_loss_kernel_train_nerf) with the loss functions of 121-189 / 1263-1278 and
// tcnn's fill_rollover / fill_rollover_and_rescale (call sites 3314-3322).
// Default-off branches (envmap, error-map CDF sampling, sharpness, depth supervision, exposure gradient: testbed.h:651-680)
// are not implemented; the always-on error-map deposit (1465-1491) is.
//
// MI355X mapping: the reference runs ONE THREAD PER RAY walking its samples three times — ~16 k threads on a 131 k-thread
// chip, every load uncoalesced.  Here ONE WAVE owns a ray: lane j holds sample j (64 per chunk), the transmittance is a
// wave-wide exclusive prefix PRODUCT, the colour integrals are wave reductions / prefix sums, all sample loads and the
// compacted stores are coalesced, and the 16 rays of a workgroup share ONE atomic for their compaction slots.
// The transmittance product is therefore associated as a scan tree instead of left-to-right: counts that hinge on
// `T < 1e-4` can differ from a sequential evaluation only when T is within rounding of the threshold.
#include "ngp_device.cuh"

namespace ngp {

struct LG { float loss[3]; float grad[3]; };

__device__ __forceinline__ LG loss_and_gradient(const float t[3], const float p[3], int loss_type) {
	LG r;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const float diff = p[c] - t[c];
		switch (loss_type) {
			case NGP_LOSS_RELATIVE_L2: { const float f = 1.0f / (p[c] * p[c] + 1e-2f); r.loss[c] = diff * diff * f; r.grad[c] = 2.0f * diff * f; break; }
			case NGP_LOSS_L1: r.loss[c] = fabsf(diff); r.grad[c] = copysignf(1.0f, diff); break;
			case NGP_LOSS_MAPE: { const float f = 1.0f / (fabsf(p[c]) + 1e-2f); r.loss[c] = fabsf(diff) * f; r.grad[c] = copysignf(f, diff); break; }
			case NGP_LOSS_SMAPE: { const float f = 1.0f / (0.5f * (fabsf(p[c]) + fabsf(t[c])) + 1e-2f); r.loss[c] = fabsf(diff) * f; r.grad[c] = copysignf(f, diff); break; }
			case NGP_LOSS_HUBER: {
				const float alpha = 0.1f, ad = fabsf(diff);
				const float square = 0.5f / alpha * diff * diff;
				r.loss[c] = (ad > alpha ? (ad - 0.5f * alpha) : square) / 5.0f;
				r.grad[c] = (ad > alpha ? (diff > 0 ? 1.0f : -1.0f) : (diff / alpha)) / 5.0f;
				break; }
			case NGP_LOSS_LOG_L1: { const float d = fabsf(diff) + 1.0f; r.loss[c] = logf(d); r.grad[c] = copysignf(1.0f / d, diff); break; }
			default: r.loss[c] = diff * diff; r.grad[c] = 2.0f * diff; break;
		}
	}
	return r;
}

struct LossArgs {
	uint32_t n_rays; Aabb aabb; Pcg32 rng; uint32_t max_samples_compacted; const uint32_t* rays_counter; float loss_scale; uint32_t mlp_stride;
	float background_color[3]; int color_space; int train_with_random_bg_color; int train_in_linear_colors; uint32_t n_training_images;
	const NgpImageMeta* metadata; const uint16_t* network_output; uint32_t* numsteps_counter; const uint32_t* ray_indices_in;
	const NgpRay* rays_in; uint32_t* numsteps_in; const NgpCoord* coords_in; NgpCoord* coords_out; uint16_t* dloss_doutput; uint32_t dl_stride;
	int loss_type; float* loss_output; int max_level_rand_training; float* max_level_compacted; int rgb_activation; int density_activation;
	int snap_to_pixel_centers; float* error_map; int32_t error_map_res[2]; const float* mean_density; const float* exposure; float near_distance;
	ErrorMapCdf cdf;
	const uint16_t* encoded_in; uint16_t* encoded_out;   // optional: [sample][32] fp16 encoding rows carried through the compaction
	float depth_supervision_lambda; int depth_loss_type;  // testbed.h:654, 680 (off by default)
	float* exposure_gradient;                             // [n_images][3] or NULL (optimize_exposure off)
	const float* envmap_data; float* envmap_gradient; int32_t envmap_res[2]; int envmap_loss_type;   // 1289-1292: fp32 [h][w][4] (TrainableBuffer<4,2,float>) or NULL
	const float* sharpness_data; int32_t sharpness_res[2]; float* sharpness_grid;                    // 1321-1323 (include_sharpness_in_error) or NULL
	uint32_t* x_row_out;   // optional: slot k of the compacted batch <- the uncompacted sample it came from (instead of carrying 64-byte encoding rows: NgpLossExtras::x_row_index_out)
};

typedef uint16_t us4 __attribute__((ext_vector_type(4)));

// Wave-wide scans on DPP (row shifts inside the 16-lane rows, then the two row broadcasts): ~6 dependent VALU operations per scan instead of 6 trips through the LDS crossbar
// (`__shfl_up` = ds_bpermute).  The kernel is one wave per ray and a ray's 64-sample chunks follow each other: the scans' latency is its critical path.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float identity, float v) {   // v of the lane the control names; `identity` where there is none
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// inclusive prefix over the wave: OP 0 = product, 1 = sum
template <int OP>
__device__ __forceinline__ float wave_inclusive(float v) {
	const float id = OP == 0 ? 1.f : 0.f;
#define NGP_SCAN_STEP(CTRL, MASK) { const float t = dpp_take<CTRL, MASK>(id, v); v = OP == 0 ? v * t : v + t; }
	NGP_SCAN_STEP(0x111, 0xf)   // row_shr:1
	NGP_SCAN_STEP(0x112, 0xf)   // row_shr:2
	NGP_SCAN_STEP(0x114, 0xf)   // row_shr:4
	NGP_SCAN_STEP(0x118, 0xf)   // row_shr:8
	NGP_SCAN_STEP(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
	NGP_SCAN_STEP(0x143, 0xc)   // row_bcast:31 into rows 2 and 3
#undef NGP_SCAN_STEP
	return v;
}
__device__ __forceinline__ float wave_last(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ __forceinline__ float wave_sum(float v) { return wave_last(wave_inclusive<1>(v)); }
// the value of the lane below (wave_shr:1); `identity` in lane 0
__device__ __forceinline__ float wave_prev(float identity, float v) { return dpp_take<0x138, 0xf>(identity, v); }

// one wave per ray; 4 rays per workgroup: the two block barriers around the compaction scan make every wave wait for the slowest ray of
// its workgroup, and ~1000 workgroups of 16 rays quantise badly over 512 resident slots (measured: 16 -> 4 rays, step -3.4 %)
#ifndef NGP_LOSS_RAYS_PER_BLOCK
#define NGP_LOSS_RAYS_PER_BLOCK 4
#endif
constexpr int LOSS_RAYS_PER_BLOCK = NGP_LOSS_RAYS_PER_BLOCK;

__global__ void __launch_bounds__(LOSS_RAYS_PER_BLOCK * 64) compute_loss_kernel(const LossArgs a) {
	__shared__ uint32_t s_counts[LOSS_RAYS_PER_BLOCK];
	__shared__ uint32_t s_block_base;
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t i = blockIdx.x * LOSS_RAYS_PER_BLOCK + w;  // ray slot
	const bool active = i < *a.rays_counter;                  // wave-uniform
	const float EPSILON = 1e-4f;

	uint32_t numsteps = 0, base = 0, compacted = 0;
	float T_final = 1.f;
	float rgb_ray[3] = {0.f, 0.f, 0.f}, depth_ray = 0.f;
	const v3 ray_o_early = active ? ld3(a.rays_in[i].o) : mk(0.f, 0.f, 0.f);
	if (active) {
		numsteps = a.numsteps_in[i * 2 + 0];
		base = a.numsteps_in[i * 2 + 1];
	}
	const NgpCoord* __restrict__ ci = a.coords_in + base;
	const uint16_t* __restrict__ no = a.network_output + (size_t)base * a.mlp_stride;

	// ---- target colour: replay the ray generator's draws (1376-1423); wave-uniform.  Placed before pass 1 (the reference computes it after): its
	// dependent loads (ray index -> image metadata -> pixel) are then in flight while pass 1 runs; only the background term needs pass 1's result.
	float bg[3] = {0.f, 0.f, 0.f};
	float rgbtarget[3] = {0, 0, 0}, xy[2] = {0, 0}, max_level = 1.0f, sample_pdf = 1.0f, pixel_pdf = 1.0f, exposure_scale[3] = {1.f, 1.f, 1.f};
	uint32_t img = 0;
	int32_t img_res[2] = {1, 1};
	v3 ray_o = mk(0, 0, 0), env_dir = mk(0, 0, 1);
	if (active) {
		ray_o = ld3(a.rays_in[i].o);
		const uint32_t ray_idx = a.ray_indices_in[i];
		Pcg32 rng = a.rng;
		rng.advance((uint64_t)(uint32_t)(ray_idx * NGP_N_MAX_RANDOM_SAMPLES_PER_RAY));
		float img_pdf = 1.0f, xy_pdf = 1.0f;
		img = image_idx(ray_idx, a.n_rays, a.n_training_images, a.cdf.cdf_img, &img_pdf);
		const NgpImageMeta& md = a.metadata[img];
		img_res[0] = md.res[0]; img_res[1] = md.res[1];
		nerf_random_image_pos_training(rng, md.res, a.snap_to_pixel_centers, a.cdf, img, xy[0], xy[1], &xy_pdf);
		sample_pdf = img_pdf * xy_pdf; pixel_pdf = xy_pdf;
		max_level = a.max_level_rand_training ? (rng.next_float() * 2.0f) : 1.0f;
		bg[0] = a.background_color[0]; bg[1] = a.background_color[1]; bg[2] = a.background_color[2];
		if (a.train_with_random_bg_color) { bg[0] = rng.next_float(); bg[1] = rng.next_float(); bg[2] = rng.next_float(); }
		// The three colour channels go through the same conversions (srgb_to_linear / linear_to_srgb: a powf each, nine of them plus three expf per ray) and are
		// independent: lane c of the wave works out channel c (every lane computes; lanes 3 .. 63 repeat channel 2) and the three results are read back from lanes
		// 0 .. 2 — a third of the instructions of walking the channels one after the other, the same arithmetic per channel.
		const uint32_t ch = lane < 3u ? lane : 2u;
		auto pick = [&](float v0, float v1, float v2) { return ch == 0u ? v0 : (ch == 1u ? v1 : v2); };
		float bg_c = srgb_to_linear(pick(bg[0], bg[1], bg[2]));
		if (a.envmap_data) {   // composit the background behind the envmap (1394-1401)
			env_dir = normalized(ld3(a.rays_in[i].d));
			float e[4];
			read_envmap(a.envmap_data, a.envmap_res[0], a.envmap_res[1], env_dir, e);
			bg_c = pick(e[0], e[1], e[2]) + bg_c * (1.0f - e[3]);
		}
		const float exposure_c = expf(0.6931471805599453f * a.exposure[img * 3 + ch]);
		float texsamp[4];
		read_rgba(xy[0], xy[1], md.res, md.pixels, md.image_data_type, texsamp);
		const float tex_c = pick(texsamp[0], texsamp[1], texsamp[2]);
		float target_c;
		if (a.train_in_linear_colors || a.color_space == NGP_COLOR_LINEAR) {
			target_c = exposure_c * tex_c + (1.0f - texsamp[3]) * bg_c;
			if (!a.train_in_linear_colors) { target_c = linear_to_srgb(target_c); bg_c = linear_to_srgb(bg_c); }
		} else {
			bg_c = linear_to_srgb(bg_c);
			if (texsamp[3] > 0) target_c = linear_to_srgb(exposure_c * tex_c / texsamp[3]) * texsamp[3] + (1.0f - texsamp[3]) * bg_c;
			else target_c = bg_c;
		}
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			bg[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bg_c), c));
			rgbtarget[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, target_c), c));
			exposure_scale[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, exposure_c), c));
		}
	}

	// ---- pass 1: transmittance, ray colour, number of samples before T < EPSILON (1341-1374)
	v3 hitpoint = mk(0.f, 0.f, 0.f);
	{
		float T_carry = 1.f;
		bool done = false;
		for (uint32_t c0 = 0; c0 < numsteps && !done; c0 += 64) {
			const uint32_t j = c0 + lane;
			const bool valid = j < numsteps;
			float alpha = 0.f, rgb[3] = {0.f, 0.f, 0.f};
			if (valid) {
				const us4 lo = *(const us4*)(no + (size_t)j * a.mlp_stride);
				const float dt = unwarp_dt(ci[j].dt);
				alpha = 1.f - __expf(-network_to_density(h2f(lo[3]), a.density_activation) * dt);
				rgb[0] = network_to_rgb(h2f(lo[0]), a.rgb_activation);
				rgb[1] = network_to_rgb(h2f(lo[1]), a.rgb_activation);
				rgb[2] = network_to_rgb(h2f(lo[2]), a.rgb_activation);
			}
			const float incl = wave_inclusive<0>(1.f - alpha);
			const float excl = wave_prev(1.f, incl);
			const float T_before = T_carry * excl;
			const bool include = valid && !(T_before < EPSILON);
			const float weight = include ? alpha * T_before : 0.f;
			rgb_ray[0] += wave_sum(weight * rgb[0]);
			rgb_ray[1] += wave_sum(weight * rgb[1]);
			rgb_ray[2] += wave_sum(weight * rgb[2]);
			if (a.sharpness_data) {   // hitpoint += weight * pos (1367)
				v3 pos = mk(0.f, 0.f, 0.f);
				if (valid) { const NgpCoord cc = ci[j]; pos = unwarp_position(mk(cc.pos[0], cc.pos[1], cc.pos[2]), a.aabb); }
				hitpoint.x += wave_sum(weight * pos.x); hitpoint.y += wave_sum(weight * pos.y); hitpoint.z += wave_sum(weight * pos.z);
			}
			if (a.depth_supervision_lambda > 0.0f) {   // depth_ray += weight * cur_depth (1366-1368)
				float cur_depth = 0.f;
				if (valid) {
					const NgpCoord cc = ci[j];
					cur_depth = norm(unwarp_position(mk(cc.pos[0], cc.pos[1], cc.pos[2]), a.aabb) - ray_o_early);
				}
				depth_ray += wave_sum(weight * cur_depth);
			}
			const uint32_t n_inc = (uint32_t)__popcll(__ballot(include));
			const uint32_t n_valid = numsteps - c0 < 64 ? numsteps - c0 : 64;
			compacted += n_inc;
			if (n_inc < n_valid) done = true;
			T_carry = T_carry * wave_last(incl);
		}
		T_final = T_carry;
	}

	if (active && compacted == numsteps) {
#pragma unroll
		for (int c = 0; c < 3; ++c) rgb_ray[c] += T_final * bg[c];
	}

	// The gradient pass's first chunk reads what pass 1 read — at addresses that do not depend on the compaction slot — so its loads are issued HERE, in front of the
	// workgroup barriers and the slot atomic's round trip (a fifth of a wave's time, tools/loss_phase_probe.py), instead of behind them.
	NgpCoord cin0 = {};
	us4 lo0 = {0, 0, 0, 0};
	uint4 enc0[4] = {};
	if (lane < compacted) {
		cin0 = ci[lane];
		lo0 = *(const us4*)(no + (size_t)lane * a.mlp_stride);
		if (a.encoded_in) {
			const uint4* src = (const uint4*)(a.encoded_in + (size_t)(base + lane) * 32);
			enc0[0] = src[0]; enc0[1] = src[1]; enc0[2] = src[2]; enc0[3] = src[3];
		}
	}
	// ---- compaction slots: one atomic per workgroup (1434)
	if (lane == 0) s_counts[w] = compacted;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int k = 0; k < LOSS_RAYS_PER_BLOCK; ++k) { const uint32_t c = s_counts[k]; s_counts[k] = total; total += c; }
		s_block_base = total ? atomicAdd(a.numsteps_counter, total) : 0u;
	}
	__syncthreads();
	// (slots that produce no loss get a zero: the reference clears the whole array before the step, :2864)
	if (!active) { if (lane == 0 && a.loss_output && i < a.n_rays) a.loss_output[i] = 0.f; return; }
	const uint32_t compacted_base = s_block_base + s_counts[w];
	const uint32_t room = a.max_samples_compacted - (a.max_samples_compacted < compacted_base ? a.max_samples_compacted : compacted_base);
	compacted = room < compacted ? room : compacted;
	if (lane == 0) { a.numsteps_in[i * 2 + 0] = compacted; a.numsteps_in[i * 2 + 1] = compacted_base; }
	if (compacted == 0) { if (lane == 0 && a.loss_output) a.loss_output[i] = 0.f; return; }

	const LG lg = loss_and_gradient(rgbtarget, rgb_ray, a.loss_type);
	// per-image exposure gradient (1558-1572): d loss / d exposure = loss_scale * (-dL/drgb / xy_pdf [/ srgb'(target)]) * 2^exposure * ln 2
	if (a.exposure_gradient && lane == 0) {
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			float dloss_by_dgt = -lg.grad[c] / pixel_pdf;
			if (!a.train_in_linear_colors) dloss_by_dgt /= srgb_to_linear_derivative(rgbtarget[c]);
			atomicAdd(&a.exposure_gradient[img * 3 + c], a.loss_scale * dloss_by_dgt * exposure_scale[c] * 0.6931471805599453f);
		}
	}
	// environment-map gradient (1573-1596): only rays whose every sample was kept see the background; the value and the bilinear weight pass through
	// network_precision_t (fp16) like the reference's vector_t<half, 4> / `T weight`, the accumulation is fp32 (TrainableBuffer<4, 2, float>)
	if (a.envmap_gradient && compacted == numsteps && lane == 0) {
		LG lge = lg;
		if (a.envmap_loss_type != a.loss_type) lge = loss_and_gradient(rgbtarget, rgb_ray, a.envmap_loss_type);
		const EnvmapTap tap = envmap_taps(a.envmap_res[0], a.envmap_res[1], env_dir);
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			float d = T_final * lge.grad[c];
			if (!a.train_in_linear_colors) d /= srgb_to_linear_derivative(bg[c]);
			const half_t v16 = (half_t)(a.loss_scale * d);
#pragma unroll
			for (int k = 0; k < 4; ++k) atomicAdd(&a.envmap_gradient[tap.idx[k] + c], (float)(half_t)(v16 * (half_t)tap.w[k]));
		}
	}
	// depth supervision (1450-1452): target = |d| * depth image at the ray's pixel, loss on the expected termination depth
	float depth_loss_gradient = 0.0f;
	if (a.depth_supervision_lambda > 0.0f) {
		const NgpImageMeta& mdd = a.metadata[img];
		float dval = -1.0f;
		if (mdd.depth) {
			int px = (int)(xy[0] * (float)mdd.res[0]), py = (int)(xy[1] * (float)mdd.res[1]);
			px = px > 0 ? px : 0; px = px < mdd.res[0] - 1 ? px : mdd.res[0] - 1;
			py = py > 0 ? py : 0; py = py < mdd.res[1] - 1 ? py : mdd.res[1] - 1;
			dval = mdd.depth[(size_t)px + (size_t)py * (size_t)mdd.res[0]];
		}
		const float target_depth = norm(ld3(a.rays_in[i].d)) * dval;
		const float t3[3] = {target_depth, target_depth, target_depth}, p3[3] = {depth_ray, depth_ray, depth_ray};
		const LG lgd = loss_and_gradient(t3, p3, a.depth_loss_type);
		depth_loss_gradient = target_depth > 0.0f ? a.depth_supervision_lambda * lgd.grad[0] : 0.0f;
	}
	// lg.loss /= img_pdf * xy_pdf (1448): the reported loss and the error map are importance-weighted, the gradient deliberately is not (1454-1458)
	const float mean_loss = (lg.loss[0] / sample_pdf + lg.loss[1] / sample_pdf + lg.loss[2] / sample_pdf) / 3.0f;
	if (lane == 0) {
		if (a.loss_output) a.loss_output[i] = mean_loss / (float)a.n_rays;
		if (a.error_map) {
			float posx = xy[0] * (float)a.error_map_res[0] - 0.5f, posy = xy[1] * (float)a.error_map_res[1] - 0.5f;
			posx = fminf(fmaxf(posx, 0.0f), (float)a.error_map_res[0] - (1.0f + 1e-4f));
			posy = fminf(fmaxf(posy, 0.0f), (float)a.error_map_res[1] - (1.0f + 1e-4f));
			const int pix = (int)posx, piy = (int)posy;
			const float wx = posx - (float)pix, wy = posy - (float)piy;
			int ix = pix < img_res[0] - 2 ? pix : img_res[0] - 2; ix = ix > 0 ? ix : 0; // 1470 clamps with the IMAGE resolution
			int iy = piy < img_res[1] - 2 ? piy : img_res[1] - 2; iy = iy > 0 ? iy : 0;
			float* em = a.error_map + (size_t)img * (size_t)a.error_map_res[0] * (size_t)a.error_map_res[1];
			float em_loss = mean_loss;
			if (a.sharpness_data) {   // 1476-1485: scale the deposited error by the tile's sharpness relative to the sharpest tile that has seen the ray's hit cell
				const float inv = 1.0f / (1.0f - T_final);   // hitpoint /= 1 - T (1374)
				const v3 hp = mk(hitpoint.x * inv, hitpoint.y * inv, hitpoint.z * inv);
				if (aabb_contains(a.aabb, hp)) {
					int sx = (int)(xy[0] * (float)a.sharpness_res[0]), sy = (int)(xy[1] * (float)a.sharpness_res[1]);
					sx = sx > 0 ? sx : 0; sx = sx < a.sharpness_res[0] - 1 ? sx : a.sharpness_res[0] - 1;
					sy = sy > 0 ? sy : 0; sy = sy < a.sharpness_res[1] - 1 ? sy : a.sharpness_res[1] - 1;
					const float sharp = a.sharpness_data[(size_t)img * a.sharpness_res[0] * a.sharpness_res[1] + (size_t)sy * a.sharpness_res[0] + sx] + 1e-6f;
					const uint32_t mip = (uint32_t)mip_from_pos(hp);
					uint32_t* cell = (uint32_t*)&a.sharpness_grid[cascaded_grid_idx_at(hp, mip) + (size_t)NGP_NERF_GRID_N_CELLS * mip];
					// the maximum of positive floats in uint format is the maximum of the floats
					float grid_sharp = __uint_as_float(atomicMax(cell, __float_as_uint(sharp)));
					grid_sharp = fmaxf(sharp, grid_sharp);   // atomicMax returns the old value
					em_loss *= fmaxf(sharp / grid_sharp, 0.01f);
				}
			}
			atomicAdd(&em[(size_t)iy * a.error_map_res[0] + ix], (1 - wx) * (1 - wy) * em_loss);
			atomicAdd(&em[(size_t)iy * a.error_map_res[0] + ix + 1], wx * (1 - wy) * em_loss);
			atomicAdd(&em[(size_t)(iy + 1) * a.error_map_res[0] + ix], (1 - wx) * wy * em_loss);
			atomicAdd(&em[(size_t)(iy + 1) * a.error_map_res[0] + ix + 1], wx * wy * em_loss);
		}
	}

	// ---- pass 2: gradients + compacted copies (1498-1556), lanes = samples
	const float ls = a.loss_scale / (float)a.n_rays;
	const float output_l2_reg = a.rgb_activation == NGP_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
	const float output_l1_reg_density = *a.mean_density < MIN_OPTICAL_THICKNESS() ? 1e-4f : 0.0f;
	NgpCoord* __restrict__ co = a.coords_out + compacted_base;
	uint16_t* __restrict__ dl = a.dloss_doutput + (size_t)compacted_base * a.dl_stride;
	float T_carry = 1.f, acc_carry[3] = {0.f, 0.f, 0.f}, depth_carry = 0.f;
	for (uint32_t c0 = 0; c0 < compacted; c0 += 64) {
		const uint32_t j = c0 + lane;
		const bool valid = j < compacted;
		float alpha = 0.f, rgb[3] = {0.f, 0.f, 0.f}, lof[4] = {0.f, 0.f, 0.f, 0.f}, dt = 0.f, depth = 0.f;
		if (valid) {
			NgpCoord cin = cin0;
			us4 lo = lo0;
			uint4 r0 = enc0[0], r1 = enc0[1], r2 = enc0[2], r3 = enc0[3];
			if (c0 != 0) {   // (wave-uniform) later chunks: in place
				cin = ci[j];
				lo = *(const us4*)(no + (size_t)j * a.mlp_stride);
				if (a.encoded_in) {
					const uint4* src = (const uint4*)(a.encoded_in + (size_t)(base + j) * 32);
					r0 = src[0]; r1 = src[1]; r2 = src[2]; r3 = src[3];
				}
			}
			co[j] = cin;
			if (a.encoded_in) {
				uint4* dst = (uint4*)(a.encoded_out + (size_t)(compacted_base + j) * 32);
				dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
			}
			if (a.x_row_out) a.x_row_out[compacted_base + j] = base + j;
			if (a.max_level_rand_training) a.max_level_compacted[compacted_base + j] = max_level;
			const v3 pos = unwarp_position(mk(cin.pos[0], cin.pos[1], cin.pos[2]), a.aabb);
			depth = norm(pos - ray_o);
			dt = unwarp_dt(cin.dt);
			lof[0] = h2f(lo[0]); lof[1] = h2f(lo[1]); lof[2] = h2f(lo[2]); lof[3] = h2f(lo[3]);
#pragma unroll
			for (int c = 0; c < 3; ++c) rgb[c] = network_to_rgb(lof[c], a.rgb_activation);
			alpha = 1.f - __expf(-network_to_density(lof[3], a.density_activation) * dt);
		}
		const float incl = wave_inclusive<0>(1.f - alpha);
		const float excl = wave_prev(1.f, incl);
		const float T_before = T_carry * excl;
		const float weight = alpha * T_before;
		const float T = T_before * (1.f - alpha);  // transmittance after this sample (1522)
		float rgb_ray2[3];
#pragma unroll
		for (int c = 0; c < 3; ++c) rgb_ray2[c] = acc_carry[c] + wave_inclusive<1>(weight * rgb[c]);
		float depth_ray2 = 0.f;
		if (a.depth_supervision_lambda > 0.0f) depth_ray2 = depth_carry + wave_inclusive<1>(weight * depth);
		if (valid) {
			us4 out;
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const float dloss_by_drgb = weight * lg.grad[c];
				out[c] = f2h(ls * (dloss_by_drgb * network_to_rgb_derivative(lof[c], a.rgb_activation) + fmaxf(0.0f, output_l2_reg * lof[c])));
			}
			const float density_derivative = network_to_density_derivative(lof[3], a.density_activation);
			const float dotv = lg.grad[0] * (T * rgb[0] - (rgb_ray[0] - rgb_ray2[0])) + lg.grad[1] * (T * rgb[1] - (rgb_ray[1] - rgb_ray2[1])) + lg.grad[2] * (T * rgb[2] - (rgb_ray[2] - rgb_ray2[2]));
			const float depth_supervision = depth_loss_gradient * (T * depth - (depth_ray - depth_ray2));   // 1536-1537; 0 when switched off
			const float dloss_by_dmlp = density_derivative * (dt * (dotv + depth_supervision));
			out[3] = f2h(ls * dloss_by_dmlp + (lof[3] < 0.0f ? -output_l1_reg_density : 0.0f) + (lof[3] > -10.0f && depth < a.near_distance ? 1e-4f : 0.0f));
			*(us4*)(dl + (size_t)j * a.dl_stride) = out;
		}
		T_carry = T_carry * wave_last(incl);
#pragma unroll
		for (int c = 0; c < 3; ++c) acc_carry[c] = wave_last(rgb_ray2[c]);
		depth_carry = wave_last(depth_ray2);
	}
}

__global__ void fill_rollover_and_rescale_f16_kernel(uint32_t n_elements, uint32_t stride, const uint32_t* __restrict__ n_input_elements_ptr, uint16_t* __restrict__ inout) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t n_total = n_elements * stride;
	const uint32_t n_in = *n_input_elements_ptr;
	if (n_in >= n_elements || n_in == 0) return;
	const uint32_t n_input = n_in * stride;
	if (i < n_input || i >= n_total) return;
	inout[i] = f2h(h2f(inout[i % n_input]) * (float)n_input / (float)n_total);
}
__global__ void fill_rollover_f32_kernel(uint32_t n_elements, uint32_t stride, const uint32_t* __restrict__ n_input_elements_ptr, float* __restrict__ inout) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t n_total = n_elements * stride;
	const uint32_t n_in = *n_input_elements_ptr;
	if (n_in >= n_elements || n_in == 0) return;
	const uint32_t n_input = n_in * stride;
	if (i < n_input || i >= n_total) return;
	inout[i] = inout[i % n_input];
}

// the three roll-overs of a training step (3314-3322 + the carried encoding rows) in one launch: element i of each array, same arithmetic
__global__ void fill_rollover_training_kernel(uint32_t n_elements, const uint32_t* __restrict__ n_input_elements_ptr, uint16_t* __restrict__ dloss, uint32_t dl_stride,
                                              float* __restrict__ coords, uint32_t coord_stride, float* __restrict__ encoded, uint32_t enc_stride) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t n_in = *n_input_elements_ptr;
	if (n_in >= n_elements || n_in == 0) return;
	{ const uint32_t n_total = n_elements * dl_stride, n_input = n_in * dl_stride;
	  if (i >= n_input && i < n_total) dloss[i] = f2h(h2f(dloss[i % n_input]) * (float)n_input / (float)n_total); }
	{ const uint32_t n_total = n_elements * coord_stride, n_input = n_in * coord_stride;
	  if (i >= n_input && i < n_total) coords[i] = coords[i % n_input]; }
	if (encoded) { const uint32_t n_total = n_elements * enc_stride, n_input = n_in * enc_stride;
	  if (i >= n_input && i < n_total) encoded[i] = encoded[i % n_input]; }
}

// The step's counter post and its three roll-overs in ONE launch of 256 workgroups: thread 0 first hands {*a, *b, *c, tag} to the (host-mapped) dst4 like
// post_words_kernel (density_grid.hip) — the host is polling for it — then all threads fill the wrapped-around tails, enumerating only the positions that need a fill
// (fill_rollover_training_kernel launches a thread per element of the widest array, 16 k workgroups that mostly have nothing to do).
__global__ void __launch_bounds__(256) post_and_rollover_training_kernel(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst4, uint32_t* zero_words, uint32_t n_zero_words,
                                                                         double* sum3_dev, uint32_t n_elements, const uint32_t* __restrict__ n_input_elements_ptr, uint16_t* __restrict__ dloss,
                                                                         uint32_t dl_stride, float* __restrict__ coords, uint32_t coord_stride, float* __restrict__ encoded, uint32_t enc_stride) {
	const uint32_t n_in = *n_input_elements_ptr;   // (read before the post clears nothing of it: the zeroed words belong to the NEXT step's slot)
	if (blockIdx.x == 0 && threadIdx.x == 0 && dst4) {
		const uint32_t va = a ? *a : 0u, vb = b ? *b : 0u, vc = c ? *c : 0u;
		dst4[0] = va; dst4[1] = vb; dst4[2] = vc;
		if (sum3_dev) { sum3_dev[0] = (double)va; sum3_dev[1] = (double)vb; sum3_dev[2] = (double)__uint_as_float(vc); }
		for (uint32_t k = 0; k < n_zero_words; ++k) zero_words[k] = 0u;
		__threadfence_system();
		__hip_atomic_store(&dst4[3], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (n_in >= n_elements || n_in == 0) return;
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
	{ const uint32_t n_total = n_elements * dl_stride, n_input = n_in * dl_stride;
	  for (uint32_t i = n_input + t; i < n_total; i += nt) dloss[i] = f2h(h2f(dloss[i % n_input]) * (float)n_input / (float)n_total); }
	{ const uint32_t n_total = n_elements * coord_stride, n_input = n_in * coord_stride;
	  for (uint32_t i = n_input + t; i < n_total; i += nt) coords[i] = coords[i % n_input]; }
	if (encoded) { const uint32_t n_total = n_elements * enc_stride, n_input = n_in * enc_stride;
	  for (uint32_t i = n_input + t; i < n_total; i += nt) encoded[i] = encoded[i % n_input]; }
}

// ---- plumbing configs P1 / P2: tcnn losses driven by Trainer::training_step, sample generation of Testbed::train_image ----------------
// [tcnn] losses/{l2,relative_l2,l1,mape}.h: per element i of the (padded) prediction matrix, n_total = n * dims,
//   value = f(d) / n_total, gradient = loss_scale * f'(d) / n_total with d = prediction - target.  Padding channels get zero gradient.
__global__ void loss_and_gradient_kernel(int loss_type, uint32_t n, uint32_t dims, float loss_scale, const half_t* __restrict__ predictions, uint32_t pred_stride,
                                         const float* __restrict__ targets, float* __restrict__ values, half_t* __restrict__ gradients, uint32_t grad_stride) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n * 4u) return;
	const uint32_t s = i >> 2, c = i & 3u;
	if (c >= dims) { gradients[(size_t)s * grad_stride + c] = (half_t)0.0f; return; }
	const float n_total = (float)(n * dims);
	const float prediction = (float)predictions[(size_t)s * pred_stride + c];
	const float target = targets[(size_t)s * dims + c];
	const float difference = prediction - target;
	float value, gradient;
	switch (loss_type) {
		case NGP_LOSS_L2: value = difference * difference; gradient = 2.0f * difference; break;
		case NGP_LOSS_RELATIVE_L2: { const float pn = prediction * prediction + 0.01f; value = difference * difference / pn; gradient = 2.0f * difference / pn; break; }
		case NGP_LOSS_L1: value = fabsf(difference); gradient = copysignf(1.0f, difference); break;
		default: { const float sc = 1.0f / (fabsf(target) + 0.01f); value = fabsf(difference) * sc; gradient = copysignf(sc, difference); break; }   // MAPE
	}
	values[(size_t)s * dims + c] = value / n_total;
	float gs = loss_scale * gradient / n_total;
	asm volatile("" : "+v"(gs));   // one rounding to fp16 of the fp32 product (no fused multiply-convert)
	gradients[(size_t)s * grad_stride + c] = (half_t)gs;
}

// [tcnn] generate_random_uniform: element k is the k-th next_float() of the generator (each thread advances to its own offset)
__global__ void random_uniform_kernel(uint32_t n_elements, Pcg32 rng, float* __restrict__ out) {
	const uint32_t i = (threadIdx.x + blockIdx.x * blockDim.x) * 4u;
	if (i >= n_elements) return;
	rng.advance(i);
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) { if (i + j < n_elements) out[i + j] = rng.next_float(); }
}

// testbed_image.cu:62-77 stratify2_kernel
__global__ void stratify2_kernel(uint32_t n_elements, uint32_t log2_batch_size, float2* __restrict__ inout) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	const uint32_t log2_size = log2_batch_size / 2, size = 1u << log2_size;
	const uint32_t in_batch_index = i & ((1u << log2_batch_size) - 1u);
	const uint32_t x = in_batch_index & ((1u << log2_size) - 1u), y = in_batch_index >> log2_size;
	const float2 val = inout[i];
	inout[i] = make_float2(val.x / (float)size + ((float)x / (float)size), val.y / (float)size + ((float)y / (float)size));
}

// testbed_image.cu:172-218 eval_image_kernel_and_snap<T, stride>: bilinear target lookup (or nearest + snap of the position)
template <typename T>
__global__ void eval_image_and_snap_kernel(uint32_t n_elements, const T* __restrict__ texture, float2* __restrict__ positions, int rx, int ry, float* __restrict__ result, uint32_t stride,
                                           bool snap_to_pixel_centers, bool linear_colors) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	float2 pos = positions[i];
	auto read_val = [&](int x, int y, float* o) {
		const T* t = texture + ((size_t)y * rx + x) * 4;
		o[0] = (float)t[0]; o[1] = (float)t[1]; o[2] = (float)t[2]; o[3] = (float)t[3];
		if (!linear_colors) { o[0] = linear_to_srgb(o[0]); o[1] = linear_to_srgb(o[1]); o[2] = linear_to_srgb(o[2]); }
	};
	float val[4];
	if (snap_to_pixel_centers) {
		int px = (int)floorf(pos.x * (float)rx), py = (int)floorf(pos.y * (float)ry);
		positions[i] = make_float2(((float)px + 0.5f) / (float)rx, ((float)py + 0.5f) / (float)ry);
		px = px < 0 ? 0 : (px > rx - 1 ? rx - 1 : px); py = py < 0 ? 0 : (py > ry - 1 ? ry - 1 : py);
		read_val(px, py, val);
	} else {
		pos.x = fminf(fmaxf(pos.x * (float)rx - 0.5f, 0.0f), (float)rx - (1.0f + 1e-4f));
		pos.y = fminf(fmaxf(pos.y * (float)ry - 0.5f, 0.0f), (float)ry - (1.0f + 1e-4f));
		const int ix = (int)pos.x, iy = (int)pos.y;
		const float wx = pos.x - (float)ix, wy = pos.y - (float)iy;
		const int x0 = ix < rx - 2 ? (ix < 0 ? 0 : ix) : (rx - 2 < 0 ? 0 : rx - 2), y0 = iy < ry - 2 ? (iy < 0 ? 0 : iy) : (ry - 2 < 0 ? 0 : ry - 2);
		float a[4], b[4], c[4], d[4];
		read_val(x0, y0, a); read_val(x0 + 1, y0, b); read_val(x0, y0 + 1, c); read_val(x0 + 1, y0 + 1, d);
#pragma unroll
		for (int k = 0; k < 4; ++k) val[k] = ((((1 - wx) * (1 - wy)) * a[k] + ((wx) * (1 - wy)) * b[k]) + ((1 - wx) * (wy)) * c[k]) + ((wx) * (wy)) * d[k];
	}
	float* r = result + (size_t)i * stride;
	r[0] = val[0]; r[1] = val[1]; r[2] = val[2];
	for (uint32_t k = 3; k < stride; ++k) r[k] = 1.0f;
}

// testbed_image.cu:79-108 init_image_coords (pixel_to_image_uv, common_device.cuh:397-417)
__global__ void init_image_coords_kernel(float2* __restrict__ positions, int rx, int ry, int irx, int iry, float view_dist, float ipx, float ipy, float scx, float scy, bool snap, uint32_t sample_index) {
	const uint32_t x = threadIdx.x + blockDim.x * blockIdx.x, y = threadIdx.y + blockDim.y * blockIdx.y;
	if (x >= (uint32_t)rx || y >= (uint32_t)ry) return;
	float jx, jy;
	ld_random_pixel_offset(snap ? 0 : sample_index, jx, jy);
	const float ox = scx * (float)rx + jx, oy = scy * (float)ry + jy;
	const float y_scale = view_dist, x_scale = y_scale * (float)rx / (float)ry;
	positions[x + (size_t)rx * y] = make_float2(((x_scale * ((float)x + ox)) / (float)rx - view_dist * ipx) / (float)irx * (float)iry, (y_scale * ((float)y + oy)) / (float)ry - view_dist * ipy);
}

// testbed_image.cu:139-170 shade_kernel_image; colours come as the network's fp16 outputs (channels 0..2 of `color_stride` halves per pixel)
__global__ void shade_image_kernel(int rx, int ry, const float2* __restrict__ positions, const half_t* __restrict__ colors, uint32_t color_stride, float4* __restrict__ frame_buffer,
                                   float* __restrict__ depth_buffer, bool linear_colors) {
	const uint32_t x = threadIdx.x + blockDim.x * blockIdx.x, y = threadIdx.y + blockDim.y * blockIdx.y;
	if (x >= (uint32_t)rx || y >= (uint32_t)ry) return;
	const size_t idx = x + (size_t)rx * y;
	const float2 uv = positions[idx];
	if (uv.x < 0.0f || uv.x > 1.0f || uv.y < 0.0f || uv.y > 1.0f) { frame_buffer[idx] = make_float4(0.f, 0.f, 0.f, 0.f); depth_buffer[idx] = 1e10f; return; }
	float r = (float)colors[idx * color_stride + 0], g = (float)colors[idx * color_stride + 1], b = (float)colors[idx * color_stride + 2];
	if (!linear_colors) { r = srgb_to_linear(r); g = srgb_to_linear(g); b = srgb_to_linear(b); }
	frame_buffer[idx] = make_float4(r, g, b, 1.0f);
	depth_buffer[idx] = 1.0f;
}

// testbed_image.cu:436-446
__global__ void image_coords_from_idx_kernel(uint32_t n_elements, uint32_t offset, float2* __restrict__ pos, int rx, int ry) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	const uint32_t idx = i + offset;
	int x = (int)(idx % (uint32_t)rx), y = (int)(idx / (uint32_t)rx);
	x = x < 0 ? 0 : (x > rx - 1 ? rx - 1 : x); y = y < 0 ? 0 : (y > ry - 1 ? ry - 1 : y);
	pos[i] = make_float2(((float)x + 0.5f) / (float)rx, ((float)y + 0.5f) / (float)ry);
}

// testbed_image.cu:448-460 image_mse_kernel; prediction = fp16 network outputs
__global__ void image_mse_kernel(uint32_t n_elements, const float* __restrict__ target, const half_t* __restrict__ prediction, uint32_t pred_stride, float* __restrict__ result, bool quantize_to_byte) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	float acc = 0.0f;
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		float p = (float)prediction[(size_t)i * pred_stride + c];
		if (quantize_to_byte) { int q = (int)(p * 255.0f + 0.5f); q = q < 0 ? 0 : (q > 255 ? 255 : q); p = (float)q / 255.0f; }
		const float d = target[(size_t)i * 3 + c] - p;
		acc += d * d;
	}
	result[i] = acc / 3.0f;
}

// compute_sharpness (src/nerf_loader.cu:129-169)
__global__ void compute_sharpness_kernel(int srx, int sry, int irx, int iry, const void* __restrict__ pixels, int image_data_type, float* __restrict__ sharpness_data) {
	const int x = threadIdx.x + blockIdx.x * blockDim.x, y = threadIdx.y + blockIdx.y * blockDim.y;
	if (x >= srx || y >= sry) return;
	int x1 = (x * irx) / srx, x2 = ((x + 1) * irx) / srx, y1 = (y * iry) / sry, y2 = ((y + 1) * iry) / sry;
	x1 = max(x1, 1); y1 = max(y1, 1); x2 = min(x2, irx - 2); y2 = min(y2, iry - 2);   // clamp to 1 pixel in from the edge
	const int32_t res[2] = {irx, iry};
	float tot_lap = 0.f, tot_lap2 = 0.f;
	const float scal = 1.f / (float)((x2 - x1) * (y2 - y1));
	auto luma_at = [&](int px, int py) {
		float c[4];
		read_rgba(((float)px + 0.5f) / (float)irx, ((float)py + 0.5f) / (float)iry, res, pixels, image_data_type, c);
		return c[0] * 0.2126f + c[1] * 0.7152f + c[2] * 0.0722f;
	};
	for (int yy = y1; yy < y2; ++yy) for (int xx = x1; xx < x2; ++xx) {
		const float lum = luma_at(xx, yy);
		const float lap = lum * 4.f - luma_at(xx, yy - 1) - luma_at(xx + 1, yy) - luma_at(xx, yy + 1) - luma_at(xx - 1, yy);
		tot_lap += lap; tot_lap2 += lap * lap;
	}
	tot_lap *= scal; tot_lap2 *= scal;
	sharpness_data[x + (size_t)y * srx] = tot_lap2 - tot_lap * tot_lap;
}
__global__ void decay_grid_kernel(uint32_t n, float decay, float* __restrict__ grid) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) grid[i] *= decay;
}

} // namespace ngp

// ---- error-map CDFs (testbed_nerf.cu:1982-2037).  One thread per (image, row) / per image like the reference: the running sums are
// sequential fp32 additions whose order is part of the result.
namespace ngp {
constexpr float MIN_PDF = 0.01f;
__global__ void construct_cdf_2d_kernel(uint32_t n_images, uint32_t height, uint32_t width, const float* __restrict__ data, float* __restrict__ cdf_x_cond_y, float* __restrict__ cdf_y) {
	const uint32_t y = threadIdx.x + blockIdx.x * blockDim.x;
	const uint32_t img = threadIdx.y + blockIdx.y * blockDim.y;
	if (y >= height || img >= n_images) return;
	const size_t off = ((size_t)img * height + y) * width;
	data += off; cdf_x_cond_y += off;
	float cum = 0;
	for (uint32_t x = 0; x < width; ++x) { cum += data[x] + 1e-10f; cdf_x_cond_y[x] = cum; }
	cdf_y[(size_t)img * height + y] = cum;
	const float norm = 1.0f / cum;   // __frcp_rn
	for (uint32_t x = 0; x < width; ++x) cdf_x_cond_y[x] = (1.0f - MIN_PDF) * cdf_x_cond_y[x] * norm + MIN_PDF * (float)(x + 1) / (float)width;
}
__global__ void construct_cdf_1d_kernel(uint32_t n_images, uint32_t height, float* __restrict__ cdf_y, float* __restrict__ cdf_img) {
	const uint32_t img = threadIdx.x + blockIdx.x * blockDim.x;
	if (img >= n_images) return;
	cdf_y += (size_t)img * height;
	float cum = 0;
	for (uint32_t y = 0; y < height; ++y) { cum += cdf_y[y]; cdf_y[y] = cum; }
	cdf_img[img] = cum;
	const float norm = 1.0f / cum;
	for (uint32_t y = 0; y < height; ++y) cdf_y[y] = (1.0f - MIN_PDF) * cdf_y[y] * norm + MIN_PDF * (float)(y + 1) / (float)height;
}
}  // namespace ngp

// ---- load-time image sharpening (nerf_loader.cu:102-123, 803-825) and the Byte -> half conversion that precedes it (common_device.cuh:562-590)
namespace ngp {
__global__ void from_rgba32_half_kernel(uint64_t num_pixels, const uint8_t* __restrict__ pixels, half_t* __restrict__ out, uint32_t mask_color) {
	const uint64_t i = threadIdx.x + (uint64_t)blockIdx.x * blockDim.x;
	if (i >= num_pixels) return;
	const uint32_t v = ((const uint32_t*)pixels)[i];
	const float alpha = (float)(v >> 24) * (1.0f / 255.0f);
	half_t o[4];
	o[0] = (half_t)(srgb_to_linear((float)(v & 0xffu) * (1.0f / 255.0f)) * alpha);
	o[1] = (half_t)(srgb_to_linear((float)((v >> 8) & 0xffu) * (1.0f / 255.0f)) * alpha);
	o[2] = (half_t)(srgb_to_linear((float)((v >> 16) & 0xffu) * (1.0f / 255.0f)) * alpha);
	o[3] = (half_t)alpha;
	if (mask_color != 0 && mask_color == v) { o[0] = o[1] = o[2] = o[3] = (half_t)-1.0f; }
#pragma unroll
	for (int j = 0; j < 4; ++j) out[i * 4 + j] = o[j];
}
template <typename T>
__global__ void sharpen_kernel(uint64_t num_pixels, uint32_t w, const T* __restrict__ pix, T* __restrict__ destpix, float center_w, float inv_totalw) {
	const uint64_t i = threadIdx.x + (uint64_t)blockIdx.x * blockDim.x;
	if (i >= num_pixels) return;
	float rgba[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) rgba[j] = (float)pix[i * 4 + j] * center_w;
	int64_t i2 = (int64_t)i - 1; if (i2 < 0) i2 = 0; i2 *= 4;
	for (int j = 0; j < 4; ++j) rgba[j] -= (float)pix[i2++];
	i2 = (int64_t)i - w; if (i2 < 0) i2 = 0; i2 *= 4;
	for (int j = 0; j < 4; ++j) rgba[j] -= (float)pix[i2++];
	i2 = (int64_t)i + 1; if (i2 >= (int64_t)num_pixels) i2 -= (int64_t)num_pixels; i2 *= 4;
	for (int j = 0; j < 4; ++j) rgba[j] -= (float)pix[i2++];
	i2 = (int64_t)i + w; if (i2 >= (int64_t)num_pixels) i2 -= (int64_t)num_pixels; i2 *= 4;
	for (int j = 0; j < 4; ++j) rgba[j] -= (float)pix[i2++];
	for (int j = 0; j < 4; ++j) destpix[i * 4 + j] = (T)fmaxf(0.f, rgba[j] * inv_totalw);
}
// safe_divide (2039-2045): the distortion gradient image over its weight image
__global__ void safe_divide_kernel(uint32_t n, float* __restrict__ inout, const float* __restrict__ divisor) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n) return;
	const float d = divisor[i];
	inout[i] = d > 0.0f ? (inout[i] / d) : 0.0f;
}
// [tcnn] Adam (+ Ema) on a TrainableBuffer<N, 2, float>: every parameter is a non-matrix parameter (layer_sizes() is empty), so an entry whose
// gradient is exactly zero is skipped and l2_reg never applies; weights, gradients and the Ema copy are all fp32.
__global__ void __launch_bounds__(256) adam_ema_f32_kernel(uint32_t n, float lr, float beta1, float beta2, float epsilon, float loss_scale, float ema_decay, float ema_debias_old,
                                                           float ema_debias_new, const float* __restrict__ grads, float* __restrict__ params, float* __restrict__ m1,
                                                           float* __restrict__ m2, float* __restrict__ ema) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float g = grads[i] / loss_scale;
	float w = params[i];
	if (g != 0.0f) {
		const float fm = beta1 * m1[i] + (1.0f - beta1) * g;
		const float sm = beta2 * m2[i] + (1.0f - beta2) * (g * g);
		m1[i] = fm; m2[i] = sm;
		w = w - (lr / (sqrtf(sm) + epsilon)) * fm;
		params[i] = w;
	}
	if (ema) ema[i] = (ema[i] * ema_decay * ema_debias_old + w * (1.0f - ema_decay)) * ema_debias_new;
}
}  // namespace ngp

using namespace ngp;

// ---------------------------------------------------------------- camera gradient (1600-1712, extrinsics part)
// The reference walks a ray's compacted samples in ONE THREAD.  Here 16 lanes share a ray (the compacted batch averages ~10-20 samples a ray):
// lane l sums samples l, l+16, ... and the 16 partial sums are folded with xor-shuffles, so the per-ray sums are associated differently
// from the sequential loop (the atomic accumulation over rays is unordered in the reference as well).
constexpr int CAMGRAD_LANES = 16;
__global__ void __launch_bounds__(256) compute_cam_gradient_kernel(
	uint32_t n_rays, Aabb aabb, Pcg32 rng_in, const uint32_t* __restrict__ rays_counter, int snap_to_pixel_centers, float* __restrict__ cam_pos_gradient,
	float* __restrict__ cam_rot_gradient, uint32_t n_training_images, const NgpImageMeta* __restrict__ metadata, const uint32_t* __restrict__ ray_indices_in,
	const NgpRay* __restrict__ rays_in, const uint32_t* __restrict__ numsteps_in, const NgpCoord* __restrict__ coords, const float* __restrict__ coords_gradient, ErrorMapCdf cdf,
	const NgpXForm* __restrict__ xforms, float* __restrict__ distortion_gradient, float* __restrict__ distortion_gradient_weight, int dist_rx, int dist_ry) {
	const uint32_t sub = threadIdx.x % CAMGRAD_LANES;
	const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / CAMGRAD_LANES;
	const bool active = i < *rays_counter;
	const uint32_t numsteps = active ? numsteps_in[i * 2 + 0] : 0;          // 0: "the ray doesn't matter" (1633-1636)
	const uint32_t base = active ? numsteps_in[i * 2 + 1] : 0;
	const v3 ray_o = active ? ld3(rays_in[i].o) : mk(0.f, 0.f, 0.f);
	const v3 inv_diag = mk(1.0f / (aabb.mx.x - aabb.mn.x), 1.0f / (aabb.mx.y - aabb.mn.y), 1.0f / (aabb.mx.z - aabb.mn.z));   // warp_position_derivative (288-290)
	v3 go = mk(0.f, 0.f, 0.f), gd = mk(0.f, 0.f, 0.f);
	for (uint32_t j = sub; j < numsteps; j += CAMGRAD_LANES) {
		const NgpCoord& c = coords[base + j];
		const float* g = coords_gradient + (size_t)(base + j) * 6;
		const v3 pg = mk(g[0] * inv_diag.x, g[1] * inv_diag.y, g[2] * inv_diag.z);
		go = go + pg;
		const v3 pos = unwarp_position(mk(c.pos[0], c.pos[1], c.pos[2]), aabb);
		const float t = norm(pos - ray_o);
		gd = gd + (pg * t + mk(g[3] * 0.5f, g[4] * 0.5f, g[5] * 0.5f));    // warp_direction_derivative = 0.5 (300-302)
	}
#pragma unroll
	for (int off = CAMGRAD_LANES / 2; off > 0; off >>= 1) {
		go.x += __shfl_xor(go.x, off, 64); go.y += __shfl_xor(go.y, off, 64); go.z += __shfl_xor(go.z, off, 64);
		gd.x += __shfl_xor(gd.x, off, 64); gd.y += __shfl_xor(gd.y, off, 64); gd.z += __shfl_xor(gd.z, off, 64);
	}
	if (sub != 0 || numsteps == 0) return;
	const uint32_t ray_idx = ray_indices_in[i];
	const uint32_t img = image_idx(ray_idx, n_rays, n_training_images, cdf.cdf_img, nullptr);
	Pcg32 rng = rng_in;
	rng.advance((uint64_t)(uint32_t)(ray_idx * NGP_N_MAX_RANDOM_SAMPLES_PER_RAY));
	float u, v, xy_pdf = 1.0f;
	nerf_random_image_pos_training(rng, metadata[img].res, snap_to_pixel_centers, cdf, img, u, v, &xy_pdf);
	if (distortion_gradient) {   // 1673-1685: the ray-direction gradient, projected onto the plane normal to the ray, rotated into the camera frame, splatted at the pixel
		const v3 d = normalized(ld3(rays_in[i].d));
		const float gdd = dot(gd, d);
		const v3 og = gd - d * gdd;
		float inv[9];
		mat3_inverse(xforms[img].start, inv);
		const v3 ipg = mat3_mul(inv, og);
		deposit_image_gradient2(ipg.x / xy_pdf, ipg.y / xy_pdf, distortion_gradient, distortion_gradient_weight, dist_rx, dist_ry, u, v);
	}
	if (cam_pos_gradient) {
		atomicAdd(&cam_pos_gradient[img * 3 + 0], go.x / xy_pdf);
		atomicAdd(&cam_pos_gradient[img * 3 + 1], go.y / xy_pdf);
		atomicAdd(&cam_pos_gradient[img * 3 + 2], go.z / xy_pdf);
	}
	if (cam_rot_gradient) {
		// rotation is averaged in log space: angle-axis = ray.d x ray_gradient.d (1692-1707)
		const v3 d = normalized(ld3(rays_in[i].d));
		atomicAdd(&cam_rot_gradient[img * 3 + 0], (d.y * gd.z - d.z * gd.y) / xy_pdf);
		atomicAdd(&cam_rot_gradient[img * 3 + 1], (d.z * gd.x - d.x * gd.z) / xy_pdf);
		atomicAdd(&cam_rot_gradient[img * 3 + 2], (d.x * gd.y - d.y * gd.x) / xy_pdf);
	}
}

extern "C" {

int ngp_hip_image_from_rgba32_f16(void* stream, uint64_t n_pixels, const uint8_t* rgba8, uint16_t* out_half4, uint32_t mask_color) {
	if (!n_pixels) return 0;
	hipLaunchKernelGGL(from_rgba32_half_kernel, dim3((uint32_t)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_pixels, rgba8, (half_t*)out_half4, mask_color);
	NGP_LAUNCH_CHECK("from_rgba32_half_kernel");
	return 0;
}
int ngp_hip_image_sharpen(void* stream, uint64_t n_pixels, uint32_t width, const void* pix, void* dest, int image_data_type, float sharpen_amount) {
	if (!n_pixels) return 0;
	if (!(sharpen_amount > 0.f) || (image_data_type != 2 && image_data_type != 3) || pix == dest) { set_last_error("ngp_hip_image_sharpen: amount > 0, half4 (2) or float4 (3) pixels, out of place", hipErrorInvalidValue); return -1; }
	const float center_w = 4.f + 1.f / sharpen_amount;   // 5 (strong) ... infinite (none)
	const float inv_totalw = 1.f / (center_w - 4.f);
	const dim3 grid((uint32_t)((n_pixels + 255) / 256));
	if (image_data_type == 2) hipLaunchKernelGGL(sharpen_kernel<half_t>, grid, dim3(256), 0, (hipStream_t)stream, n_pixels, width, (const half_t*)pix, (half_t*)dest, center_w, inv_totalw);
	else hipLaunchKernelGGL(sharpen_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, n_pixels, width, (const float*)pix, (float*)dest, center_w, inv_totalw);
	NGP_LAUNCH_CHECK("sharpen_kernel");
	return 0;
}

int ngp_hip_construct_cdf_2d(void* stream, uint32_t n_images, uint32_t height, uint32_t width, const float* data, float* cdf_x_cond_y, float* cdf_y) {
	if (!n_images || !height || !width) return 0;
	hipLaunchKernelGGL(construct_cdf_2d_kernel, dim3(div_up(height, 16), div_up(n_images, 8)), dim3(16, 8), 0, (hipStream_t)stream, n_images, height, width, data, cdf_x_cond_y, cdf_y);
	NGP_LAUNCH_CHECK("construct_cdf_2d_kernel");
	return 0;
}
int ngp_hip_construct_cdf_1d(void* stream, uint32_t n_images, uint32_t height, float* cdf_y, float* cdf_img) {
	if (!n_images || !height) return 0;
	hipLaunchKernelGGL(construct_cdf_1d_kernel, dim3(div_up(n_images, 128)), dim3(128), 0, (hipStream_t)stream, n_images, height, cdf_y, cdf_img);
	NGP_LAUNCH_CHECK("construct_cdf_1d_kernel");
	return 0;
}

int ngp_hip_compute_loss(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint64_t rng_state, uint64_t rng_inc, uint32_t max_samples_compacted,
	const uint32_t* rays_counter, float loss_scale, uint32_t mlp_stride, const float* background_color_host, int color_space,
	int train_with_random_bg_color, int train_in_linear_colors, uint32_t n_training_images, const NgpImageMeta* metadata,
	const uint16_t* network_output, uint32_t* numsteps_counter, const uint32_t* ray_indices_in, const NgpRay* rays_in_unnormalized,
	uint32_t* numsteps_in, const NgpCoord* coords_in, NgpCoord* coords_out, uint16_t* dloss_doutput, uint32_t dl_stride, int loss_type,
	float* loss_output, int max_level_rand_training, float* max_level_compacted, int rgb_activation, int density_activation,
	int snap_to_pixel_centers, float* error_map, const int32_t* error_map_res_host, const float* mean_density, const float* exposure,
	float near_distance, const NgpErrorMapCdf* cdf_host, const uint16_t* encoded_in, uint16_t* encoded_out, float depth_supervision_lambda, int depth_loss_type,
	float* exposure_gradient, const NgpLossExtras* extras_host) {
	if (!n_rays) return 0;
	if ((encoded_in == nullptr) != (encoded_out == nullptr)) { set_last_error("ngp_hip_compute_loss: encoded_in and encoded_out go together", hipErrorInvalidValue); return -1; }
	if ((mlp_stride & 3) || (dl_stride & 3)) { set_last_error("ngp_hip_compute_loss: strides must be multiples of 4 halves", hipErrorInvalidValue); return -1; }
	LossArgs a;
	a.n_rays = n_rays; a.aabb = aabb_from_host(aabb_host); a.rng.state = rng_state; a.rng.inc = rng_inc; a.max_samples_compacted = max_samples_compacted;
	a.rays_counter = rays_counter; a.loss_scale = loss_scale; a.mlp_stride = mlp_stride;
	for (int c = 0; c < 3; ++c) a.background_color[c] = background_color_host[c];
	a.color_space = color_space; a.train_with_random_bg_color = train_with_random_bg_color; a.train_in_linear_colors = train_in_linear_colors;
	a.n_training_images = n_training_images; a.metadata = metadata; a.network_output = network_output; a.numsteps_counter = numsteps_counter;
	a.ray_indices_in = ray_indices_in; a.rays_in = rays_in_unnormalized; a.numsteps_in = numsteps_in; a.coords_in = coords_in; a.coords_out = coords_out;
	a.dloss_doutput = dloss_doutput; a.dl_stride = dl_stride; a.loss_type = loss_type; a.loss_output = loss_output;
	a.max_level_rand_training = max_level_rand_training; a.max_level_compacted = max_level_compacted; a.rgb_activation = rgb_activation;
	a.cdf = make_error_map_cdf(cdf_host); a.encoded_in = encoded_in; a.encoded_out = encoded_out;
	a.depth_supervision_lambda = depth_supervision_lambda; a.depth_loss_type = depth_loss_type; a.exposure_gradient = exposure_gradient;
	a.density_activation = density_activation; a.snap_to_pixel_centers = snap_to_pixel_centers; a.error_map = error_map;
	a.error_map_res[0] = error_map_res_host ? error_map_res_host[0] : 0; a.error_map_res[1] = error_map_res_host ? error_map_res_host[1] : 0;
	a.mean_density = mean_density; a.exposure = exposure; a.near_distance = near_distance;
	a.envmap_data = nullptr; a.envmap_gradient = nullptr; a.envmap_res[0] = a.envmap_res[1] = 0; a.envmap_loss_type = loss_type;
	if (extras_host && extras_host->envmap_data && extras_host->envmap_res[0] > 0 && extras_host->envmap_res[1] > 0) {
		a.envmap_data = extras_host->envmap_data; a.envmap_gradient = extras_host->envmap_gradient;
		a.envmap_res[0] = extras_host->envmap_res[0]; a.envmap_res[1] = extras_host->envmap_res[1]; a.envmap_loss_type = extras_host->envmap_loss_type;
	}
	a.x_row_out = extras_host ? extras_host->x_row_index_out : nullptr;
	a.sharpness_data = nullptr; a.sharpness_grid = nullptr; a.sharpness_res[0] = a.sharpness_res[1] = 0;
	if (extras_host && extras_host->sharpness_data && extras_host->sharpness_grid && extras_host->sharpness_res[0] > 0 && extras_host->sharpness_res[1] > 0) {
		a.sharpness_data = extras_host->sharpness_data; a.sharpness_grid = extras_host->sharpness_grid;
		a.sharpness_res[0] = extras_host->sharpness_res[0]; a.sharpness_res[1] = extras_host->sharpness_res[1];
	}
	// n_rays upper-bounds *rays_counter (the number of ray slots the generator filled); one wave per slot
	hipLaunchKernelGGL(compute_loss_kernel, dim3(div_up(n_rays, LOSS_RAYS_PER_BLOCK)), dim3(LOSS_RAYS_PER_BLOCK * 64), 0, (hipStream_t)stream, a);
	NGP_LAUNCH_CHECK("compute_loss_kernel");
	return 0;
}

// nerf_loader.cu:121-169 compute_sharpness: one thread per tile of the sharpness grid; variance of the Laplacian of the luma of read_rgba, one pixel in from the edge
int ngp_hip_compute_sharpness(void* stream, const int32_t* sharpness_res_host, const int32_t* image_res_host, const void* pixels, int image_data_type, float* sharpness_out) {
	const dim3 threads(16, 8, 1), blocks(div_up((uint32_t)sharpness_res_host[0], 16), div_up((uint32_t)sharpness_res_host[1], 8), 1);
	hipLaunchKernelGGL(compute_sharpness_kernel, blocks, threads, 0, (hipStream_t)stream, sharpness_res_host[0], sharpness_res_host[1], image_res_host[0], image_res_host[1], pixels, image_data_type, sharpness_out);
	NGP_LAUNCH_CHECK("compute_sharpness_kernel");
	return 0;
}
// decay_sharpness_grid_nerf (src/testbed_nerf.cu:557-561)
int ngp_hip_decay_grid(void* stream, uint32_t n_elements, float decay, float* grid) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(decay_grid_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, decay, grid);
	NGP_LAUNCH_CHECK("decay_grid_kernel");
	return 0;
}

int ngp_hip_compute_cam_gradient(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint64_t rng_state, uint64_t rng_inc, const uint32_t* rays_counter, int snap_to_pixel_centers,
	float* cam_pos_gradient, float* cam_rot_gradient, uint32_t n_training_images, const NgpImageMeta* metadata, const uint32_t* ray_indices_in,
	const NgpRay* rays_in_unnormalized, const uint32_t* numsteps_in, const NgpCoord* coords_compacted, const float* coords_gradient, const NgpErrorMapCdf* cdf_host,
	const NgpXForm* xforms, float* distortion_gradient, float* distortion_gradient_weight, const int32_t* distortion_resolution_host) {
	if (!n_rays || (!cam_pos_gradient && !cam_rot_gradient && !distortion_gradient)) return 0;
	if (distortion_gradient && (!xforms || !distortion_gradient_weight || !distortion_resolution_host || distortion_resolution_host[0] <= 0 || distortion_resolution_host[1] <= 0)) {
		set_last_error("ngp_hip_compute_cam_gradient: the distortion gradient needs the training transforms, a weight buffer and a resolution", hipErrorInvalidValue); return -1;
	}
	Pcg32 rng; rng.state = rng_state; rng.inc = rng_inc;
	hipLaunchKernelGGL(compute_cam_gradient_kernel, dim3(div_up(n_rays * CAMGRAD_LANES, 256u)), dim3(256), 0, (hipStream_t)stream, n_rays, aabb_from_host(aabb_host), rng,
	                   rays_counter, snap_to_pixel_centers, cam_pos_gradient, cam_rot_gradient, n_training_images, metadata, ray_indices_in, rays_in_unnormalized, numsteps_in,
	                   coords_compacted, coords_gradient, make_error_map_cdf(cdf_host), xforms, distortion_gradient, distortion_gradient_weight,
	                   distortion_gradient ? distortion_resolution_host[0] : 0, distortion_gradient ? distortion_resolution_host[1] : 0);
	NGP_LAUNCH_CHECK("compute_cam_gradient_kernel");
	return 0;
}

int ngp_hip_safe_divide(void* stream, uint32_t n_elements, float* inout, const float* divisor) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(safe_divide_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, inout, divisor);
	NGP_LAUNCH_CHECK("safe_divide_kernel");
	return 0;
}

int ngp_hip_optimizer_step_f32(void* stream, uint32_t n_params, uint32_t step, float learning_rate, float beta1, float beta2, float epsilon, float loss_scale, float ema_decay,
                               const float* grads, float* params, float* first_moments, float* second_moments, float* ema) {
	if (!n_params) return 0;
	if (step == 0) { set_last_error("ngp_hip_optimizer_step_f32: step is the 1-based count of optimizer steps", hipErrorInvalidValue); return -1; }
	const float lr = learning_rate * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
	const float ema_debias_old = 1.0f - powf(ema_decay, (float)(step - 1));
	const float ema_debias_new = 1.0f / (1.0f - powf(ema_decay, (float)step));
	hipLaunchKernelGGL(adam_ema_f32_kernel, dim3(div_up(n_params, 256)), dim3(256), 0, (hipStream_t)stream, n_params, lr, beta1, beta2, epsilon, loss_scale, ema_decay, ema_debias_old, ema_debias_new,
	                   grads, params, first_moments, second_moments, ema);
	NGP_LAUNCH_CHECK("adam_ema_f32_kernel");
	return 0;
}

int ngp_hip_fill_rollover_and_rescale_f16(void* stream, uint32_t n_elements, uint32_t stride, const uint32_t* n_input_elements, uint16_t* inout) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(fill_rollover_and_rescale_f16_kernel, dim3(div_up(n_elements * stride, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, stride, n_input_elements, inout);
	NGP_LAUNCH_CHECK("fill_rollover_and_rescale_f16_kernel");
	return 0;
}
int ngp_hip_fill_rollover_f32(void* stream, uint32_t n_elements, uint32_t stride, const uint32_t* n_input_elements, float* inout) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(fill_rollover_f32_kernel, dim3(div_up(n_elements * stride, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, stride, n_input_elements, inout);
	NGP_LAUNCH_CHECK("fill_rollover_f32_kernel");
	return 0;
}

int ngp_hip_fill_rollover_training(void* stream, uint32_t n_elements, const uint32_t* n_input_elements, uint16_t* dloss, uint32_t dl_stride, float* coords, uint32_t coord_stride_floats,
                                   float* encoded, uint32_t encoded_stride_floats) {
	if (!n_elements) return 0;
	uint32_t widest = dl_stride > coord_stride_floats ? dl_stride : coord_stride_floats;
	if (encoded && encoded_stride_floats > widest) widest = encoded_stride_floats;
	hipLaunchKernelGGL(fill_rollover_training_kernel, dim3(div_up(n_elements * widest, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, n_input_elements, dloss, dl_stride, coords,
	                   coord_stride_floats, encoded, encoded_stride_floats);
	NGP_LAUNCH_CHECK("fill_rollover_training_kernel");
	return 0;
}

int ngp_hip_post_words_and_fill_rollover_training(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst4, uint32_t* zero_words, uint32_t n_zero_words,
                                                  double* sum3_dev, uint32_t n_elements, const uint32_t* n_input_elements, uint16_t* dloss, uint32_t dl_stride, float* coords,
                                                  uint32_t coord_stride_floats, float* encoded, uint32_t encoded_stride_floats) {
	if (!n_elements || !n_input_elements) { set_last_error("ngp_hip_post_words_and_fill_rollover_training: n_elements and n_input_elements are required", hipErrorInvalidValue); return -1; }
	hipLaunchKernelGGL(post_and_rollover_training_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, a, b, c, tag, dst4, zero_words, zero_words ? n_zero_words : 0u, sum3_dev, n_elements,
	                   n_input_elements, dloss, dl_stride, coords, coord_stride_floats, encoded, encoded_stride_floats);
	NGP_LAUNCH_CHECK("post_and_rollover_training_kernel");
	return 0;
}

int ngp_hip_loss_and_gradient(void* stream, int loss_type, uint32_t n, uint32_t dims, float loss_scale, const uint16_t* predictions, uint32_t pred_stride, const float* targets,
                              float* values, uint16_t* gradients, uint32_t grad_stride) {
	if (!n) return 0;
	if (dims < 1 || dims > 4 || pred_stride < dims || grad_stride < 4) { set_last_error("ngp_hip_loss_and_gradient: 1 <= dims <= 4, pred_stride >= dims, grad_stride >= 4", hipErrorInvalidValue); return -1; }
	if (loss_type != NGP_LOSS_L2 && loss_type != NGP_LOSS_RELATIVE_L2 && loss_type != NGP_LOSS_L1 && loss_type != NGP_LOSS_MAPE) { set_last_error("ngp_hip_loss_and_gradient: loss must be L2, RelativeL2, L1 or MAPE", hipErrorInvalidValue); return -1; }
	hipLaunchKernelGGL(loss_and_gradient_kernel, dim3(div_up(n * 4u, 256)), dim3(256), 0, (hipStream_t)stream, loss_type, n, dims, loss_scale, (const half_t*)predictions, pred_stride, targets, values,
	                   (half_t*)gradients, grad_stride);
	NGP_LAUNCH_CHECK("loss_and_gradient_kernel");
	return 0;
}

int ngp_hip_generate_random_uniform(void* stream, uint64_t rng_state, uint64_t rng_inc, uint32_t n_elements, float* out) {
	if (!n_elements) return 0;
	Pcg32 rng; rng.state = rng_state; rng.inc = rng_inc;
	hipLaunchKernelGGL(random_uniform_kernel, dim3(div_up(div_up(n_elements, 4u), 256)), dim3(256), 0, (hipStream_t)stream, n_elements, rng, out);
	NGP_LAUNCH_CHECK("random_uniform_kernel");
	return 0;
}

int ngp_hip_image_stratify2(void* stream, uint32_t n_elements, uint32_t log2_batch_size, float* inout_xy) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(stratify2_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, log2_batch_size, (float2*)inout_xy);
	NGP_LAUNCH_CHECK("stratify2_kernel");
	return 0;
}

int ngp_hip_image_eval_and_snap(void* stream, uint32_t n_elements, const void* texture, int image_data_type, float* positions_xy, const int32_t* resolution_host, float* result,
                                uint32_t stride, int snap_to_pixel_centers, int linear_colors) {
	if (!n_elements) return 0;
	if (stride < 3) { set_last_error("ngp_hip_image_eval_and_snap: stride must be >= 3", hipErrorInvalidValue); return -1; }
	if (image_data_type == 2) hipLaunchKernelGGL(eval_image_and_snap_kernel<half_t>, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, (const half_t*)texture, (float2*)positions_xy,
	                                            resolution_host[0], resolution_host[1], result, stride, snap_to_pixel_centers != 0, linear_colors != 0);
	else if (image_data_type == 3) hipLaunchKernelGGL(eval_image_and_snap_kernel<float>, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, (const float*)texture, (float2*)positions_xy,
	                                                 resolution_host[0], resolution_host[1], result, stride, snap_to_pixel_centers != 0, linear_colors != 0);
	else { set_last_error("ngp_hip_image_eval_and_snap: image_data_type must be 2 (half4) or 3 (float4)", hipErrorInvalidValue); return -1; }
	NGP_LAUNCH_CHECK("eval_image_and_snap_kernel");
	return 0;
}

int ngp_hip_image_init_coords(void* stream, float* positions_xy, const int32_t* res_host, const int32_t* image_res_host, float view_dist, const float* image_pos_host,
                              const float* screen_center_host, int snap_to_pixel_centers, uint32_t sample_index) {
	if (res_host[0] <= 0 || res_host[1] <= 0) return 0;
	const dim3 threads(16, 8, 1), blocks(div_up((uint32_t)res_host[0], 16u), div_up((uint32_t)res_host[1], 8u), 1);
	hipLaunchKernelGGL(init_image_coords_kernel, blocks, threads, 0, (hipStream_t)stream, (float2*)positions_xy, res_host[0], res_host[1], image_res_host[0], image_res_host[1], view_dist,
	                   image_pos_host[0], image_pos_host[1], screen_center_host[0], screen_center_host[1], snap_to_pixel_centers != 0, sample_index);
	NGP_LAUNCH_CHECK("init_image_coords_kernel");
	return 0;
}

int ngp_hip_image_shade(void* stream, const int32_t* res_host, const float* positions_xy, const uint16_t* colors, uint32_t color_stride, float* frame_buffer, float* depth_buffer, int linear_colors) {
	if (res_host[0] <= 0 || res_host[1] <= 0) return 0;
	const dim3 threads(16, 8, 1), blocks(div_up((uint32_t)res_host[0], 16u), div_up((uint32_t)res_host[1], 8u), 1);
	hipLaunchKernelGGL(shade_image_kernel, blocks, threads, 0, (hipStream_t)stream, res_host[0], res_host[1], (const float2*)positions_xy, (const half_t*)colors, color_stride, (float4*)frame_buffer,
	                   depth_buffer, linear_colors != 0);
	NGP_LAUNCH_CHECK("shade_image_kernel");
	return 0;
}

int ngp_hip_image_coords_from_idx(void* stream, uint32_t n_elements, uint32_t offset, float* positions_xy, const int32_t* res_host) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(image_coords_from_idx_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, offset, (float2*)positions_xy, res_host[0], res_host[1]);
	NGP_LAUNCH_CHECK("image_coords_from_idx_kernel");
	return 0;
}

int ngp_hip_image_mse(void* stream, uint32_t n_elements, const float* target, const uint16_t* prediction, uint32_t pred_stride, float* result, int quantize_to_byte) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(image_mse_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, target, (const half_t*)prediction, pred_stride, result, quantize_to_byte != 0);
	NGP_LAUNCH_CHECK("image_mse_kernel");
	return 0;
}

} // extern "C"
