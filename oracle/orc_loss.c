/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_core.h header).
 * Loss / compositing / compaction of one training step.
 * Follows src/testbed_nerf.cu:121-189 (losses), 1263-1278, 1280-1597 (compute_loss_kernel_train_nerf),
 * 3314-3322 (fill_rollover call sites; kernels are [tcnn] common_device.h).
 * Default-off features (envmap, error-map CDF sampling, sharpness, depth supervision, exposure gradients,
 * testbed.h:651-680) are not restated; the always-on error-map deposit (1465-1491) is.
 */
#include "ngp_oracle.h"

typedef struct { float loss[3]; float gradient[3]; } orc_lg;

static inline float orc_copysign1(float a, float b) { return copysignf(a, b); }

/* testbed_nerf.cu:121-189, 1263-1278.  ELossType: common.h:118-126 {L2, L1, Mape, Smape, Huber, LogL1, RelativeL2} */
static orc_lg orc_loss_and_gradient(const float target[3], const float prediction[3], int loss_type) {
	orc_lg r;
	for (int c = 0; c < 3; ++c) {
		float t = target[c], p = prediction[c];
		float diff = p - t;
		switch (loss_type) {
			case ORC_LOSS_RELATIVE_L2: {
				float factor = 1.0f / (p * p + 1e-2f);
				r.loss[c] = diff * diff * factor; r.gradient[c] = 2.0f * diff * factor; break; }
			case ORC_LOSS_L1: r.loss[c] = fabsf(diff); r.gradient[c] = orc_copysign1(1.0f, diff); break;
			case ORC_LOSS_MAPE: {
				float factor = 1.0f / (fabsf(p) + 1e-2f);
				r.loss[c] = fabsf(diff) * factor; r.gradient[c] = orc_copysign1(factor, diff); break; }
			case ORC_LOSS_SMAPE: {
				float factor = 1.0f / (0.5f * (fabsf(p) + fabsf(t)) + 1e-2f);
				r.loss[c] = fabsf(diff) * factor; r.gradient[c] = orc_copysign1(factor, diff); break; }
			case ORC_LOSS_HUBER: {
				const float alpha = 0.1f;
				float ad = fabsf(diff);
				float square = 0.5f / alpha * diff * diff;
				float l = ad > alpha ? (ad - 0.5f * alpha) : square;
				float g = ad > alpha ? (diff > 0 ? 1.0f : -1.0f) : (diff / alpha);
				r.loss[c] = l / 5.0f; r.gradient[c] = g / 5.0f; break; }
			case ORC_LOSS_LOG_L1: {
				float divisor = fabsf(diff) + 1.0f;
				r.loss[c] = logf(divisor); r.gradient[c] = orc_copysign1(1.0f / divisor, diff); break; }
			default: r.loss[c] = diff * diff; r.gradient[c] = 2.0f * diff; break;
		}
	}
	return r;
}

void orc_loss_and_gradient_export(const float* target, const float* prediction, int loss_type, float* loss, float* grad) {
	orc_lg r = orc_loss_and_gradient(target, prediction, loss_type);
	for (int c = 0; c < 3; ++c) { loss[c] = r.loss[c]; grad[c] = r.gradient[c]; }
}

/* testbed_nerf.cu:1280-1597.  mlp_out: fp16 [sample][mlp_stride] (channels 0..3 = rgb,sigma), coords 7 floats/sample.
 * Rays are visited in index order so the compacted order is deterministic; the GPU order is not, tests compare per ray
 * through numsteps_in[i*2+1] (compacted base). */
void orc_compute_loss(
	uint32_t n_rays /* rays_per_batch: normalisation + image_idx */, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc,
	uint32_t max_samples_compacted, uint32_t n_rays_alive /* *rays_counter */, float loss_scale, uint32_t mlp_stride,
	const float background_color_in[3], int color_space_srgb, int train_with_random_bg_color, int train_in_linear_colors,
	uint32_t n_training_images, const orc_image_meta* metadata, const uint16_t* network_output, uint32_t* numsteps_counter,
	const uint32_t* ray_indices_in, const orc_ray* rays_in_unnormalized, uint32_t* numsteps_in, const orc_coord* coords_in_all,
	orc_coord* coords_out_all, uint16_t* dloss_doutput_all /* [max_samples_compacted][mlp_stride] */, int loss_type,
	float* loss_output, int max_level_rand_training, float* max_level_compacted_ptr_all, int rgb_activation, int density_activation,
	int snap_to_pixel_centers, float* error_map, const int32_t error_map_res[2], float mean_density, const float* exposure /* [n_images][3] */,
	float near_distance, const orc_error_map_cdf* cdf, const uint16_t* encoded_in, uint16_t* encoded_out, float depth_supervision_lambda, int depth_loss_type,
	float* exposure_gradient) {
	orc_compute_loss_ex(n_rays, aabb, rng_state, rng_inc, max_samples_compacted, n_rays_alive, loss_scale, mlp_stride, background_color_in, color_space_srgb, train_with_random_bg_color,
	                    train_in_linear_colors, n_training_images, metadata, network_output, numsteps_counter, ray_indices_in, rays_in_unnormalized, numsteps_in, coords_in_all, coords_out_all,
	                    dloss_doutput_all, loss_type, loss_output, max_level_rand_training, max_level_compacted_ptr_all, rgb_activation, density_activation, snap_to_pixel_centers, error_map,
	                    error_map_res, mean_density, exposure, near_distance, cdf, encoded_in, encoded_out, depth_supervision_lambda, depth_loss_type, exposure_gradient, NULL);
}

/* the same kernel with its environment-map arguments (:1289-1292): the map in front of the background colour (:1394-1401), its gradient (:1573-1596) */
void orc_compute_loss_ex(
	uint32_t n_rays, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc,
	uint32_t max_samples_compacted, uint32_t n_rays_alive, float loss_scale, uint32_t mlp_stride,
	const float background_color_in[3], int color_space_srgb, int train_with_random_bg_color, int train_in_linear_colors,
	uint32_t n_training_images, const orc_image_meta* metadata, const uint16_t* network_output, uint32_t* numsteps_counter,
	const uint32_t* ray_indices_in, const orc_ray* rays_in_unnormalized, uint32_t* numsteps_in, const orc_coord* coords_in_all,
	orc_coord* coords_out_all, uint16_t* dloss_doutput_all, int loss_type,
	float* loss_output, int max_level_rand_training, float* max_level_compacted_ptr_all, int rgb_activation, int density_activation,
	int snap_to_pixel_centers, float* error_map, const int32_t error_map_res[2], float mean_density, const float* exposure,
	float near_distance, const orc_error_map_cdf* cdf, const uint16_t* encoded_in, uint16_t* encoded_out, float depth_supervision_lambda, int depth_loss_type,
	float* exposure_gradient, const orc_loss_extras* ex) {
	const float* envmap_data = ex && ex->envmap_data && ex->envmap_res[0] > 0 && ex->envmap_res[1] > 0 ? ex->envmap_data : NULL;
	float* envmap_gradient = envmap_data ? ex->envmap_gradient : NULL;
	for (uint32_t i = 0; i < n_rays_alive; ++i) {
		uint32_t numsteps = numsteps_in[i * 2 + 0];
		uint32_t base = numsteps_in[i * 2 + 1];
		const orc_coord* coords_in = coords_in_all + base;
		const uint16_t* no = network_output + (size_t)base * mlp_stride;

		float T = 1.f;
		const float EPSILON = 1e-4f;
		float rgb_ray[3] = {0, 0, 0};
		orc_vec3 hitpoint = orc_v3(0.f, 0.f, 0.f);
		float depth_ray = 0.f;
		uint32_t compacted_numsteps = 0;
		orc_vec3 ray_o = rays_in_unnormalized[i].o;
		for (; compacted_numsteps < numsteps; ++compacted_numsteps) {
			if (T < EPSILON) break;
			const uint16_t* lo = no + (size_t)compacted_numsteps * mlp_stride;
			float rgb[3];
			for (int c = 0; c < 3; ++c) rgb[c] = orc_network_to_rgb(orc_h2f(lo[c]), rgb_activation);
			const orc_coord* ci = &coords_in[compacted_numsteps];
			orc_vec3 pos = orc_unwarp_position(orc_v3(ci->pos[0], ci->pos[1], ci->pos[2]), aabb);
			float dt = orc_unwarp_dt(ci->dt);
			float cur_depth = orc_norm(orc_sub(pos, ray_o));
			float density = orc_network_to_density(orc_h2f(lo[3]), density_activation);
			float alpha = 1.f - expf(-density * dt);
			float weight = alpha * T;
			for (int c = 0; c < 3; ++c) rgb_ray[c] += weight * rgb[c];
			hitpoint = orc_add(hitpoint, orc_scale(pos, weight));   /* :1367 */
			depth_ray += weight * cur_depth;
			T *= (1.f - alpha);
		}
		hitpoint = orc_scale(hitpoint, 1.0f / (1.0f - T));          /* :1374 (hitpoint /= 1 - T) */

		uint32_t ray_idx = ray_indices_in[i];
		orc_pcg32 rng = {rng_state, rng_inc};
		orc_pcg32_advance(&rng, (int64_t)((uint64_t)(uint32_t)(ray_idx * ORC_N_MAX_RANDOM_SAMPLES_PER_RAY)));
		float img_pdf = 1.0f, xy_pdf = 1.0f;
		uint32_t img = orc_image_idx(ray_idx, n_rays, n_training_images, cdf ? cdf->cdf_img : NULL, &img_pdf);
		const orc_image_meta* md = &metadata[img];
		float xy[2];
		orc_nerf_random_image_pos_training(&rng, md->res, snap_to_pixel_centers, cdf, img, xy, &xy_pdf);
		float max_level = max_level_rand_training ? (orc_pcg32_next_float(&rng) * 2.0f) : 1.0f;

		float background_color[3] = {background_color_in[0], background_color_in[1], background_color_in[2]};
		if (train_with_random_bg_color) {
			for (int c = 0; c < 3; ++c) background_color[c] = orc_pcg32_next_float(&rng);
		}
		for (int c = 0; c < 3; ++c) background_color[c] = orc_srgb_to_linear(background_color[c]);
		float env_dir[3] = {0.f, 0.f, 1.f};
		if (envmap_data) {   /* :1394-1401 */
			const orc_vec3 d = orc_normalized(rays_in_unnormalized[i].d);
			env_dir[0] = d.x; env_dir[1] = d.y; env_dir[2] = d.z;
			float e[4];
			orc_read_envmap(envmap_data, ex->envmap_res, env_dir, e);
			for (int c = 0; c < 3; ++c) background_color[c] = e[c] + background_color[c] * (1.0f - e[3]);
		}

		float exposure_scale[3];
		for (int c = 0; c < 3; ++c) exposure_scale[c] = expf(0.6931471805599453f * exposure[img * 3 + c]);
		float texsamp[4];
		orc_read_rgba(xy, md->res, md->pixels, md->image_data_type, texsamp);

		float rgbtarget[3];
		if (train_in_linear_colors || !color_space_srgb) {
			for (int c = 0; c < 3; ++c) rgbtarget[c] = exposure_scale[c] * texsamp[c] + (1.0f - texsamp[3]) * background_color[c];
			if (!train_in_linear_colors) {
				for (int c = 0; c < 3; ++c) { rgbtarget[c] = orc_linear_to_srgb(rgbtarget[c]); background_color[c] = orc_linear_to_srgb(background_color[c]); }
			}
		} else {
			for (int c = 0; c < 3; ++c) background_color[c] = orc_linear_to_srgb(background_color[c]);
			if (texsamp[3] > 0) {
				for (int c = 0; c < 3; ++c) rgbtarget[c] = orc_linear_to_srgb(exposure_scale[c] * texsamp[c] / texsamp[3]) * texsamp[3] + (1.0f - texsamp[3]) * background_color[c];
			} else {
				for (int c = 0; c < 3; ++c) rgbtarget[c] = background_color[c];
			}
		}

		if (compacted_numsteps == numsteps) {
			for (int c = 0; c < 3; ++c) rgb_ray[c] += T * background_color[c];
		}

		uint32_t compacted_base = *numsteps_counter; *numsteps_counter += compacted_numsteps;
		uint32_t room = max_samples_compacted - (max_samples_compacted < compacted_base ? max_samples_compacted : compacted_base);
		compacted_numsteps = room < compacted_numsteps ? room : compacted_numsteps;
		numsteps_in[i * 2 + 0] = compacted_numsteps;
		numsteps_in[i * 2 + 1] = compacted_base;
		if (compacted_numsteps == 0) continue;

		float* max_level_compacted_ptr = max_level_compacted_ptr_all ? max_level_compacted_ptr_all + compacted_base : NULL;
		orc_coord* coords_out = coords_out_all + compacted_base;
		uint16_t* dloss_doutput = dloss_doutput_all + (size_t)compacted_base * mlp_stride;

		orc_lg lg = orc_loss_and_gradient(rgbtarget, rgb_ray, loss_type);
		if (exposure_gradient) {   /* 1558-1572 */
			for (int c = 0; c < 3; ++c) {
				float dloss_by_dgt = -lg.gradient[c] / xy_pdf;
				if (!train_in_linear_colors) dloss_by_dgt /= orc_srgb_to_linear_derivative(rgbtarget[c]);
				exposure_gradient[img * 3 + c] += loss_scale * dloss_by_dgt * exposure_scale[c] * 0.6931471805599453f;
			}
		}
		/* depth supervision (1450-1452) */
		float depth_loss_gradient = 0.0f;
		if (depth_supervision_lambda > 0.0f) {
			float dval = -1.0f;
			if (md->depth) {
				int px = (int)(xy[0] * (float)md->res[0]), py = (int)(xy[1] * (float)md->res[1]);
				px = px > 0 ? px : 0; px = px < md->res[0] - 1 ? px : md->res[0] - 1;
				py = py > 0 ? py : 0; py = py < md->res[1] - 1 ? py : md->res[1] - 1;
				dval = md->depth[(size_t)px + (size_t)py * (size_t)md->res[0]];
			}
			const float target_depth = orc_norm(rays_in_unnormalized[i].d) * dval;
			const float t3[3] = {target_depth, target_depth, target_depth}, p3[3] = {depth_ray, depth_ray, depth_ray};
			orc_lg lgd = orc_loss_and_gradient(t3, p3, depth_loss_type);
			depth_loss_gradient = target_depth > 0.0f ? depth_supervision_lambda * lgd.gradient[0] : 0.0f;
		}
		{ const float pdf = img_pdf * xy_pdf; for (int c = 0; c < 3; ++c) lg.loss[c] /= pdf; }   /* 1448; == 1 without CDF sampling */
		float mean_loss = (lg.loss[0] + lg.loss[1] + lg.loss[2]) / 3.0f;
		if (loss_output) loss_output[i] = mean_loss / (float)n_rays;

		if (error_map) {
			float posx = xy[0] * (float)error_map_res[0] - 0.5f, posy = xy[1] * (float)error_map_res[1] - 0.5f;
			posx = fminf(fmaxf(posx, 0.0f), (float)error_map_res[0] - (1.0f + 1e-4f));
			posy = fminf(fmaxf(posy, 0.0f), (float)error_map_res[1] - (1.0f + 1e-4f));
			int pix = (int)posx, piy = (int)posy;
			float wx = posx - (float)pix, wy = posy - (float)piy;
			/* 1470: idx = pos_int.cwiseMin(resolution - 2).cwiseMax(0) — `resolution` is the IMAGE resolution there */
			int ix = pix < md->res[0] - 2 ? pix : md->res[0] - 2; ix = ix > 0 ? ix : 0;
			int iy = piy < md->res[1] - 2 ? piy : md->res[1] - 2; iy = iy > 0 ? iy : 0;
			size_t b = (size_t)img * (size_t)error_map_res[0] * (size_t)error_map_res[1];
			if (ex && ex->sharpness_data && orc_aabb_contains(aabb, hitpoint)) {   /* :1476-1485: the error a ray deposits is scaled by how sharp its image tile is relative to the sharpest one that saw the cell */
				int sx = (int)(xy[0] * (float)ex->sharpness_res[0]), sy = (int)(xy[1] * (float)ex->sharpness_res[1]);
				sx = sx > 0 ? sx : 0; sx = sx < ex->sharpness_res[0] - 1 ? sx : ex->sharpness_res[0] - 1;
				sy = sy > 0 ? sy : 0; sy = sy < ex->sharpness_res[1] - 1 ? sy : ex->sharpness_res[1] - 1;
				const float sharp = ex->sharpness_data[(size_t)img * ex->sharpness_res[0] * ex->sharpness_res[1] + (size_t)sy * ex->sharpness_res[0] + sx] + 1e-6f;
				const uint32_t mip = (uint32_t)orc_mip_from_pos(hitpoint, ORC_NERF_CASCADES - 1);
				float* cell = &ex->sharpness_grid[orc_cascaded_grid_idx_at(hitpoint, mip) + (size_t)ORC_NERF_GRIDSIZE * ORC_NERF_GRIDSIZE * ORC_NERF_GRIDSIZE * mip];
				float grid_sharp = *cell;
				if (sharp > grid_sharp) *cell = sharp;        /* atomicMax on the bit pattern of positive floats */
				grid_sharp = fmaxf(sharp, grid_sharp);
				mean_loss *= fmaxf(sharp / grid_sharp, 0.01f);
			}
			error_map[b + (size_t)iy * error_map_res[0] + ix] += (1 - wx) * (1 - wy) * mean_loss;
			error_map[b + (size_t)iy * error_map_res[0] + ix + 1] += wx * (1 - wy) * mean_loss;
			error_map[b + (size_t)(iy + 1) * error_map_res[0] + ix] += (1 - wx) * wy * mean_loss;
			error_map[b + (size_t)(iy + 1) * error_map_res[0] + ix + 1] += wx * wy * mean_loss;
		}

		float ls = loss_scale / (float)n_rays;
		const float output_l2_reg = rgb_activation == ORC_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
		const float output_l1_reg_density = mean_density < ORC_NERF_MIN_OPTICAL_THICKNESS ? 1e-4f : 0.0f;

		float rgb_ray2[3] = {0.f, 0.f, 0.f};
		float depth_ray2 = 0.f;
		T = 1.f;
		for (uint32_t j = 0; j < compacted_numsteps; ++j) {
			if (max_level_rand_training && max_level_compacted_ptr) max_level_compacted_ptr[j] = max_level;
			coords_out[j] = coords_in[j];
			if (encoded_in) memcpy(encoded_out + ((size_t)compacted_base + j) * 32, encoded_in + ((size_t)base + j) * 32, 64);   /* see ngp_hip.h "Forward pass" */
			const orc_coord* ci = &coords_in[j];
			orc_vec3 pos = orc_unwarp_position(orc_v3(ci->pos[0], ci->pos[1], ci->pos[2]), aabb);
			float depth = orc_norm(orc_sub(pos, ray_o));
			float dt = orc_unwarp_dt(ci->dt);
			const uint16_t* lo = no + (size_t)j * mlp_stride;
			float lof[4];
			for (int c = 0; c < 4; ++c) lof[c] = orc_h2f(lo[c]);
			float rgb[3];
			for (int c = 0; c < 3; ++c) rgb[c] = orc_network_to_rgb(lof[c], rgb_activation);
			float density = orc_network_to_density(lof[3], density_activation);
			float alpha = 1.f - expf(-density * dt);
			float weight = alpha * T;
			for (int c = 0; c < 3; ++c) rgb_ray2[c] += weight * rgb[c];
			depth_ray2 += weight * depth;
			T *= (1.f - alpha);

			float suffix[3], dloss_by_drgb[3];
			for (int c = 0; c < 3; ++c) { suffix[c] = rgb_ray[c] - rgb_ray2[c]; dloss_by_drgb[c] = weight * lg.gradient[c]; }

			uint16_t* dl = dloss_doutput + (size_t)j * mlp_stride;
			for (int c = 0; c < 3; ++c) {
				dl[c] = orc_f2h(ls * (dloss_by_drgb[c] * orc_network_to_rgb_derivative(lof[c], rgb_activation) + fmaxf(0.0f, output_l2_reg * lof[c])));
			}
			float density_derivative = orc_network_to_density_derivative(lof[3], density_activation);
			float dotv = lg.gradient[0] * (T * rgb[0] - suffix[0]) + lg.gradient[1] * (T * rgb[1] - suffix[1]) + lg.gradient[2] * (T * rgb[2] - suffix[2]);
			const float depth_supervision = depth_loss_gradient * (T * depth - (depth_ray - depth_ray2));   /* 1536-1537 */
			float dloss_by_dmlp = density_derivative * (dt * (dotv + depth_supervision));
			dl[3] = orc_f2h(
				ls * dloss_by_dmlp +
				(lof[3] < 0.0f ? -output_l1_reg_density : 0.0f) +
				(lof[3] > -10.0f && depth < near_distance ? 1e-4f : 0.0f));
		}
		if (compacted_numsteps == numsteps && envmap_gradient) {   /* :1573-1596; deposit_envmap_gradient, envmap.cuh:65-103 (value and weight pass through fp16) */
			orc_lg lge = lg;
			if (ex->envmap_loss_type != loss_type) lge = orc_loss_and_gradient(rgbtarget, rgb_ray, ex->envmap_loss_type);
			const float PI = 3.14159265358979323846f;
			const float dx = env_dir[2], dy = -env_dir[0], dz = env_dir[1];
			const float theta = acosf(fminf(fmaxf(dz, -1.0f), 1.0f)), phi = atan2f(dy, dx);
			const float fx = (phi / (2.0f * PI) + 0.5f) * (float)(ex->envmap_res[0] - 1), fy = (theta / PI) * (float)(ex->envmap_res[1] - 1);
			const int tx = (int)fx, ty = (int)fy;
			const float wx = fx - (float)tx, wy = fy - (float)ty;
			for (int c = 0; c < 3; ++c) {
				float d = T * lge.gradient[c];
				if (!train_in_linear_colors) d /= orc_srgb_to_linear_derivative(background_color[c]);
				const float v16 = orc_h2f(orc_f2h(loss_scale * d));
				for (int k = 0; k < 4; ++k) {
					int x = tx + (k & 1), y = ty + (k >> 1);
					if (x < 0) x += ex->envmap_res[0]; else if (x >= ex->envmap_res[0]) x -= ex->envmap_res[0];
					y = y < ex->envmap_res[1] - 1 ? y : ex->envmap_res[1] - 1; y = y > 0 ? y : 0;
					const float w16 = orc_h2f(orc_f2h(((k & 1) ? wx : 1 - wx) * ((k >> 1) ? wy : 1 - wy)));
					envmap_gradient[((size_t)x + (size_t)y * ex->envmap_res[0]) * 4 + c] += orc_h2f(orc_f2h(v16 * w16));
				}
			}
		}
	}
}

/* [tcnn] common_device.h fill_rollover / fill_rollover_and_rescale, call sites testbed_nerf.cu:3314-3322:
 *   n_input = *n_input_elements_ptr * stride; n_total = n_elements * stride;
 *   if (i < n_input || i >= n_total || n_input == 0) return;
 *   fill_rollover:             inout[i] = inout[i % n_input]
 *   fill_rollover_and_rescale: inout[i] = (T)((float)inout[i % n_input] * n_input / n_total)
 * i.e. only the wrapped-around copies are rescaled; the kept originals are left untouched. */
/* compute_cam_gradient_train_nerf (src/testbed_nerf.cu:1600-1712), the extrinsics part (cam_pos_gradient / cam_rot_gradient).
 * numsteps_in holds the COMPACTED (numsteps, base) pairs compute_loss left (:1472-1473); coords / coords_gradient are the compacted batch and the
 * network's input gradient ([sample][6]: d/dpos xyz, d/ddir xyz in warped coordinates, what tcnn's backward writes with dL_dinput).
 * The distortion branch (:1671-1683) is out of scope; cam_focal_length_gradient is a parameter the reference kernel never writes. */
void orc_compute_cam_gradient(
	uint32_t n_rays, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc, uint32_t n_rays_alive, int snap_to_pixel_centers,
	float* cam_pos_gradient /* [n_images][3] or NULL, accumulated */, float* cam_rot_gradient /* likewise */, uint32_t n_training_images,
	const orc_image_meta* metadata, const uint32_t* ray_indices_in, const orc_ray* rays_in_unnormalized, const uint32_t* numsteps_in,
	const orc_coord* coords_all, const float* coords_gradient_all /* [sample][6] */, const orc_error_map_cdf* cdf) {
	orc_compute_cam_gradient_ex(n_rays, aabb, rng_state, rng_inc, n_rays_alive, snap_to_pixel_centers, cam_pos_gradient, cam_rot_gradient, n_training_images, metadata, ray_indices_in,
	                            rays_in_unnormalized, numsteps_in, coords_all, coords_gradient_all, cdf, NULL, NULL, NULL, NULL);
}

/* deposit_image_gradient<2> (common_device.cuh:112-143) */
static void orc_deposit_image_gradient2(const float value[2], float* gradient, float* gradient_weight, const int32_t res[2], const float pos[2]) {
	const float fx = pos[0] * (float)(res[0] - 1), fy = pos[1] * (float)(res[1] - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	for (int k = 0; k < 4; ++k) {
		int x = tx + (k & 1), y = ty + (k >> 1);
		x = x < res[0] - 1 ? x : res[0] - 1; x = x > 0 ? x : 0;
		y = y < res[1] - 1 ? y : res[1] - 1; y = y > 0 ? y : 0;
		const float w = ((k & 1) ? wx : 1 - wx) * ((k >> 1) ? wy : 1 - wy);
		for (int c = 0; c < 2; ++c) {
			gradient[((size_t)x + (size_t)y * res[0]) * 2 + c] += value[c] * w;
			gradient_weight[((size_t)x + (size_t)y * res[0]) * 2 + c] += w;
		}
	}
}

/* the same kernel with its lens-distortion branch (:1671-1685) */
void orc_compute_cam_gradient_ex(
	uint32_t n_rays, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc, uint32_t n_rays_alive, int snap_to_pixel_centers,
	float* cam_pos_gradient, float* cam_rot_gradient, uint32_t n_training_images,
	const orc_image_meta* metadata, const uint32_t* ray_indices_in, const orc_ray* rays_in_unnormalized, const uint32_t* numsteps_in,
	const orc_coord* coords_all, const float* coords_gradient_all, const orc_error_map_cdf* cdf,
	const orc_xform* xforms, float* distortion_gradient /* [h][w][2] or NULL, accumulated */, float* distortion_gradient_weight, const int32_t* distortion_resolution) {
	const orc_vec3 diag = orc_sub(aabb->max, aabb->min);
	for (uint32_t i = 0; i < n_rays_alive; ++i) {
		const uint32_t numsteps = numsteps_in[i * 2 + 0];
		if (numsteps == 0) continue;                                         /* :1633-1636 */
		const uint32_t base = numsteps_in[i * 2 + 1];
		const orc_coord* coords = coords_all + base;
		const float* cg = coords_gradient_all + (size_t)base * 6;
		const uint32_t ray_idx = ray_indices_in[i];
		const uint32_t img = orc_image_idx(ray_idx, n_rays, n_training_images, cdf ? cdf->cdf_img : NULL, NULL);
		const orc_vec3 ray_o = rays_in_unnormalized[i].o;
		const orc_vec3 ray_d = orc_normalized(rays_in_unnormalized[i].d);
		orc_vec3 go = orc_v3(0, 0, 0), gd = orc_v3(0, 0, 0);
		for (uint32_t j = 0; j < numsteps; ++j) {
			const orc_vec3 warped = orc_v3(coords[j].pos[0], coords[j].pos[1], coords[j].pos[2]);
			/* warp_position_derivative = 1 / aabb.diag() (:288-290) */
			const orc_vec3 pg = orc_v3(cg[j * 6 + 0] * (1.0f / diag.x), cg[j * 6 + 1] * (1.0f / diag.y), cg[j * 6 + 2] * (1.0f / diag.z));
			go = orc_add(go, pg);
			const orc_vec3 pos = orc_unwarp_position(warped, aabb);
			const float t = orc_norm(orc_sub(pos, ray_o));
			/* warp_direction_derivative = 0.5 (:300-302) */
			const orc_vec3 dg = orc_v3(cg[j * 6 + 3] * 0.5f, cg[j * 6 + 4] * 0.5f, cg[j * 6 + 5] * 0.5f);
			gd = orc_add(gd, orc_add(orc_scale(pg, t), dg));
		}
		orc_pcg32 rng = {rng_state, rng_inc};
		orc_pcg32_advance(&rng, (int64_t)((uint64_t)(uint32_t)(ray_idx * ORC_N_MAX_RANDOM_SAMPLES_PER_RAY)));
		float xy_pdf = 1.0f, xy[2];
		orc_nerf_random_image_pos_training(&rng, metadata[img].res, snap_to_pixel_centers, cdf, img, xy, &xy_pdf);
		if (distortion_gradient) {   /* :1673-1685 */
			const orc_vec3 og = orc_sub(gd, orc_scale(ray_d, orc_dot(gd, ray_d)));
			const float* m = xforms[img].start;
			const double a = m[0], b = m[3], c = m[6], d = m[1], e = m[4], f = m[7], g = m[2], h = m[5], k = m[8];   /* inverse of the rotation block by cofactors */
			const double A = e * k - f * h, B = -(d * k - f * g), C = d * h - e * g, det = a * A + b * B + c * C;
			const double inv[9] = {A / det, B / det, C / det, -(b * k - c * h) / det, (a * k - c * g) / det, -(a * h - b * g) / det, (b * f - c * e) / det, -(a * f - c * d) / det, (a * e - b * d) / det};
			const float ipg[2] = {(float)(inv[0] * og.x + inv[3] * og.y + inv[6] * og.z) / xy_pdf, (float)(inv[1] * og.x + inv[4] * og.y + inv[7] * og.z) / xy_pdf};
			orc_deposit_image_gradient2(ipg, distortion_gradient, distortion_gradient_weight, distortion_resolution, xy);
		}
		if (cam_pos_gradient) {
			cam_pos_gradient[img * 3 + 0] += go.x / xy_pdf; cam_pos_gradient[img * 3 + 1] += go.y / xy_pdf; cam_pos_gradient[img * 3 + 2] += go.z / xy_pdf;
		}
		if (cam_rot_gradient) {
			const orc_vec3 aa = orc_v3(ray_d.y * gd.z - ray_d.z * gd.y, ray_d.z * gd.x - ray_d.x * gd.z, ray_d.x * gd.y - ray_d.y * gd.x);                           /* :1697-1701 */
			cam_rot_gradient[img * 3 + 0] += aa.x / xy_pdf; cam_rot_gradient[img * 3 + 1] += aa.y / xy_pdf; cam_rot_gradient[img * 3 + 2] += aa.z / xy_pdf;
		}
	}
}

/* safe_divide (:2039-2045) */
void orc_safe_divide(uint32_t n, float* inout, const float* divisor) { for (uint32_t i = 0; i < n; ++i) inout[i] = divisor[i] > 0.0f ? (inout[i] / divisor[i]) : 0.0f; }

/* [tcnn] Adam (+ Ema) on a TrainableBuffer<N, 2, float> (envmap.cuh / trainable_buffer.cuh users; Trainer::optimizer_step at :2955, :3091): every
 * parameter is a non-matrix parameter (zero gradient: skipped, no l2_reg); fp32 throughout.  `step` 1-based, ema NULL without the Ema wrapper. */
void orc_optimizer_step_f32(uint32_t n, uint32_t step, float base_lr_after_decay, float beta1, float beta2, float epsilon, float loss_scale, float ema_decay,
                            const float* grads, float* params, float* m1, float* m2, float* ema) {
	const float lr = base_lr_after_decay * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
	const float ema_debias_old = 1.0f - powf(ema_decay, (float)(step - 1)), ema_debias_new = 1.0f / (1.0f - powf(ema_decay, (float)step));
	for (uint32_t i = 0; i < n; ++i) {
		const float g = grads[i] / loss_scale;
		float w = params[i];
		if (g != 0.0f) {
			const float fm = m1[i] = beta1 * m1[i] + (1.0f - beta1) * g;
			const float sm = m2[i] = beta2 * m2[i] + (1.0f - beta2) * (g * g);
			w = w - (lr / (sqrtf(sm) + epsilon)) * fm;
			params[i] = w;
		}
		if (ema) ema[i] = (ema[i] * ema_decay * ema_debias_old + w * (1.0f - ema_decay)) * ema_debias_new;
	}
}

void orc_fill_rollover_and_rescale_f16(uint32_t n_elements, uint32_t stride, uint32_t n_input_elements, uint16_t* inout) {
	size_t total = (size_t)n_elements * stride, avail = (size_t)n_input_elements * stride;
	if (avail == 0 || avail >= total) return;
	for (size_t i = avail; i < total; ++i) inout[i] = orc_f2h(orc_h2f(inout[i % avail]) * (float)avail / (float)total);
}
void orc_fill_rollover_f32(uint32_t n_elements, uint32_t stride, uint32_t n_input_elements, float* inout) {
	size_t total = (size_t)n_elements * stride, avail = (size_t)n_input_elements * stride;
	if (avail == 0 || avail >= total) return;
	for (size_t i = avail; i < total; ++i) inout[i] = inout[i % avail];
}

/* common_device.cuh:562-590 from_rgba32<__half> (white / black -> transparent are applied to the bytes beforehand, nerf_loader.cu:59-81) */
void orc_image_from_rgba32_f16(uint64_t n_pixels, const uint8_t* rgba8, uint16_t* out_half4, uint32_t mask_color) {
	for (uint64_t i = 0; i < n_pixels; ++i) {
		uint32_t v; memcpy(&v, rgba8 + i * 4, 4);
		float alpha = (float)(v >> 24) * (1.0f / 255.0f);
		uint16_t* o = out_half4 + i * 4;
		o[0] = orc_f2h(orc_srgb_to_linear((float)(v & 0xffu) * (1.0f / 255.0f)) * alpha);
		o[1] = orc_f2h(orc_srgb_to_linear((float)((v >> 8) & 0xffu) * (1.0f / 255.0f)) * alpha);
		o[2] = orc_f2h(orc_srgb_to_linear((float)((v >> 16) & 0xffu) * (1.0f / 255.0f)) * alpha);
		o[3] = orc_f2h(alpha);
		if (mask_color != 0 && mask_color == v) o[0] = o[1] = o[2] = o[3] = orc_f2h(-1.0f);
	}
}
/* nerf_loader.cu:102-123 sharpen<T> with center_w = 4 + 1 / amount (:817-823); half4 (is_half) or float4 pixels */
void orc_image_sharpen(uint64_t n_pixels, uint32_t w, const void* pix, void* dest, int is_half, float sharpen_amount) {
	const float center_w = 4.f + 1.f / sharpen_amount, inv_totalw = 1.f / (center_w - 4.f);
	const uint16_t* ph = (const uint16_t*)pix; const float* pf = (const float*)pix;
#define ORC_PX(k) (is_half ? orc_h2f(ph[k]) : pf[k])
	for (uint64_t i = 0; i < n_pixels; ++i) {
		float rgba[4];
		for (int j = 0; j < 4; ++j) rgba[j] = ORC_PX(i * 4 + j) * center_w;
		int64_t i2 = (int64_t)i - 1; if (i2 < 0) i2 = 0; i2 *= 4;
		for (int j = 0; j < 4; ++j) rgba[j] -= ORC_PX(i2++);
		i2 = (int64_t)i - w; if (i2 < 0) i2 = 0; i2 *= 4;
		for (int j = 0; j < 4; ++j) rgba[j] -= ORC_PX(i2++);
		i2 = (int64_t)i + 1; if (i2 >= (int64_t)n_pixels) i2 -= (int64_t)n_pixels; i2 *= 4;
		for (int j = 0; j < 4; ++j) rgba[j] -= ORC_PX(i2++);
		i2 = (int64_t)i + w; if (i2 >= (int64_t)n_pixels) i2 -= (int64_t)n_pixels; i2 *= 4;
		for (int j = 0; j < 4; ++j) rgba[j] -= ORC_PX(i2++);
		for (int j = 0; j < 4; ++j) {
			const float v = fmaxf(0.f, rgba[j] * inv_totalw);
			if (is_half) ((uint16_t*)dest)[i * 4 + j] = orc_f2h(v); else ((float*)dest)[i * 4 + j] = v;
		}
	}
#undef ORC_PX
}

/* read_rgba(Vector2i px, ...) (common_device.cuh:677-705) through the float-position reader: the centre of pixel px maps back to px */
static void orc_read_rgba_px(const int32_t px[2], const int32_t res[2], const void* pixels, int type, float out[4]) {
	const float xy[2] = {((float)px[0] + 0.5f) / (float)res[0], ((float)px[1] + 0.5f) / (float)res[1]};
	orc_read_rgba(xy, res, pixels, type, out);
}

/* nerf_loader.cu:121-169 compute_sharpness: per tile of a sharpness_res grid over the image, the variance of the Laplacian of the luma of read_rgba (one pixel in from the edge) */
void orc_compute_sharpness(const int32_t sharpness_res[2], const int32_t image_res[2], const void* pixels, int image_data_type, float* sharpness_out) {
	for (int y = 0; y < sharpness_res[1]; ++y) for (int x = 0; x < sharpness_res[0]; ++x) {
		int x1 = (x * image_res[0]) / sharpness_res[0], x2 = ((x + 1) * image_res[0]) / sharpness_res[0];
		int y1 = (y * image_res[1]) / sharpness_res[1], y2 = ((y + 1) * image_res[1]) / sharpness_res[1];
		x1 = x1 > 1 ? x1 : 1; y1 = y1 > 1 ? y1 : 1;
		x2 = x2 < image_res[0] - 2 ? x2 : image_res[0] - 2; y2 = y2 < image_res[1] - 2 ? y2 : image_res[1] - 2;
		float tot_lap = 0.f, tot_lap2 = 0.f, tot_lum = 0.f;
		const float scal = 1.f / (float)((x2 - x1) * (y2 - y1));
		for (int yy = y1; yy < y2; ++yy) for (int xx = x1; xx < x2; ++xx) {
			float c[4], n[4], e[4], s_[4], w[4];
			const int32_t pc[2] = {xx, yy}, pn[2] = {xx, yy - 1}, pw[2] = {xx - 1, yy}, ps[2] = {xx, yy + 1}, pe[2] = {xx + 1, yy};
			orc_read_rgba_px(pc, image_res, pixels, image_data_type, c); orc_read_rgba_px(pn, image_res, pixels, image_data_type, n);
			orc_read_rgba_px(pw, image_res, pixels, image_data_type, w); orc_read_rgba_px(ps, image_res, pixels, image_data_type, s_);
			orc_read_rgba_px(pe, image_res, pixels, image_data_type, e);
#define ORC_LUMA(v) ((v)[0] * 0.2126f + (v)[1] * 0.7152f + (v)[2] * 0.0722f)
			const float lum = ORC_LUMA(c);
			const float lap = lum * 4.f - ORC_LUMA(n) - ORC_LUMA(e) - ORC_LUMA(s_) - ORC_LUMA(w);
#undef ORC_LUMA
			tot_lap += lap; tot_lap2 += lap * lap; tot_lum += lum;
		}
		tot_lap *= scal; tot_lap2 *= scal; tot_lum *= scal;
		sharpness_out[x + (size_t)y * sharpness_res[0]] = tot_lap2 - tot_lap * tot_lap;
	}
}
