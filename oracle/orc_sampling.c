/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_core.h header).
 * Occupancy ("density") grid maintenance + training ray marching.
 * Follows src/testbed_nerf.cu:369-610 (grid kernels), 1047-1260 (sampling),
 * 2844-2859 (mean + bitfield + pooling).
 */
#include "ngp_oracle.h"

/* random_val.cuh:90-146 Sobol direction numbers (Burley 2019 / Joe-Kuo), 5 dims */
static const uint32_t k_sobol_directions[5][32] = {
	{0x80000000, 0x40000000, 0x20000000, 0x10000000, 0x08000000, 0x04000000, 0x02000000, 0x01000000,
	 0x00800000, 0x00400000, 0x00200000, 0x00100000, 0x00080000, 0x00040000, 0x00020000, 0x00010000,
	 0x00008000, 0x00004000, 0x00002000, 0x00001000, 0x00000800, 0x00000400, 0x00000200, 0x00000100,
	 0x00000080, 0x00000040, 0x00000020, 0x00000010, 0x00000008, 0x00000004, 0x00000002, 0x00000001},
	{0x80000000, 0xc0000000, 0xa0000000, 0xf0000000, 0x88000000, 0xcc000000, 0xaa000000, 0xff000000,
	 0x80800000, 0xc0c00000, 0xa0a00000, 0xf0f00000, 0x88880000, 0xcccc0000, 0xaaaa0000, 0xffff0000,
	 0x80008000, 0xc000c000, 0xa000a000, 0xf000f000, 0x88008800, 0xcc00cc00, 0xaa00aa00, 0xff00ff00,
	 0x80808080, 0xc0c0c0c0, 0xa0a0a0a0, 0xf0f0f0f0, 0x88888888, 0xcccccccc, 0xaaaaaaaa, 0xffffffff},
	{0x80000000, 0xc0000000, 0x60000000, 0x90000000, 0xe8000000, 0x5c000000, 0x8e000000, 0xc5000000,
	 0x68800000, 0x9cc00000, 0xee600000, 0x55900000, 0x80680000, 0xc09c0000, 0x60ee0000, 0x90550000,
	 0xe8808000, 0x5cc0c000, 0x8e606000, 0xc5909000, 0x6868e800, 0x9c9c5c00, 0xeeee8e00, 0x5555c500,
	 0x8000e880, 0xc0005cc0, 0x60008e60, 0x9000c590, 0xe8006868, 0x5c009c9c, 0x8e00eeee, 0xc5005555},
	{0x80000000, 0xc0000000, 0x20000000, 0x50000000, 0xf8000000, 0x74000000, 0xa2000000, 0x93000000,
	 0xd8800000, 0x25400000, 0x59e00000, 0xe6d00000, 0x78080000, 0xb40c0000, 0x82020000, 0xc3050000,
	 0x208f8000, 0x51474000, 0xfbea2000, 0x75d93000, 0xa0858800, 0x914e5400, 0xdbe79e00, 0x25db6d00,
	 0x58800080, 0xe54000c0, 0x79e00020, 0xb6d00050, 0x800800f8, 0xc00c0074, 0x200200a2, 0x50050093},
	{0x80000000, 0x40000000, 0x20000000, 0xb0000000, 0xf8000000, 0xdc000000, 0x7a000000, 0x9d000000,
	 0x5a800000, 0x2fc00000, 0xa1600000, 0xf0b00000, 0xda880000, 0x6fc40000, 0x81620000, 0x40bb0000,
	 0x22878000, 0xb3c9c000, 0xfb65a000, 0xddb2d000, 0x78022800, 0x9c0b3c00, 0x5a0fb600, 0x2d0ddb00,
	 0xa2878080, 0xf3c9c040, 0xdb65a020, 0x6db2d0b0, 0x800228f8, 0x400b3cdc, 0x200fb67a, 0xb00ddb9d},
};

/* random_val.cuh:88-156 */
uint32_t orc_sobol(uint32_t index, uint32_t dim) {
	uint32_t X = 0;
	for (uint32_t bit = 0; bit < 32; bit++) {
		uint32_t mask = (index >> bit) & 1;
		X ^= mask * k_sobol_directions[dim][bit];
	}
	return X;
}

float orc_ld_random_val_export(uint32_t index, uint32_t seed, uint32_t dim) { return orc_ld_random_val(index, seed, dim); }
void orc_ld_random_pixel_offset_export(uint32_t spp, float* out) { orc_ld_random_pixel_offset(spp, out); }
uint32_t orc_morton3D_export(uint32_t x, uint32_t y, uint32_t z) { return orc_morton3D(x, y, z); }
uint32_t orc_morton3D_invert_export(uint32_t x) { return orc_morton3D_invert(x); }
uint16_t orc_f2h_export(float f) { return orc_f2h(f); }
float orc_h2f_export(uint16_t h) { return orc_h2f(h); }

/* draws `n` floats after advancing by `advance` from default_rng_t{seed} (random_val.cuh:28-35) */
void orc_pcg32_floats(uint64_t seed, int64_t advance, uint32_t n, float* out, uint64_t* state_out) {
	orc_pcg32 r = orc_pcg32_make(seed);
	if (advance) orc_pcg32_advance(&r, advance);
	for (uint32_t i = 0; i < n; ++i) out[i] = orc_pcg32_next_float(&r);
	if (state_out) { state_out[0] = r.state; state_out[1] = r.inc; }
}
void orc_pcg32_uints(uint64_t seed, int64_t advance, uint32_t n, uint32_t* out) {
	orc_pcg32 r = orc_pcg32_make(seed);
	if (advance) orc_pcg32_advance(&r, advance);
	for (uint32_t i = 0; i < n; ++i) out[i] = orc_pcg32_next_uint(&r);
}

/* ------------------------------------------------------------------ */
/* testbed_nerf.cu:369-416 mark_untrained_density_grid                 */
/* ------------------------------------------------------------------ */
void orc_mark_untrained_density_grid(uint32_t n_elements, float* grid_out, uint32_t n_training_images,
                                     const orc_image_meta* metadata, const orc_xform* xforms, int clear_visible_voxels) {
	#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n_elements; ++i) {
		uint32_t level = i / ORC_NERF_GRID_N_CELLS;
		uint32_t pos_idx = i % ORC_NERF_GRID_N_CELLS;
		uint32_t x = orc_morton3D_invert(pos_idx >> 0);
		uint32_t y = orc_morton3D_invert(pos_idx >> 1);
		uint32_t z = orc_morton3D_invert(pos_idx >> 2);
		float s = ldexpf(1.0f, (int)level);
		orc_vec3 pos = orc_v3(
			(((float)x + 0.5f) / (float)ORC_NERF_GRIDSIZE - 0.5f) * s + 0.5f,
			(((float)y + 0.5f) / (float)ORC_NERF_GRIDSIZE - 0.5f) * s + 0.5f,
			(((float)z + 0.5f) / (float)ORC_NERF_GRIDSIZE - 0.5f) * s + 0.5f);
		float voxel_radius = 0.5f * ORC_SQRT3 * s / (float)ORC_NERF_GRIDSIZE;
		int count = 0;
		for (uint32_t j = 0; j < n_training_images; ++j) {
			if (metadata[j].lens_mode == 2 || metadata[j].lens_mode == 3) { count++; break; }
			float half_resx = (float)metadata[j].res[0] * 0.5f;
			float half_resy = (float)metadata[j].res[1] * 0.5f;
			const float* xf = xforms[j].start;
			orc_vec3 ploc = orc_sub(pos, orc_col(xf, 3));
			float px = orc_dot(ploc, orc_col(xf, 0));
			float py = orc_dot(ploc, orc_col(xf, 1));
			float pz = orc_dot(ploc, orc_col(xf, 2));
			if (pz > 0.f) {
				if (fabsf(px) - voxel_radius < pz / metadata[j].focal_length[0] * half_resx &&
				    fabsf(py) - voxel_radius < pz / metadata[j].focal_length[1] * half_resy) {
					count++;
					if (count > 0) break;
				}
			}
		}
		if (clear_visible_voxels || (grid_out[i] < 0) != (count <= 0)) {
			grid_out[i] = (count > 0) ? 0.f : -1.f;
		}
	}
}

/* ------------------------------------------------------------------ */
/* testbed_nerf.cu:465-494 generate_grid_samples_nerf_nonuniform       */
/* rng passed by value; positions out are warped (NerfPosition = 3 floats) */
/* ------------------------------------------------------------------ */
void orc_generate_grid_samples_nonuniform(uint32_t n_elements, uint64_t rng_state, uint64_t rng_inc, uint32_t step,
                                          const orc_aabb* aabb, const float* grid_in, float* out_pos, uint32_t* indices,
                                          uint32_t n_cascades, float thresh) {
	#pragma omp parallel for schedule(static)
	for (uint32_t i = 0; i < n_elements; ++i) {
		orc_pcg32 rng = {rng_state, rng_inc};
		orc_pcg32_advance(&rng, (int64_t)((uint64_t)i * 4u)); /* i*4 is evaluated in uint32 in the reference; n<=2^23 so no wrap */
		uint32_t level = (uint32_t)(orc_pcg32_next_float(&rng) * (float)n_cascades) % n_cascades;
		uint32_t idx = 0;
		for (uint32_t j = 0; j < 10; ++j) {
			idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % ORC_NERF_GRID_N_CELLS;
			idx += level * ORC_NERF_GRID_N_CELLS;
			if (grid_in[idx] > thresh) break;
		}
		uint32_t pos_idx = idx % ORC_NERF_GRID_N_CELLS;
		uint32_t x = orc_morton3D_invert(pos_idx >> 0);
		uint32_t y = orc_morton3D_invert(pos_idx >> 1);
		uint32_t z = orc_morton3D_invert(pos_idx >> 2);
		float rx = orc_pcg32_next_float(&rng), ry = orc_pcg32_next_float(&rng), rz = orc_pcg32_next_float(&rng);
		float s = ldexpf(1.0f, (int)level);
		orc_vec3 pos = orc_v3(
			(((float)x + rx) / (float)ORC_NERF_GRIDSIZE - 0.5f) * s + 0.5f,
			(((float)y + ry) / (float)ORC_NERF_GRIDSIZE - 0.5f) * s + 0.5f,
			(((float)z + rz) / (float)ORC_NERF_GRIDSIZE - 0.5f) * s + 0.5f);
		orc_vec3 w = orc_aabb_relative_pos(aabb, pos);
		out_pos[3 * i + 0] = w.x; out_pos[3 * i + 1] = w.y; out_pos[3 * i + 2] = w.z;
		indices[i] = idx;
	}
}

/* testbed_nerf.cu:496-512 splat (atomicMax on uint view of non-negative floats) */
void orc_splat_grid_samples_max(uint32_t n_elements, const uint32_t* indices, const uint16_t* network_output /* fp16, channel 0 row */,
                                float* grid_out, int density_activation) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		uint32_t local_idx = indices[i];
		float mlp = orc_network_to_density(orc_h2f(network_output[i]), density_activation);
		float optical_thickness = mlp * ldexpf(ORC_MIN_CONE_STEPSIZE, 0);
		uint32_t u = orc_f2u(optical_thickness);
		uint32_t cur = orc_f2u(grid_out[local_idx]);
		if (u > cur) grid_out[local_idx] = optical_thickness;
	}
}

/* testbed_nerf.cu:532-555 */
void orc_ema_grid_samples(uint32_t n_elements, float decay, float* grid_out, const float* grid_in) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		float importance = grid_in[i];
		float prev_val = grid_out[i];
		float val = (prev_val < 0.f) ? prev_val : fmaxf(prev_val * decay, importance);
		grid_out[i] = val;
	}
}

/* testbed_nerf.cu:2851-2852: mean of max(v,0)/n over cascade 0.  The reduction order of tcnn::reduce_sum is
 * unspecified; restated with a double accumulator (tests use a tolerance on the mean, exactness on the bitfield
 * given an identical mean input). */
float orc_density_grid_mean(const float* grid, uint32_t n_elements) {
	double acc = 0.0;
	for (uint32_t i = 0; i < n_elements; ++i) acc += (double)(fmaxf(grid[i], 0.f) / (float)n_elements);
	return (float)acc;
}

/* testbed_nerf.cu:563-587 */
void orc_grid_to_bitfield(uint32_t n_elements, uint32_t n_nonzero_elements, const float* grid, uint8_t* bitfield, float mean_density) {
	float thresh = fminf(ORC_NERF_MIN_OPTICAL_THICKNESS, mean_density);
	for (uint32_t i = 0; i < n_elements; ++i) {
		if (i >= n_nonzero_elements) { bitfield[i] = 0; continue; }
		uint8_t bits = 0;
		for (uint8_t j = 0; j < 8; ++j) bits |= grid[i * 8 + j] > thresh ? ((uint8_t)1 << j) : 0;
		bitfield[i] = bits;
	}
}

/* testbed_nerf.cu:589-610 */
void orc_bitfield_max_pool(uint32_t n_elements, const uint8_t* prev_level, uint8_t* next_level) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		uint8_t bits = 0;
		for (uint8_t j = 0; j < 8; ++j) bits |= prev_level[i * 8 + j] > 0 ? ((uint8_t)1 << j) : 0;
		uint32_t x = orc_morton3D_invert(i >> 0) + ORC_NERF_GRIDSIZE / 8;
		uint32_t y = orc_morton3D_invert(i >> 1) + ORC_NERF_GRIDSIZE / 8;
		uint32_t z = orc_morton3D_invert(i >> 2) + ORC_NERF_GRIDSIZE / 8;
		next_level[orc_morton3D(x, y, z)] |= bits;
	}
}

/* testbed_nerf.cu:2844-2859 update_density_grid_mean_and_bitfield (given the mean) */
void orc_update_bitfield(const float* grid, uint32_t n_cascades_used, float mean_density, uint8_t* bitfield /* 8 * G/8 bytes */) {
	const uint32_t n = ORC_NERF_GRID_N_CELLS;
	orc_grid_to_bitfield(n / 8 * ORC_NERF_CASCADES, n / 8 * n_cascades_used, grid, bitfield, mean_density);
	for (uint32_t level = 1; level < ORC_NERF_CASCADES; ++level) {
		orc_bitfield_max_pool(n / 64, bitfield + orc_grid_mip_offset(level - 1) / 8, bitfield + orc_grid_mip_offset(level) / 8);
	}
}

/* ------------------------------------------------------------------ */
/* camera helpers                                                       */
/* ------------------------------------------------------------------ */

/* common_device.cuh:80-111 read_image<2,float> */
void orc_read_image2(const float* data, const int32_t res[2], const float pos[2], float out[2]) {
	float pfx = pos[0] * (float)(res[0] - 1), pfy = pos[1] * (float)(res[1] - 1);
	int tx = (int)pfx, ty = (int)pfy;
	float wx = pfx - (float)tx, wy = pfy - (float)ty;
	float acc[2] = {0, 0};
	const int dx[4] = {0, 1, 0, 1}, dy[4] = {0, 0, 1, 1};
	const float w[4] = {(1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy};
	/* [eigen] the four scaled vectors are summed left to right */
	for (int k = 0; k < 4; ++k) {
		int px = tx + dx[k], py = ty + dy[k];
		px = px < res[0] - 1 ? px : res[0] - 1; px = px > 0 ? px : 0;
		py = py < res[1] - 1 ? py : res[1] - 1; py = py > 0 ? py : 0;
		const float* v = &data[(px + py * res[0]) * 2];
		if (k == 0) { acc[0] = w[k] * v[0]; acc[1] = w[k] * v[1]; }
		else { acc[0] = acc[0] + w[k] * v[0]; acc[1] = acc[1] + w[k] * v[1]; }
	}
	out[0] = acc[0]; out[1] = acc[1];
}

/* common_device.cuh:145-161 */
static void orc_apply_opencv_lens_distortion(const float* p, float u, float v, float* du, float* dv) {
	const float k1 = p[0], k2 = p[1], p1 = p[2], p2 = p[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	*du = u * radial + 2.0f * p1 * uv + p2 * (r2 + 2.0f * u2);
	*dv = v * radial + 2.0f * p2 * uv + p1 * (r2 + 2.0f * v2);
}
/* common_device.cuh:163-200; [eigen] Matrix2f::inverse() = adjugate * (1/det) */
void orc_iterative_opencv_lens_undistortion(const float* params, float* u, float* v) {
	const float kMaxStepNorm = 1e-10f, kRelStepSize = 1e-6f, eps = 1.1920928955078125e-07f;
	const float x0[2] = {*u, *v};
	float x[2] = {*u, *v};
	for (uint32_t i = 0; i < 100; ++i) {
		const float step0 = fmaxf(eps, fabsf(kRelStepSize * x[0]));
		const float step1 = fmaxf(eps, fabsf(kRelStepSize * x[1]));
		float dx[2], dx0b[2], dx0f[2], dx1b[2], dx1f[2];
		orc_apply_opencv_lens_distortion(params, x[0], x[1], &dx[0], &dx[1]);
		orc_apply_opencv_lens_distortion(params, x[0] - step0, x[1], &dx0b[0], &dx0b[1]);
		orc_apply_opencv_lens_distortion(params, x[0] + step0, x[1], &dx0f[0], &dx0f[1]);
		orc_apply_opencv_lens_distortion(params, x[0], x[1] - step1, &dx1b[0], &dx1b[1]);
		orc_apply_opencv_lens_distortion(params, x[0], x[1] + step1, &dx1f[0], &dx1f[1]);
		float J00 = 1 + (dx0f[0] - dx0b[0]) / (2 * step0);
		float J01 = (dx1f[0] - dx1b[0]) / (2 * step1);
		float J10 = (dx0f[1] - dx0b[1]) / (2 * step0);
		float J11 = 1 + (dx1f[1] - dx1b[1]) / (2 * step1);
		float det = J00 * J11 - J10 * J01;
		float invdet = 1.0f / det;
		float i00 = J11 * invdet, i01 = -J01 * invdet, i10 = -J10 * invdet, i11 = J00 * invdet;
		float r0 = x[0] + dx[0] - x0[0], r1 = x[1] + dx[1] - x0[1];
		float s0 = i00 * r0 + i01 * r1, s1 = i10 * r0 + i11 * r1;
		x[0] -= s0; x[1] -= s1;
		if (s0 * s0 + s1 * s1 < kMaxStepNorm) break;
	}
	*u = x[0]; *v = x[1];
}

/* common_device.cuh:236-249 */
orc_vec3 orc_f_theta_undistortion(float uvx, float uvy, const float* params, orc_vec3 error_direction) {
	float xpix = uvx * params[5], ypix = uvy * params[6];
	float norm = sqrtf(xpix * xpix + ypix * ypix);
	float alpha = params[0] + norm * (params[1] + norm * (params[2] + norm * (params[3] + norm * params[4])));
	float sin_alpha = sinf(alpha), cos_alpha = cosf(alpha);
	if (cos_alpha <= 1.17549435e-38f || norm == 0.f) return error_direction;
	sin_alpha *= 1.f / norm;
	return orc_v3(sin_alpha * xpix, sin_alpha * ypix, cos_alpha);
}
/* common_device.cuh:251-258 */
orc_vec3 orc_latlong_to_dir(float u, float v) {
	const float PI = 3.14159265358979323846f;
	float theta = (v - 0.5f) * PI, phi = (u - 0.5f) * PI * 2.0f;
	float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
	return orc_v3(sp * ct, st, cp * ct);
}

/* common_device.cuh:223-234 get_xform_given_rolling_shutter.
 * [eigen] Quaternionf(Matrix3f) (Shepperd), slerp (threshold 1-eps), normalized(), toRotationMatrix(). */
static void orc_quat_from_mat(const float* m /* 3x4 col-major */, float q[4] /* x y z w */) {
#define M(r, c) m[(c) * 3 + (r)]
	float t = M(0, 0) + M(1, 1) + M(2, 2);
	if (t > 0.0f) {
		t = sqrtf(t + 1.0f);
		q[3] = 0.5f * t;
		t = 0.5f / t;
		q[0] = (M(2, 1) - M(1, 2)) * t;
		q[1] = (M(0, 2) - M(2, 0)) * t;
		q[2] = (M(1, 0) - M(0, 1)) * t;
	} else {
		int i = 0;
		if (M(1, 1) > M(0, 0)) i = 1;
		if (M(2, 2) > M(i, i)) i = 2;
		int j = (i + 1) % 3, k = (j + 1) % 3;
		t = sqrtf(M(i, i) - M(j, j) - M(k, k) + 1.0f);
		q[i] = 0.5f * t;
		t = 0.5f / t;
		q[3] = (M(k, j) - M(j, k)) * t;
		q[j] = (M(j, i) + M(i, j)) * t;
		q[k] = (M(k, i) + M(i, k)) * t;
	}
#undef M
}
void orc_get_xform_given_rolling_shutter(const orc_xform* xf, const float rs[4], float u, float v, float motionblur_time, float out[12]) {
	float pixel_t = rs[0] + rs[1] * u + rs[2] * v + rs[3] * motionblur_time;
	orc_vec3 s3 = orc_col(xf->start, 3), e3 = orc_col(xf->end, 3);
	orc_vec3 pos = orc_add(s3, orc_scale(orc_sub(e3, s3), pixel_t));
	float qa[4], qb[4], q[4];
	orc_quat_from_mat(xf->start, qa);
	orc_quat_from_mat(xf->end, qb);
	/* slerp: dot over coeffs (x,y,z,w) */
	const float one = 1.0f - 1.1920928955078125e-07f;
	float d = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3];
	float absD = fabsf(d);
	float scale0, scale1;
	if (absD >= one) { scale0 = 1.0f - pixel_t; scale1 = pixel_t; }
	else {
		float theta = acosf(absD), sinTheta = sinf(theta);
		scale0 = sinf((1.0f - pixel_t) * theta) / sinTheta;
		scale1 = sinf(pixel_t * theta) / sinTheta;
	}
	if (d < 0.0f) scale1 = -scale1;
	for (int i = 0; i < 4; ++i) q[i] = scale0 * qa[i] + scale1 * qb[i];
	/* normalized(): coeffs / norm */
	float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
	float n = sqrtf(n2);
	for (int i = 0; i < 4; ++i) q[i] = q[i] / n;
	const float x = q[0], y = q[1], z = q[2], w = q[3];
	const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
	const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
	out[0] = 1.0f - (tyy + tzz); out[3] = txy - twz;          out[6] = txz + twy;
	out[1] = txy + twz;          out[4] = 1.0f - (txx + tzz); out[7] = tyz - twx;
	out[2] = txz - twy;          out[5] = tyz + twx;          out[8] = 1.0f - (txx + tyy);
	out[9] = pos.x; out[10] = pos.y; out[11] = pos.z;
}

/* common_device.cuh:677-709 read_rgba */
void orc_read_rgba(const float xy[2], const int32_t res[2], const void* pixels, int type, float out[4]) {
	int px = (int)(xy[0] * (float)res[0]), py = (int)(xy[1] * (float)res[1]);
	px = px < res[0] - 1 ? px : res[0] - 1; px = px > 0 ? px : 0;
	py = py < res[1] - 1 ? py : res[1] - 1; py = py > 0 ? py : 0;
	uint64_t idx = (uint64_t)px + (uint64_t)py * (uint64_t)res[0];
	switch (type) {
		case 1: {
			uint32_t raw = ((const uint32_t*)pixels)[idx];
			if (raw == 0x00FF00FFu) { out[0] = out[1] = out[2] = out[3] = -1.0f; return; }
			uint8_t val[4]; memcpy(val, &raw, 4);
			float alpha = (float)val[3] * (1.0f / 255.0f);
			out[0] = orc_srgb_to_linear((float)val[0] * (1.0f / 255.0f)) * alpha;
			out[1] = orc_srgb_to_linear((float)val[1] * (1.0f / 255.0f)) * alpha;
			out[2] = orc_srgb_to_linear((float)val[2] * (1.0f / 255.0f)) * alpha;
			out[3] = alpha;
			return;
		}
		case 2: {
			const uint16_t* h = (const uint16_t*)pixels + idx * 4;
			for (int i = 0; i < 4; ++i) out[i] = orc_h2f(h[i]);
			return;
		}
		case 3: {
			const float* f = (const float*)pixels + idx * 4;
			for (int i = 0; i < 4; ++i) out[i] = f[i];
			return;
		}
		default: out[0] = 5.0f; out[1] = 0.0f; out[2] = 0.0f; out[3] = 1.0f; return;
	}
}

/* common.h:201-224 */
uint32_t orc_binary_search(float val, const float* data, uint32_t length) {
	if (length == 0) return 0;
	uint32_t first = 0, count = length;
	while (count > 0) {
		uint32_t step = count / 2, it = first + step;
		if (data[it] < val) { first = it + 1; count -= step + 1; }
		else count = step;
	}
	return first < length - 1 ? first : length - 1;
}

/* testbed_nerf.cu:991-1022 sample_cdf_2d (UNIFORM_SAMPLING_FRACTION = 0.5) */
static void orc_sample_cdf_2d(float xy[2], uint32_t img, const int32_t res[2], const float* cdf_x_cond_y, const float* cdf_y, float* pdf) {
	const float UNIFORM = 0.5f;
	if (xy[0] < UNIFORM) { xy[0] /= UNIFORM; return; }   /* the reference leaves *pdf untouched (1.0) on this branch */
	xy[0] = (xy[0] - UNIFORM) / (1.0f - UNIFORM);
	cdf_y += (size_t)img * res[1];
	uint32_t y = orc_binary_search(xy[1], cdf_y, (uint32_t)res[1]);
	float prev = y > 0 ? cdf_y[y - 1] : 0.0f;
	float pmf_y = cdf_y[y] - prev;
	xy[1] = (xy[1] - prev) / pmf_y;
	cdf_x_cond_y += (size_t)img * res[1] * res[0] + (size_t)y * res[0];
	uint32_t x = orc_binary_search(xy[0], cdf_x_cond_y, (uint32_t)res[0]);
	prev = x > 0 ? cdf_x_cond_y[x - 1] : 0.0f;
	float pmf_x = cdf_x_cond_y[x] - prev;
	xy[0] = (xy[0] - prev) / pmf_x;
	if (pdf) *pdf = pmf_x * pmf_y * (float)(res[0] * res[1]);
	xy[0] = ((float)x + xy[0]) / (float)res[0];
	xy[1] = ((float)y + xy[1]) / (float)res[1];
}

/* testbed_nerf.cu:1047-1060; cdf may be NULL (error-map sampling is default-off, testbed.h:668-669) */
void orc_nerf_random_image_pos_training(orc_pcg32* rng, const int32_t res[2], int snap_to_pixel_centers, const orc_error_map_cdf* cdf, uint32_t img, float xy[2], float* pdf) {
	xy[0] = orc_pcg32_next_float(rng);
	xy[1] = orc_pcg32_next_float(rng);
	if (pdf) *pdf = 1.0f;
	if (cdf && cdf->cdf_x_cond_y) orc_sample_cdf_2d(xy, img, cdf->res, cdf->cdf_x_cond_y, cdf->cdf_y, pdf);
	if (snap_to_pixel_centers) {
		for (int k = 0; k < 2; ++k) {
			int p = (int)(xy[k] * (float)res[k]);
			p = p > 0 ? p : 0; p = p < res[k] - 1 ? p : res[k] - 1;
			xy[k] = ((float)p + 0.5f) / (float)res[k];
		}
	}
}

/* testbed_nerf.cu:1062-1083 */
uint32_t orc_image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_training_images, const float* cdf, float* pdf) {
	if (cdf) {
		float sample = orc_ld_random_val(base_idx, 0xdeadbeefu, 0);
		uint32_t img = orc_binary_search(sample, cdf, n_training_images);
		if (pdf) { float prev = img > 0 ? cdf[img - 1] : 0.0f; *pdf = (cdf[img] - prev) * (float)n_training_images; }
		return img;
	}
	if (pdf) *pdf = 1.0f;
	return ((base_idx * n_training_images) / n_rays) % n_training_images;
}

/* testbed_nerf.cu:1982-2037 construct_cdf_2d / construct_cdf_1d (MIN_PDF = 0.01; __frcp_rn = correctly rounded 1/x) */
void orc_construct_cdf_2d(uint32_t n_images, uint32_t height, uint32_t width, const float* data, float* cdf_x_cond_y, float* cdf_y) {
	const float MIN_PDF = 0.01f;
	for (uint32_t img = 0; img < n_images; ++img) for (uint32_t y = 0; y < height; ++y) {
		const size_t off = ((size_t)img * height + y) * width;
		float cum = 0;
		for (uint32_t x = 0; x < width; ++x) { cum += data[off + x] + 1e-10f; cdf_x_cond_y[off + x] = cum; }
		cdf_y[(size_t)img * height + y] = cum;
		const float norm = 1.0f / cum;
		for (uint32_t x = 0; x < width; ++x) cdf_x_cond_y[off + x] = (1.0f - MIN_PDF) * cdf_x_cond_y[off + x] * norm + MIN_PDF * (float)(x + 1) / (float)width;
	}
}
void orc_construct_cdf_1d(uint32_t n_images, uint32_t height, float* cdf_y, float* cdf_img) {
	const float MIN_PDF = 0.01f;
	for (uint32_t img = 0; img < n_images; ++img) {
		float* c = cdf_y + (size_t)img * height;
		float cum = 0;
		for (uint32_t y = 0; y < height; ++y) { cum += c[y]; c[y] = cum; }
		cdf_img[img] = cum;
		const float norm = 1.0f / cum;
		for (uint32_t y = 0; y < height; ++y) c[y] = (1.0f - MIN_PDF) * c[y] * norm + MIN_PDF * (float)(y + 1) / (float)height;
	}
}
/* testbed_nerf.cu:3000-3015: host-side image CDF from the per-image sums (MIN_PMF = 0.1) */
void orc_image_cdf_host(uint32_t n_images, const float* pmf_unnormalized, float* pmf_out, float* cdf_out) {
	float cum = 0;
	for (uint32_t i = 0; i < n_images; ++i) { cum += pmf_unnormalized[i]; cdf_out[i] = cum; }
	const float norm = 1.0f / cum, MIN_PMF = 0.1f;
	for (uint32_t i = 0; i < n_images; ++i) {
		pmf_out[i] = (1.0f - MIN_PMF) * pmf_unnormalized[i] * norm + MIN_PMF / (float)n_images;
		cdf_out[i] = (1.0f - MIN_PMF) * cdf_out[i] * norm + MIN_PMF * (float)(i + 1) / (float)n_images;
	}
}

/* ------------------------------------------------------------------ */
/* testbed_nerf.cu:1085-1260 generate_training_samples_nerf            */
/* Rays are processed in index order, so slot reservation (the two     */
/* atomicAdd's at 1225/1232) is deterministic here; the GPU order is    */
/* not, so tests compare per ray, keyed by ray_indices_out.            */
/* `ray_offset`/`n_rays_global` are the data-parallel extension         */
/* (SURVEY §8e): thread i works on global ray ray_offset+i of           */
/* n_rays_global; (0, n_rays) reproduces the reference exactly.         */
/* ------------------------------------------------------------------ */
void orc_generate_training_samples(
	uint32_t n_rays, const orc_aabb* aabb, uint32_t max_samples, uint64_t rng_state, uint64_t rng_inc,
	uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, orc_ray* rays_out_unnormalized,
	uint32_t* numsteps_out, orc_coord* coords_out, uint32_t n_training_images, const orc_image_meta* metadata,
	const orc_xform* xforms, const uint8_t* density_grid, int max_level_rand_training, float* max_level_ptr,
	int snap_to_pixel_centers, int train_envmap, float cone_angle_constant, const float* distortion_data,
	const int32_t distortion_resolution[2], uint32_t ray_offset, uint32_t n_rays_global, const orc_error_map_cdf* cdf) {
	for (uint32_t li = 0; li < n_rays; ++li) {
		const uint32_t i = li + ray_offset;
		uint32_t img = orc_image_idx(i, n_rays_global, n_training_images, cdf ? cdf->cdf_img : NULL, NULL);
		const orc_image_meta* md = &metadata[img];
		orc_pcg32 rng = {rng_state, rng_inc};
		orc_pcg32_advance(&rng, (int64_t)((uint64_t)(uint32_t)(i * ORC_N_MAX_RANDOM_SAMPLES_PER_RAY)));
		float xy[2];
		orc_nerf_random_image_pos_training(&rng, md->res, snap_to_pixel_centers, cdf, img, xy, NULL);

		float texel[4];
		orc_read_rgba(xy, md->res, md->pixels, md->image_data_type, texel);
		if (texel[0] < 0.0f) continue;

		float max_level = max_level_rand_training ? (orc_pcg32_next_float(&rng) * 2.0f) : 1.0f;
		float motionblur_time = orc_pcg32_next_float(&rng);

		float xform[12];
		orc_get_xform_given_rolling_shutter(&xforms[img], md->rolling_shutter, xy[0], xy[1], motionblur_time, xform);

		orc_ray ray_unnormalized;
		if (md->rays) {
			int px = (int)(xy[0] * (float)md->res[0]), py = (int)(xy[1] * (float)md->res[1]);
			px = px < md->res[0] - 1 ? px : md->res[0] - 1; px = px > 0 ? px : 0;
			py = py < md->res[1] - 1 ? py : md->res[1] - 1; py = py > 0 ? py : 0;
			ray_unnormalized = md->rays[(uint64_t)px + (uint64_t)py * (uint64_t)md->res[0]];
		} else {
			ray_unnormalized.o = orc_col(xform, 3);
			orc_vec3 d;
			if (md->lens_mode == 2) {
				d = orc_f_theta_undistortion(xy[0] - md->principal_point[0], xy[1] - md->principal_point[1], md->lens_params, orc_v3(0.f, 0.f, 1.f));
			} else if (md->lens_mode == 3) {
				d = orc_latlong_to_dir(xy[0], xy[1]);
			} else {
				d = orc_v3(
					(xy[0] - md->principal_point[0]) * (float)md->res[0] / md->focal_length[0],
					(xy[1] - md->principal_point[1]) * (float)md->res[1] / md->focal_length[1],
					1.0f);
				if (md->lens_mode == 1) orc_iterative_opencv_lens_undistortion(md->lens_params, &d.x, &d.y);
			}
			if (distortion_data) {
				float off[2];
				orc_read_image2(distortion_data, distortion_resolution, xy, off);
				d.x += off[0]; d.y += off[1];
			}
			ray_unnormalized.d = orc_mat3_mul(xform, d); /* NOT normalized */
		}

		orc_vec3 ray_d_normalized = orc_normalized(ray_unnormalized.d);
		float tminmax[2];
		orc_aabb_ray_intersect(aabb, ray_unnormalized.o, ray_d_normalized, tminmax);
		float cone_angle = cone_angle_constant; /* calc_cone_angle: testbed_nerf.cu:87-94 */
		tminmax[0] = fmaxf(tminmax[0], 0.0f);

		float startt = tminmax[0];
		startt += orc_calc_dt(startt, cone_angle) * orc_pcg32_next_float(&rng);
		orc_vec3 idir = orc_v3(1.0f / ray_d_normalized.x, 1.0f / ray_d_normalized.y, 1.0f / ray_d_normalized.z);

		uint32_t j = 0;
		float t = startt;
		orc_vec3 pos;
		while (orc_aabb_contains(aabb, pos = orc_add(ray_unnormalized.o, orc_scale(ray_d_normalized, t))) && j < ORC_NERF_STEPS) {
			float dt = orc_calc_dt(t, cone_angle);
			uint32_t mip = (uint32_t)orc_mip_from_dt(dt, pos, ORC_NERF_CASCADES - 1);
			if (orc_density_grid_occupied_at(pos, density_grid, mip)) {
				++j;
				t += dt;
			} else {
				uint32_t res = ORC_NERF_GRIDSIZE >> mip;
				t = orc_advance_to_next_voxel(t, cone_angle, pos, ray_d_normalized, idir, res);
			}
		}
		if (j == 0 && !train_envmap) continue;

		uint32_t numsteps = j;
		uint32_t base = *numsteps_counter; *numsteps_counter += numsteps;
		if (base + numsteps > max_samples) continue;

		orc_coord* co = coords_out + base;
		uint32_t ray_idx = (*ray_counter)++;
		ray_indices_out[ray_idx] = i; /* global ray index (== thread id when ray_offset == 0) */
		rays_out_unnormalized[ray_idx] = ray_unnormalized;
		numsteps_out[ray_idx * 2 + 0] = numsteps;
		numsteps_out[ray_idx * 2 + 1] = base;

		orc_vec3 warped_dir = orc_warp_direction(ray_d_normalized);
		t = startt;
		j = 0;
		while (orc_aabb_contains(aabb, pos = orc_add(ray_unnormalized.o, orc_scale(ray_d_normalized, t))) && j < numsteps) {
			float dt = orc_calc_dt(t, cone_angle);
			uint32_t mip = (uint32_t)orc_mip_from_dt(dt, pos, ORC_NERF_CASCADES - 1);
			if (orc_density_grid_occupied_at(pos, density_grid, mip)) {
				orc_vec3 wp = orc_aabb_relative_pos(aabb, pos);
				co[j].pos[0] = wp.x; co[j].pos[1] = wp.y; co[j].pos[2] = wp.z;
				co[j].dt = orc_warp_dt(dt);
				co[j].dir[0] = warped_dir.x; co[j].dir[1] = warped_dir.y; co[j].dir[2] = warped_dir.z;
				++j;
				t += dt;
			} else {
				uint32_t res = ORC_NERF_GRIDSIZE >> mip;
				t = orc_advance_to_next_voxel(t, cone_angle, pos, ray_d_normalized, idir, res);
			}
		}
		if (max_level_rand_training) {
			for (j = 0; j < numsteps; ++j) max_level_ptr[base + j] = max_level;
		}
	}
}
