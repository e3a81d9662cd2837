/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orc_core.h header).
 * Single-NeRF renderer (Testbed::NerfTracer) + frame accumulation / tonemap.
 * Follows src/testbed_nerf.cu:612-664 (advance_pos_nerf), 705-765 (generate_next_nerf_network_inputs),
 * 767-989 (composite_kernel_nerf, Shade mode, no masks/glow), 1748-1781 (shade), 1784-1807 (compact),
 * 1809-1978 (init rays, Perspective camera), 2140-2267 (trace loop), common_device.cuh:260-317 (pixel_to_ray),
 * src/render_buffer.cu:235-272 (accumulate), 274-348 + 540-567 (tonemap).
 */
#include "ngp_oracle.h"
#include <stdlib.h>

/* random_val.cuh:109-125 */
static void orc_square2disk_shirley(float a, float b, float* ox, float* oy) {
	const float PI = 3.14159265358979323846f;
	float phi, r;
	if (a * a > b * b) { r = a; phi = (PI / 4.0f) * (b / a); }
	else { r = b; phi = (PI / 2.0f) - (PI / 4.0f) * (a / b); }
	*ox = r * cosf(phi); *oy = r * sinf(phi);
}

/* common_device.cuh:260-317 pixel_to_ray (no distortion grid) */
static orc_ray orc_pixel_to_ray(uint32_t spp, int px, int py, const int32_t res[2], const float focal_length[2], const float* cam /* 3x4 */,
                                const float screen_center[2], const float parallax_shift[3], int snap_to_pixel_centers, float near_distance,
                                int lens_mode, const float* lens_params, float focus_z, float aperture_size, const float* distortion_grid, const int32_t* distortion_res) {
	float offset[2];
	orc_ld_random_pixel_offset(snap_to_pixel_centers ? 0 : spp, offset);
	float u = ((float)px + offset[0]) / (float)res[0];
	float v = ((float)py + offset[1]) / (float)res[1];
	orc_vec3 dir;
	if (lens_mode == 2) {          /* FTheta (:281-285) */
		dir = orc_f_theta_undistortion(u - screen_center[0], v - screen_center[1], lens_params, orc_v3(1000.f, 0.f, 0.f));
		if (dir.x == 1000.f) { orc_ray out = {orc_v3(1000.f, 0.f, 0.f), orc_v3(0.f, 0.f, 1.f)}; return out; }   /* a point outside the aabb: pixel not rendered */
	} else if (lens_mode == 3) {   /* LatLong (:286-287) */
		dir = orc_latlong_to_dir(u, v);
	} else {
		dir = orc_v3(
			(u - screen_center[0]) * (float)res[0] / focal_length[0],
			(v - screen_center[1]) * (float)res[1] / focal_length[1],
			1.0f);
		if (lens_mode == 1) orc_iterative_opencv_lens_undistortion(lens_params, &dir.x, &dir.y);
	}
	if (distortion_grid) {   /* :297-299 */
		float uv[2] = {u, v}, off[2];
		orc_read_image2(distortion_grid, distortion_res, uv, off);
		dir.x += off[0]; dir.y += off[1];
	}
	orc_vec3 head_pos = orc_v3(parallax_shift[0], parallax_shift[1], 0.f);
	dir = orc_sub(dir, orc_scale(head_pos, parallax_shift[2]));
	dir = orc_mat3_mul(cam, dir);
	orc_vec3 origin = orc_add(orc_mat3_mul(cam, head_pos), orc_col(cam, 3));
	if (aperture_size > 0.0f) {   /* depth of field (:307-312): jitter the origin on the lens disk, keep the point at focus_z fixed */
		orc_vec3 lookat = orc_add(origin, orc_scale(dir, focus_z));
		float rv[2], bx, by;
		orc_ld_random_val_2d(spp, (uint32_t)px * 19349663u + (uint32_t)py * 96925573u, rv);
		orc_square2disk_shirley(rv[0] * 2.0f - 1.0f, rv[1] * 2.0f - 1.0f, &bx, &by);
		bx *= aperture_size; by *= aperture_size;
		origin = orc_add(origin, orc_v3(cam[0] * bx + cam[3] * by, cam[1] * bx + cam[4] * by, cam[2] * bx + cam[5] * by));
		dir = orc_v3((lookat.x - origin.x) / focus_z, (lookat.y - origin.y) / focus_z, (lookat.z - origin.z) / focus_z);
	}
	origin = orc_add(origin, orc_scale(dir, near_distance));
	orc_ray r = {origin, dir};
	return r;
}

/* envmap.cuh:29-63 read_envmap (fp32 map), random_val.cuh:64-69 dir_to_spherical_unorm */
void orc_read_envmap(const float* data, const int32_t res[2], const float dir_in[3], float out[4]) {
	const float PI = 3.14159265358979323846f;
	const float dx = dir_in[2], dy = -dir_in[0], dz = dir_in[1];
	const float cos_theta = fminf(fmaxf(dz, -1.0f), 1.0f);
	const float theta = acosf(cos_theta);
	const float phi = atan2f(dy, dx);
	const float cyl[2] = {theta / PI, phi / (2.0f * PI) + 0.5f};
	const float fx = cyl[1] * (float)(res[0] - 1), fy = cyl[0] * (float)(res[1] - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	for (int c = 0; c < 4; ++c) out[c] = 0.0f;
	for (int k = 0; k < 4; ++k) {
		int x = tx + (k & 1), y = ty + (k >> 1);
		if (x < 0) x += res[0]; else if (x >= res[0]) x -= res[0];
		y = y < res[1] - 1 ? y : res[1] - 1; y = y > 0 ? y : 0;
		const float w = ((k & 1) ? wx : 1 - wx) * ((k >> 1) ? wy : 1 - wy);
		const float* t = data + ((size_t)x + (size_t)y * res[0]) * 4;
		for (int c = 0; c < 4; ++c) out[c] = k == 0 ? w * t[c] : out[c] + w * t[c];
	}
}
/* common_device.cuh:594-619 */
static void orc_hsv_to_rgb(float h, float s, float v, float out[3]) {
	if (s == 0.0f) { out[0] = out[1] = out[2] = v; return; }
	h = fmodf(h, 1.0f) * 6.0f;
	int i = (int)h;
	float f = h - (float)i;
	float p = v * (1.0f - s), q = v * (1.0f - s * f), t = v * (1.0f - s * (1.0f - f));
	switch (i) {
		case 0: out[0] = v; out[1] = t; out[2] = p; break;
		case 1: out[0] = q; out[1] = v; out[2] = p; break;
		case 2: out[0] = p; out[1] = v; out[2] = t; break;
		case 3: out[0] = p; out[1] = q; out[2] = v; break;
		case 4: out[0] = t; out[1] = p; out[2] = v; break;
		default: out[0] = v; out[1] = p; out[2] = q; break;
	}
}
/* common_device.cuh:541-560 */
static void orc_apply_quilting(uint32_t* x, uint32_t* y, const int32_t res[2], float parallax_shift[3], const int32_t qd[2]) {
	float resx = (float)res[0] / (float)qd[0], resy = (float)res[1] / (float)qd[1];
	int panelx = (int)floorf((float)*x / resx), panely = (int)floorf((float)*y / resy);
	*x = (uint32_t)((float)*x - (float)panelx * resx);
	*y = (uint32_t)((float)*y - (float)panely * resy);
	int idx = panelx + qd[0] * panely;
	if (qd[0] == 2 && qd[1] == 1) {
		parallax_shift[0] = idx ? (-0.5f * parallax_shift[0]) : (0.5f * parallax_shift[0]);
	} else {
		const float max_parallax_angle = 17.5f;
		float parallax_angle = max_parallax_angle * 3.14159265358979323846f / 180.f * (((float)idx + 0.5f) * 2.f / (float)(qd[1] * qd[0]) - 1.f);
		parallax_shift[0] = atanf(parallax_angle) / parallax_shift[2];
	}
}

void orc_init_rays(uint32_t sample_index, orc_payload* payloads, const int32_t res[2], const float focal_length[2],
                   const float* camera_matrix0, const float* camera_matrix1, const float rolling_shutter[4], const float screen_center[2],
                   const float parallax_shift[3], int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local /* 3x3 */,
                   float near_distance, int lens_mode, const float* lens_params, float* depthbuffer, float plane_z, float aperture_size,
                   const orc_render_camera* camera_models) {
	orc_init_rays_ex(sample_index, payloads, res, focal_length, camera_matrix0, camera_matrix1, rolling_shutter, screen_center, parallax_shift, snap_to_pixel_centers, render_aabb,
	                 render_aabb_to_local, near_distance, lens_mode, lens_params, depthbuffer, plane_z, aperture_size, camera_models, NULL);
}

/* testbed_nerf.cu:1809-1978 init_rays_with_payload_kernel_nerf: camera models, quilting, envmap background, crop masks, Distortion mode;
 * render_aabb_to_local = identity-or-given 3x3 (column-major). */
void orc_init_rays_ex(uint32_t sample_index, orc_payload* payloads, const int32_t res[2], const float focal_length[2],
                      const float* camera_matrix0, const float* camera_matrix1, const float rolling_shutter[4], const float screen_center[2],
                      const float parallax_shift_in[3], int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local /* 3x3 */,
                      float near_distance, int lens_mode, const float* lens_params, float* depthbuffer, float plane_z, float aperture_size,
                      const orc_render_camera* camera_models /* NULL or model 0: Perspective; only model / sq_* / qh_* are read (1868-1908) */,
                      const orc_render_extras* ex) {
	if (plane_z < 0) aperture_size = 0.0f;   /* :1849-1851 */
	const int32_t one[2] = {1, 1};
	const int32_t* qd = ex && ex->quilting_dims[0] > 0 && ex->quilting_dims[1] > 0 ? ex->quilting_dims : one;
	const float* distortion = ex && ex->distortion && ex->distortion_res[0] > 0 ? ex->distortion : NULL;
	const float* envmap = ex && ex->envmap && ex->envmap_res[0] > 0 ? ex->envmap : NULL;
	/* a frame rendered in row shards (include/ngp_hip.h NgpRenderExtras.row_begin / row_end): only the rows of the range are set up; payload slot = pixel - first pixel of
	 * the range, payload.idx and every random-number key = the pixel's index in the whole frame */
	const int row_begin = ex && (ex->row_begin || ex->row_end) ? ex->row_begin : 0, row_end = ex && (ex->row_begin || ex->row_end) ? ex->row_end : res[1];
	for (int yy = row_begin; yy < row_end; ++yy) for (int xx = 0; xx < res[0]; ++xx) {
		uint32_t x = (uint32_t)xx, y = (uint32_t)yy;
		uint32_t idx = x + (uint32_t)res[0] * y;
		const uint32_t slot = idx - (uint32_t)res[0] * (uint32_t)row_begin;
		float parallax_shift[3] = {parallax_shift_in[0], parallax_shift_in[1], parallax_shift_in[2]};
		if (qd[0] != 1 || qd[1] != 1) orc_apply_quilting(&x, &y, res, parallax_shift, qd);
		float u = ((float)x + 0.5f) * (1.f / (float)res[0]);
		float v = ((float)y + 0.5f) * (1.f / (float)res[1]);
		float ray_time = rolling_shutter[0] + rolling_shutter[1] * u + rolling_shutter[2] * v + rolling_shutter[3] * orc_ld_random_val(sample_index, idx * 72239731u, 0);
		float cam[12];
		for (int k = 0; k < 12; ++k) cam[k] = camera_matrix0[k] * ray_time + camera_matrix1[k] * (1.f - ray_time);
		const int32_t qres[2] = {res[0] / qd[0], res[1] / qd[1]};   /* :1863 */
		orc_ray ray;
		if (camera_models && camera_models->model != 0) {
			orc_extra_camera_model_pixel_to_ray(camera_models->model, sample_index, x, y, (float)qres[0], (float)qres[1], cam, camera_models->sq_width, camera_models->sq_height,
			                                    camera_models->sq_curvature, camera_models->qh_front, camera_models->qh_back, near_distance, plane_z, aperture_size, &ray.o, &ray.d);
		} else {
			ray = orc_pixel_to_ray(sample_index, (int)x, (int)y, qres, focal_length, cam, screen_center, parallax_shift, snap_to_pixel_centers, near_distance, lens_mode, lens_params, plane_z, aperture_size,
			                       distortion, ex ? ex->distortion_res : NULL);
		}

		orc_payload* p = &payloads[slot];
		p->max_weight = 0.0f;
		if (plane_z < 0) {   /* slice plane (:1913-1923): the ray stops at depth -plane_z along the view axis */
			float n = orc_norm(ray.d);
			p->origin = ray.o;
			p->dir = orc_scale(ray.d, 1.0f / n);
			p->t = -plane_z * n;
			p->idx = idx;
			p->n_steps = 0;
			p->alive = 0;
			depthbuffer[idx] = -plane_z;
			continue;
		}
		depthbuffer[idx] = 1e10f;
		ray.d = orc_normalized(ray.d);
		if (envmap) {   /* :1931-1933 */
			const float d3[3] = {ray.d.x, ray.d.y, ray.d.z};
			orc_read_envmap(envmap, ex->envmap_res, d3, ex->frame_buffer + 4 * (size_t)idx);
		}
		orc_vec3 lo = orc_mat3_mul(render_aabb_to_local, ray.o), ld = orc_mat3_mul(render_aabb_to_local, ray.d);
		float tmm[2];
		orc_aabb_ray_intersect(render_aabb, lo, ld, tmm);
		float t = fmaxf(tmm[0], 0.0f) + 1e-6f;
		if (!orc_aabb_contains(render_aabb, orc_mat3_mul(render_aabb_to_local, orc_add(ray.o, orc_scale(ray.d, t))))) {
			p->origin = ray.o;
			p->alive = 0;
			continue;
		}
		int ray_intersects_any_mask = !ex || ex->n_render_masks == 0;   /* :1943-1956 */
		if (ex) for (uint32_t k = 0; k < ex->n_render_masks && !ray_intersects_any_mask; ++k) {
			const float ro[3] = {ray.o.x, ray.o.y, ray.o.z}, rd[3] = {ray.d.x, ray.d.y, ray.d.z};
			ray_intersects_any_mask = orc_mask_intersects_ray(&ex->render_masks[k], ro, rd);
		}
		if (!ray_intersects_any_mask) {
			p->origin = ray.o;
			p->alive = 0;
			continue;
		}
		if (ex && ex->render_mode == 5) {   /* Distortion (:1959-1970) */
			float off[2] = {0.0f, 0.0f};
			if (distortion) {
				const float uv[2] = {((float)x + 0.5f) / (float)res[0], ((float)y + 0.5f) / (float)res[1]};
				orc_read_image2(distortion, ex->distortion_res, uv, off);
			}
			float* fb = ex->frame_buffer + 4 * (size_t)idx;
			const float ox = off[0] * 50.0f, oy = off[1] * 50.0f;
			orc_hsv_to_rgb(atan2f(oy, ox) / (2.0f * 3.14159265358979323846f) + 0.5f, 1.0f, sqrtf(ox * ox + oy * oy), fb);
			fb[3] = 1.0f;
			depthbuffer[idx] = 1.0f;
			p->origin = orc_add(ray.o, orc_scale(ray.d, 10000.0f));
			p->alive = 0;
			continue;
		}
		p->origin = ray.o;
		p->dir = ray.d;
		p->t = t;
		p->idx = idx;
		p->n_steps = 0;
		p->alive = 1;
	}
}

/* testbed_nerf.cu:612-664 */
void orc_advance_pos(uint32_t n_elements, const orc_aabb* render_aabb, const float* render_aabb_to_local, uint32_t sample_index,
                     orc_payload* payloads, const uint8_t* density_grid, uint32_t min_mip, float cone_angle_constant) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		orc_payload* p = &payloads[i];
		if (!p->alive) continue;
		orc_vec3 origin = p->origin, dir = p->dir;
		orc_vec3 idir = orc_v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
		float cone_angle = cone_angle_constant;
		float t = p->t;
		float dt = orc_calc_dt(t, cone_angle);
		t += orc_ld_random_val(sample_index, p->idx * 786433u, 0) * dt;   /* the reference keys by the ray index i = the pixel index of a frame traced at once (:625) */
		orc_vec3 pos;
		while (1) {
			pos = orc_add(origin, orc_scale(dir, t));
			if (!orc_aabb_contains(render_aabb, orc_mat3_mul(render_aabb_to_local, pos))) { p->alive = 0; break; }
			dt = orc_calc_dt(t, cone_angle);
			uint32_t mip = (uint32_t)orc_mip_from_dt(dt, pos, ORC_NERF_CASCADES - 1);
			if (mip < min_mip) mip = min_mip;
			if (!density_grid || orc_density_grid_occupied_at(pos, density_grid, mip)) break;
			uint32_t res = ORC_NERF_GRIDSIZE >> mip;
			t = orc_advance_to_next_voxel(t, cone_angle, pos, dir, idir, res);
		}
		p->t = t;
	}
}

/* testbed_nerf.cu:1784-1807 (sequential => deterministic order) */
void orc_compact_rays(uint32_t n_elements, const float* src_rgba, const float* src_depth, const orc_payload* src_payloads,
                      float* dst_rgba, float* dst_depth, orc_payload* dst_payloads,
                      float* dst_final_rgba, float* dst_final_depth, orc_payload* dst_final_payloads, uint32_t* counter, uint32_t* final_counter) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		if (src_payloads[i].alive) {
			uint32_t idx = (*counter)++;
			dst_payloads[idx] = src_payloads[i];
			memcpy(dst_rgba + 4 * idx, src_rgba + 4 * i, 16);
			dst_depth[idx] = src_depth[i];
		} else if (src_rgba[4 * i + 3] > 0.001f) {
			uint32_t idx = (*final_counter)++;
			dst_final_payloads[idx] = src_payloads[i];
			memcpy(dst_final_rgba + 4 * idx, src_rgba + 4 * i, 16);
			dst_final_depth[idx] = src_depth[i];
		}
	}
}

/* testbed_nerf.cu:705-765; network_input is SoA-strided: sample j of ray i at [i + j*n_elements] */
void orc_generate_next_inputs(uint32_t n_elements, const orc_aabb* render_aabb, const orc_aabb* train_aabb, orc_payload* payloads,
                              orc_coord* network_input, uint32_t n_steps, const uint8_t* density_grid, uint32_t min_mip, float cone_angle_constant) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		orc_payload* p = &payloads[i];
		if (!p->alive) continue;
		orc_vec3 origin = p->origin, dir = p->dir;
		orc_vec3 idir = orc_v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
		float cone_angle = cone_angle_constant;
		float t = p->t;
		uint32_t j;
		int done = 0;
		for (j = 0; j < n_steps; ++j) {
			orc_vec3 pos;
			float dt = 0.0f;
			while (1) {
				pos = orc_add(origin, orc_scale(dir, t));
				if (!orc_aabb_contains(render_aabb, pos)) { p->n_steps = (uint16_t)j; done = 1; break; }
				dt = orc_calc_dt(t, cone_angle);
				uint32_t mip = (uint32_t)orc_mip_from_dt(dt, pos, ORC_NERF_CASCADES - 1);
				if (mip < min_mip) mip = min_mip;
				if (!density_grid || orc_density_grid_occupied_at(pos, density_grid, mip)) break;
				uint32_t res = ORC_NERF_GRIDSIZE >> mip;
				t = orc_advance_to_next_voxel(t, cone_angle, pos, dir, idir, res);
			}
			if (done) break;
			orc_coord* c = &network_input[i + (size_t)j * n_elements];
			orc_vec3 wp = orc_aabb_relative_pos(train_aabb, pos);
			orc_vec3 wd = orc_warp_direction(dir);
			c->pos[0] = wp.x; c->pos[1] = wp.y; c->pos[2] = wp.z;
			c->dt = orc_warp_dt(dt);
			c->dir[0] = wd.x; c->dir[1] = wd.y; c->dir[2] = wd.z;
			t += dt;
		}
		if (done) continue; /* payload.t is NOT written on the early return (745) */
		p->t = t;
		p->n_steps = (uint16_t)n_steps;
	}
}

/* testbed_nerf.cu:767-989, ERenderMode::Shade, no masks, no glow, show_accel < 0.
 * network_output: fp16 rgbsigma for sample j of ray i at (i + j*n_elements), `out_stride` halves apart, channel c at +c
 * (the reference reads a row-major [channel][sample] matrix, 816-819; same values, different addressing). */
void orc_composite(uint32_t n_elements, uint32_t current_step, const orc_aabb* aabb, const float* camera_matrix /* 3x4 */,
                   float* rgba, float* depth, orc_payload* payloads, const orc_coord* network_input, const uint16_t* network_output,
                   uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance) {
	orc_composite_mode(n_elements, current_step, aabb, camera_matrix, rgba, depth, payloads, network_input, network_output, out_stride, n_steps, rgb_activation, density_activation,
	                   min_transmittance, 1 /* Shade */, 1.0f, -1);
}

/* the visualisation modes that only need the sample itself (testbed_nerf.cu:938-968): ERenderMode AO 0, Shade 1, Positions 3 (with the
 * show_accel colouring), Depth 4; Cost 6 and Slice 7 composite like Shade */
void orc_composite_mode(uint32_t n_elements, uint32_t current_step, const orc_aabb* aabb, const float* camera_matrix /* 3x4 */,
                        float* rgba, float* depth, orc_payload* payloads, const orc_coord* network_input, const uint16_t* network_output,
                        uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance,
                        int render_mode, float depth_scale, int show_accel) {
	orc_composite_ex(n_elements, current_step, aabb, camera_matrix, rgba, depth, payloads, network_input, network_output, out_stride, n_steps, rgb_activation, density_activation,
	                 min_transmittance, render_mode, depth_scale, show_accel, NULL);
}

/* glow / grid-line visualisation (testbed_nerf.cu:843-939; the "#if 0" block there is dead) */
static void orc_glow_shading(int glow_mode, float glow_y_cutoff, orc_vec3 pos, orc_vec3 cam_pos, float rgb[3], float* weight) {
	float glow = 0.f;
	int green_grid = glow_mode & 1, green_cutline = glow_mode & 2, mask_to_alpha = glow_mode & 4, radial_mode = glow_mode & 8, grid_mode = glow_mode & 16;
	float dist;
	if (radial_mode) {
		dist = orc_norm(orc_sub(pos, cam_pos));
		dist = fminf(dist, (4.5f - pos.y) * 0.333f);
	} else {
		dist = pos.y;
	}
	if (grid_mode) {
		glow = 1.f / fmaxf(1.f, dist);
	} else {
		float y = glow_y_cutoff - dist;
		float mask = 0.f;
		if (y > 0.f) {
			y *= 80.f;
			mask = fminf(1.f, y);
			if (green_cutline) glow += fmaxf(0.f, 1.f - fabsf(1.f - y)) * 4.f;
			if (y > 1.f) y = 1.f - (y - 1.f) * 0.05f;
			if (green_grid) glow += fmaxf(0.f, y / fmaxf(1.f, dist));
		}
		if (mask_to_alpha) *weight *= mask;
	}
	if (glow > 0.f) {
		float line = 0.0f;
		for (int o = 0; o < 4; ++o) {
			const float f = (float)(2 << o);
			line += fmaxf(0.f, cosf(pos.y * f * 3.141592653589793f * 16.f) - 0.975f);
			line += fmaxf(0.f, cosf(pos.x * f * 3.141592653589793f * 16.f) - 0.975f);
			line += fmaxf(0.f, cosf(pos.z * f * 3.141592653589793f * 16.f) - 0.975f);
		}
		if (grid_mode) {
			glow = glow * line * 15.f;
			rgb[1] = glow; rgb[2] = glow * 0.5f; rgb[0] = glow * 0.25f;
		} else {
			glow = glow * glow * 0.25f + glow * line * 15.f;
			rgb[1] += glow; rgb[2] += glow * 0.5f; rgb[0] += glow * 0.25f;
		}
	}
}

/* composite_kernel_nerf in full (testbed_nerf.cu:767-989): crop masks, glow, every render mode.  Normals 2: network_input.pos holds
 * d(density output)/d(pos) (input_gradient wrote it there, :2225-2226); EncodingVis 8: it holds the visualised activation (:2227-2228). */
void orc_composite_ex(uint32_t n_elements, uint32_t current_step, const orc_aabb* aabb, const float* camera_matrix /* 3x4 */,
                      float* rgba, float* depth, orc_payload* payloads, const orc_coord* network_input, const uint16_t* network_output,
                      uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance,
                      int render_mode, float depth_scale, int show_accel, const orc_render_extras* ex) {
	orc_vec3 cam_fwd = orc_col(camera_matrix, 2);
	orc_vec3 cam_pos = orc_col(camera_matrix, 3);
	for (uint32_t i = 0; i < n_elements; ++i) {
		orc_payload* p = &payloads[i];
		if (!p->alive) continue;
		float local_rgba[4]; memcpy(local_rgba, rgba + 4 * i, 16);
		float local_depth = depth[i];
		uint32_t actual_n_steps = p->n_steps;
		uint32_t j = 0;
		for (; j < actual_n_steps; ++j) {
			const uint16_t* lo = network_output + ((size_t)i + (size_t)j * n_elements) * out_stride;
			const orc_coord* in = &network_input[i + (size_t)j * n_elements];
			orc_vec3 pos = orc_unwarp_position(orc_v3(in->pos[0], in->pos[1], in->pos[2]), aabb);
			float T = 1.f - local_rgba[3];
			float dt = orc_unwarp_dt(in->dt);
			float alpha = 1.f - expf(-orc_network_to_density(orc_h2f(lo[3]), density_activation) * dt);
			if (show_accel >= 0) alpha = 1.f;   /* :827-829 */
			float weight = alpha * T;
			float rgb[3];
			for (int c = 0; c < 3; ++c) rgb[c] = orc_network_to_rgb(orc_h2f(lo[c]), rgb_activation);
			if (ex && ex->n_render_masks) {   /* :833-840 */
				float mask_weight = 1.f;
				const float p3[3] = {pos.x, pos.y, pos.z};
				for (uint32_t k = 0; k < ex->n_render_masks; ++k) mask_weight = orc_clampf(mask_weight + orc_mask_sample(&ex->render_masks[k], p3), 0.0f, 1.0f);
				weight *= mask_weight;
			}
			if (ex && ex->glow_mode) orc_glow_shading(ex->glow_mode, ex->glow_y_cutoff, pos, cam_pos, rgb, &weight);
			if (render_mode == 2) {            /* Normals (:941-946) */
				float k = -orc_network_to_density_derivative(orc_h2f(lo[3]), density_activation);
				orc_vec3 nrm = orc_normalized(orc_v3(k * in->pos[0], k * in->pos[1], k * in->pos[2]));
				rgb[0] = nrm.x; rgb[1] = nrm.y; rgb[2] = nrm.z;
			} else if (render_mode == 8) {     /* EncodingVis (:961-962) */
				rgb[0] = in->pos[0]; rgb[1] = in->pos[1]; rgb[2] = in->pos[2];
			} else if (render_mode == 3) {     /* Positions */
				if (show_accel >= 0) {
					int mp = orc_mip_from_pos(pos, 7);
					uint32_t mip = (uint32_t)(show_accel > mp ? show_accel : mp);
					uint32_t res = 128u >> mip;
					int ix = (int)(pos.x * (float)res), iy = (int)(pos.y * (float)res), iz = (int)(pos.z * (float)res);
					orc_pcg32 rng = orc_pcg32_make((uint64_t)(int64_t)(ix + iy * 232323 + iz * 727272));   /* default_rng_t(int) */
					rgb[0] = 1.f - (float)mip * (1.f / 7.f);
					rgb[1] = orc_pcg32_next_float(&rng);
					rgb[2] = orc_pcg32_next_float(&rng);
				} else {
					rgb[0] = (pos.x - 0.5f) / 2.0f + 0.5f; rgb[1] = (pos.y - 0.5f) / 2.0f + 0.5f; rgb[2] = (pos.z - 0.5f) / 2.0f + 0.5f;
				}
			} else if (render_mode == 4) {     /* Depth */
				rgb[0] = rgb[1] = rgb[2] = orc_dot(cam_fwd, orc_sub(pos, p->origin)) * depth_scale;
			} else if (render_mode == 0) {     /* AO */
				rgb[0] = rgb[1] = rgb[2] = alpha;
			}
			for (int c = 0; c < 3; ++c) local_rgba[c] += rgb[c] * weight;
			local_rgba[3] += weight;
			if (weight > p->max_weight) {
				p->max_weight = weight;
				local_depth = orc_dot(cam_fwd, orc_sub(pos, cam_pos));
			}
			if (local_rgba[3] > (1.0f - min_transmittance)) {
				float w = local_rgba[3];
				for (int c = 0; c < 4; ++c) local_rgba[c] /= w;
				break;
			}
		}
		if (j < n_steps) {
			p->alive = 0;
			p->n_steps = (uint16_t)(j + current_step);
		}
		memcpy(rgba + 4 * i, local_rgba, 16);
		depth[i] = local_depth;
	}
}

/* testbed_nerf.cu:1748-1781 Shade mode */
void orc_shade(uint32_t n_elements, const float* rgba, const float* depth, const orc_payload* payloads, int train_in_linear_colors,
               float* frame_buffer, float* depth_buffer) {
	orc_shade_mode(n_elements, rgba, depth, payloads, train_in_linear_colors, frame_buffer, depth_buffer, 1 /* Shade */);
}
/* shade_kernel_nerf with render_mode (1748-1781): Cost shows n_steps / 128; only Shade / Slice colours are sRGB-decoded */
void orc_shade_mode(uint32_t n_elements, const float* rgba, const float* depth, const orc_payload* payloads, int train_in_linear_colors,
                    float* frame_buffer, float* depth_buffer, int render_mode) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		float tmp[4]; memcpy(tmp, rgba + 4 * i, 16);
		if (render_mode == 2) {   /* Normals (:1764-1767) */
			orc_vec3 n = orc_normalized(orc_v3(tmp[0], tmp[1], tmp[2]));
			tmp[0] = (0.5f * n.x + 0.5f) * tmp[3]; tmp[1] = (0.5f * n.y + 0.5f) * tmp[3]; tmp[2] = (0.5f * n.z + 0.5f) * tmp[3];
		} else if (render_mode == 6) { float col = (float)payloads[i].n_steps / 128; tmp[0] = tmp[1] = tmp[2] = col; tmp[3] = 1.0f; }
		if (!train_in_linear_colors && (render_mode == 1 || render_mode == 7)) for (int c = 0; c < 3; ++c) tmp[c] = orc_srgb_to_linear(tmp[c]);
		float* fb = frame_buffer + 4 * (size_t)payloads[i].idx;
		for (int c = 0; c < 4; ++c) fb[c] = tmp[c] + fb[c] * (1.0f - tmp[3]);
		if (render_mode != 7 && tmp[3] > 0.2f) depth_buffer[payloads[i].idx] = depth[i];
	}
}

/* Slice mode (testbed_nerf.cu:2445-2476): generate_nerf_network_inputs_at_current_position (:676-682), compute_nerf_rgba (:684-703) */
void orc_generate_inputs_at_current_position(uint32_t n_elements, const orc_aabb* aabb, const orc_payload* payloads, orc_coord* network_input) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		orc_vec3 dir = payloads[i].dir;
		orc_vec3 wp = orc_aabb_relative_pos(aabb, orc_add(payloads[i].origin, orc_scale(dir, payloads[i].t))), wd = orc_warp_direction(dir);
		orc_coord* c = &network_input[i];
		c->pos[0] = wp.x; c->pos[1] = wp.y; c->pos[2] = wp.z; c->dt = orc_warp_dt(ORC_MIN_CONE_STEPSIZE); c->dir[0] = wd.x; c->dir[1] = wd.y; c->dir[2] = wd.z;
	}
}
void orc_compute_nerf_rgba(uint32_t n_elements, const uint16_t* network_output, uint32_t out_stride, float* rgba, int rgb_activation, int density_activation, float depth, int density_as_alpha) {
	for (uint32_t i = 0; i < n_elements; ++i) {
		const uint16_t* lo = network_output + (size_t)i * out_stride;
		float density = orc_network_to_density(orc_h2f(lo[3]), density_activation);
		float alpha = 1.f, w;
		if (density_as_alpha) w = density;
		else w = alpha = orc_clampf(1.f - expf(-density * depth), 0.0f, 1.0f);
		for (int c = 0; c < 3; ++c) rgba[4 * (size_t)i + c] = orc_network_to_rgb(orc_h2f(lo[c]), rgb_activation) * alpha;
		rgba[4 * (size_t)i + 3] = w;
	}
}

/* render_buffer.cu:235-272 (Linear / SRGB colour spaces) */
void orc_accumulate(const int32_t res[2], const float* frame_buffer, float* accumulate_buffer, float sample_count, int color_space_srgb) {
	size_t n = (size_t)res[0] * res[1];
	for (size_t idx = 0; idx < n; ++idx) {
		float color[4]; memcpy(color, frame_buffer + 4 * idx, 16);
		float* tmp = accumulate_buffer + 4 * idx;
		if (color_space_srgb) for (int c = 0; c < 3; ++c) color[c] = orc_linear_to_srgb(color[c]);
		for (int c = 0; c < 3; ++c) tmp[c] = (tmp[c] * sample_count + color[c]) / (sample_count + 1);
		tmp[3] = (tmp[3] * sample_count + color[3]) / (sample_count + 1);
	}
}

/* render_buffer.cu:274-348: ETonemapCurve {Identity, ACES, Hable, Reinhard} (common.h:93-98) */
static void orc_tonemap_curve(float x[3], int curve) {
	if (curve == 0) return;
	for (int c = 0; c < 3; ++c) x[c] = fmaxf(x[c], 0.f);
	float k0, k1, k2, k3, k4, k5;
	if (curve == 1) {
		k0 = 0.6f * 0.6f * 2.51f; k1 = 0.6f * 0.03f; k2 = 0.0f; k3 = 0.6f * 0.6f * 2.43f; k4 = 0.6f * 0.59f; k5 = 0.14f;
	} else if (curve == 2) {
		const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
		k0 = A * F - A * E; k1 = C * B * F - B * E; k2 = 0.0f; k3 = A * F; k4 = B * F; k5 = D * F * F;
		const float W = 11.2f;
		const float nom = k0 * (W * W) + k1 * W + k2;
		const float denom = k3 * (W * W) + k4 * W + k5;
		const float white_scale = denom / nom;
		k0 = 4.0f * k0 * white_scale; k1 = 2.0f * k1 * white_scale; k2 = k2 * white_scale; k3 = 4.0f * k3; k4 = 2.0f * k4;
	} else {
		float Y = 0.2126f * x[0] + 0.7152f * x[1] + 0.0722f * x[2];
		for (int c = 0; c < 3; ++c) x[c] = x[c] * (1.f / (Y + 1.0f));
		return;
	}
	for (int c = 0; c < 3; ++c) {
		float sq = x[c] * x[c];
		float nom = sq * k0 + k1 * x[c] + k2;
		float denom = k3 * sq + k4 * x[c] + k5;
		x[c] = nom / denom;
	}
}

/* render_buffer.cu:327-348, 540-567 */
void orc_tonemap(const int32_t res[2], float exposure, const float background_color_in[4], const float* accumulate_buffer,
                 int color_space_srgb, int output_color_space_srgb, int tonemap_curve, int clamp_output_color, float* surface) {
	size_t n = (size_t)res[0] * res[1];
	float bg[4]; memcpy(bg, background_color_in, 16);
	if (!color_space_srgb) for (int c = 0; c < 3; ++c) bg[c] = orc_srgb_to_linear(bg[c]);
	for (size_t idx = 0; idx < n; ++idx) {
		float color[4]; memcpy(color, accumulate_buffer + 4 * idx, 16);
		float weight = (1 - color[3]) * bg[3];
		for (int c = 0; c < 3; ++c) color[c] += bg[c] * weight;
		color[3] += weight;
		if (color_space_srgb) for (int c = 0; c < 3; ++c) color[c] = orc_srgb_to_linear(color[c]);
		float e = powf(2.0f, exposure);
		for (int c = 0; c < 3; ++c) color[c] *= e;
		orc_tonemap_curve(color, tonemap_curve);
		if (output_color_space_srgb) for (int c = 0; c < 3; ++c) color[c] = orc_linear_to_srgb(color[c]);
		if (clamp_output_color) for (int c = 0; c < 4; ++c) color[c] = fminf(fmaxf(color[c], 0.0f), 1.0f);
		memcpy(surface + 4 * idx, color, 16);
	}
}

/* testbed_nerf.cu:2354-2500 render_nerf (Shade) = init_rays_from_camera (2047-2138) + trace (2140-2267) + shade (2478).
 * Writes into frame_buffer/depth_buffer (cleared by the caller as render_frame does, testbed.cu:2698).
 * `inference_params` = the EMA copy.  Returns the number of network samples evaluated. */
uint64_t orc_render_nerf(const orc_net* net, const uint16_t* inference_params, uint32_t sample_index, const int32_t res[2],
                         const float focal_length[2], const float* camera_matrix0, const float* camera_matrix1, const float screen_center[2],
                         int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local, const orc_aabb* train_aabb,
                         float near_distance, const uint8_t* density_grid, float cone_angle_constant, int rgb_activation, int density_activation,
                         float min_transmittance, int train_in_linear_colors, float* frame_buffer, float* depth_buffer) {
	return orc_render_nerf_rows(net, inference_params, sample_index, res, focal_length, camera_matrix0, camera_matrix1, screen_center, snap_to_pixel_centers, render_aabb, render_aabb_to_local,
	                            train_aabb, near_distance, density_grid, cone_angle_constant, rgb_activation, density_activation, min_transmittance, train_in_linear_colors, frame_buffer, depth_buffer,
	                            0, res[1]);
}

/* The same frame, rows [row_begin, row_end) only (SURVEY.md §8e "Render: image tiles/rows per rank + gather"): frame_buffer / depth_buffer are the WHOLE frame's, the
 * pixels of the other rows are not touched.  Rays do not interact and every random number is keyed by the pixel's index in the whole frame, so the rows written
 * are, bit for bit, those of the frame rendered at once — whatever the partition (the n_steps-per-pass schedule differs, the per-ray sample sequence does not). */
uint64_t orc_render_nerf_rows(const orc_net* net, const uint16_t* inference_params, uint32_t sample_index, const int32_t res[2],
                              const float focal_length[2], const float* camera_matrix0, const float* camera_matrix1, const float screen_center[2],
                              int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local, const orc_aabb* train_aabb,
                              float near_distance, const uint8_t* density_grid, float cone_angle_constant, int rgb_activation, int density_activation,
                              float min_transmittance, int train_in_linear_colors, float* frame_buffer, float* depth_buffer, int row_begin, int row_end) {
	if (row_end <= row_begin) return 0;
	const uint32_t n_pixels = (uint32_t)res[0] * (uint32_t)(row_end - row_begin);
	orc_render_extras rows_ex;
	memset(&rows_ex, 0, sizeof(rows_ex));
	rows_ex.quilting_dims[0] = rows_ex.quilting_dims[1] = 1; rows_ex.render_mode = 1; rows_ex.row_begin = row_begin; rows_ex.row_end = row_end;
	const float zero4[4] = {0, 0, 0, 0}, zero3[3] = {0, 0, 0};
	orc_payload* payload[2]; float* rgba[2]; float* depth[2];
	for (int b = 0; b < 2; ++b) {
		payload[b] = (orc_payload*)calloc(n_pixels, sizeof(orc_payload));
		rgba[b] = (float*)calloc((size_t)n_pixels * 4, 4);
		depth[b] = (float*)calloc(n_pixels, 4);
	}
	orc_payload* hit_payload = (orc_payload*)calloc(n_pixels, sizeof(orc_payload));
	float* hit_rgba = (float*)calloc((size_t)n_pixels * 4, 4);
	float* hit_depth = (float*)calloc(n_pixels, 4);
	orc_coord* net_in = (orc_coord*)calloc((size_t)n_pixels * 8, sizeof(orc_coord));
	uint16_t* net_out = (uint16_t*)calloc((size_t)n_pixels * 8 * 4, 2);

	orc_init_rays_ex(sample_index, payload[0], res, focal_length, camera_matrix0, camera_matrix1, zero4, screen_center, zero3,
	                 snap_to_pixel_centers, render_aabb, render_aabb_to_local, near_distance, 0, NULL, depth_buffer, 1.0f, 0.0f, NULL, &rows_ex);
	orc_advance_pos(n_pixels, render_aabb, render_aabb_to_local, sample_index, payload[0], density_grid, 0, cone_angle_constant);

	uint32_t n_alive = n_pixels, n_hit = 0, i = 1, dbi = 0;
	uint64_t n_samples = 0;
	while (i < 10000) {
		int cur = (dbi + 1) % 2, tmp = dbi % 2;
		++dbi;
		uint32_t alive = 0;
		orc_compact_rays(n_alive, rgba[tmp], depth[tmp], payload[tmp], rgba[cur], depth[cur], payload[cur], hit_rgba, hit_depth, hit_payload, &alive, &n_hit);
		n_alive = alive;
		if (n_alive == 0) break;
		uint32_t n_steps = n_pixels / n_alive; n_steps = n_steps < 1 ? 1 : (n_steps > 8 ? 8 : n_steps);
		orc_generate_next_inputs(n_alive, render_aabb, train_aabb, payload[cur], net_in, n_steps, density_grid, 0, cone_angle_constant);
		/* only slots of alive rays with j < payload.n_steps are consumed by the compositor */
		for (uint32_t j = 0; j < n_steps; ++j) for (uint32_t r = 0; r < n_alive; ++r) {
			if (j < payload[cur][r].n_steps) {
				size_t s = r + (size_t)j * n_alive;
				orc_nerf_inference(net, inference_params, (const float*)&net_in[s], 7, 1, net_out + s * 4, 4);
				++n_samples;
			}
		}
		orc_composite(n_alive, i, train_aabb, camera_matrix1, rgba[cur], depth[cur], payload[cur], net_in, net_out, 4, n_steps, rgb_activation, density_activation, min_transmittance);
		i += n_steps;
	}
	orc_shade(n_hit, hit_rgba, hit_depth, hit_payload, train_in_linear_colors, frame_buffer, depth_buffer);
	for (int b = 0; b < 2; ++b) { free(payload[b]); free(rgba[b]); free(depth[b]); }
	free(hit_payload); free(hit_rgba); free(hit_depth); free(net_in); free(net_out);
	return n_samples;
}
