/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (the reference ships no golden vectors for this path; see orc_core.h).
 *
 * orc_multi.c — CPU restatement of the Blender add-on's multi-NeRF renderer:
 *   src/nerf_renderer.cu:17-563 (kernels) and 565-791 (host loop), src/nerf_utils.cu (runtime-parameter marching helpers),
 *   include/neural-graphics-primitives/nerf/mask_3D.cuh (SDF masks), camera_models.cuh:80-241 (cameras), common.h:337-355 (DownsampleInfo).
 * Plain sequential C, one loop iteration per CUDA thread.  4x4 matrices are column-major like Eigen's default.
 */
#include "ngp_oracle.h"

#include <stdlib.h>

static orc_vec3 xform_point(const float* m, orc_vec3 p) {   /* (M * p.homogeneous()).head<3>() */
	return orc_v3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12], ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13], ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
static orc_vec3 xform_dir(const float* m, orc_vec3 d) {     /* M.topLeftCorner<3,3>() * d */
	return orc_v3((m[0] * d.x + m[4] * d.y) + m[8] * d.z, (m[1] * d.x + m[5] * d.y) + m[9] * d.z, (m[2] * d.x + m[6] * d.y) + m[10] * d.z);
}
static orc_vec3 ld3(const float* p) { return orc_v3(p[0], p[1], p[2]); }

/* ---- nerf_utils.cu ---- */
static float get_dt(float t, float cone_angle, float mn, float mx) { return orc_clampf(t * cone_angle, mn, mx); }                 /* :27-29 */
static int get_mip_from_dt(float dt, orc_vec3 pos, uint32_t grid_size, uint32_t max_cascade) {                                   /* :146-153 */
	int mip = orc_mip_from_pos(pos, max_cascade);
	dt *= (float)(2u * grid_size);
	if (dt < 1.f) return mip;
	int exponent;
	frexpf(dt, &exponent);
	int m = exponent > mip ? exponent : mip;
	return (int)max_cascade < m ? (int)max_cascade : m;
}
static uint32_t get_cascaded_grid_idx_at(orc_vec3 pos, uint32_t mip, uint32_t grid_size) {                                       /* :155-174 */
	float mip_scale = scalbnf(1.0f, -(int)mip);
	pos.x -= 0.5f; pos.y -= 0.5f; pos.z -= 0.5f;
	pos.x *= mip_scale; pos.y *= mip_scale; pos.z *= mip_scale;
	pos.x += 0.5f; pos.y += 0.5f; pos.z += 0.5f;
	int ix = (int)(pos.x * (float)grid_size), iy = (int)(pos.y * (float)grid_size), iz = (int)(pos.z * (float)grid_size);
	int hi = (int)grid_size - 1;
	return orc_morton3D((uint32_t)orc_clampi(ix, 0, hi), (uint32_t)orc_clampi(iy, 0, hi), (uint32_t)orc_clampi(iz, 0, hi));
}
static int get_is_occupied(orc_vec3 pos, const uint8_t* bitfield, uint32_t mip, uint32_t grid_size, uint32_t grid_volume) {      /* :176-179 */
	uint32_t idx = get_cascaded_grid_idx_at(pos, mip, grid_size);
	return bitfield[idx / 8 + (grid_volume * mip) / 8] & (1u << (idx % 8));
}
static float get_t_advanced_to_next_voxel(float t, float cone, orc_vec3 pos, orc_vec3 dir, orc_vec3 idir, uint32_t res, float mn, float mx) { /* :31-43 */
	float t_target = t + orc_distance_to_next_voxel(pos, dir, idir, res);
	do { t += get_dt(t, cone, mn, mx); } while (t < t_target);
	return t;
}
static float get_warped_dt(float dt, float mn, uint32_t cascades) {                                                              /* :128-131 */
	float max_stepsize = mn * (float)(1u << (cascades - 1));
	return (dt - mn) / (max_stepsize - mn);
}
static float get_unwarped_dt(float dt, float mn, uint32_t cascades) {                                                            /* :133-136 */
	float max_stepsize = mn * (float)(1u << (cascades - 1));
	return dt * (max_stepsize - mn) + mn;
}

/* ---- mask_3D.cuh ---- */
static float sdf_box(orc_vec3 p, orc_vec3 b) {                                                                                   /* :33-36 */
	orc_vec3 d = orc_v3(fabsf(p.x) - 0.5f * b.x, fabsf(p.y) - 0.5f * b.y, fabsf(p.z) - 0.5f * b.z);
	orc_vec3 dm = orc_v3(fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f), fmaxf(d.z, 0.0f));
	return orc_norm(dm) + fminf(fmaxf(d.x, fmaxf(d.y, d.z)), 0.0f);
}
static float sdf_cylinder(orc_vec3 p, float r, float h) {                                                                        /* :38-41 */
	float dx = fabsf(sqrtf(p.y * p.y + p.x * p.x)) - r, dy = fabsf(p.z) - 0.5f * h;
	float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
	return sqrtf(mx * mx + my * my) + fminf(fmaxf(dx, dy), 0.0f);
}
static float mask_signed_distance(const orc_mask3d* m, orc_vec3 p) {                                                             /* :159-181 */
	orc_vec3 pl = xform_point(m->itransform, p);
	float d = 0.0f;
	switch (m->shape) {
		case 0: d = sdf_box(pl, orc_v3(m->config[0], m->config[1], m->config[2])); break;
		case 1: d = sdf_cylinder(pl, m->config[0], m->config[1]); break;
		case 2: d = orc_norm(pl) - m->config[0]; break;
		default: d = -1.0f; break;
	}
	return d * (m->mode == 0 ? 1.0f : -1.0f);
}
float orc_mask_sample(const orc_mask3d* m, const float p[3]) {                                                                   /* :192-211 */
	float k = m->mode == 0 ? 1.0f : -1.0f;
	if (m->shape == 3) return k;
	float d = mask_signed_distance(m, ld3(p));
	float alpha;
	if (m->feather == 0.0f) alpha = d < 0.0f ? 1.0f : 0.0f;
	else alpha = orc_clampf(0.5f - d / m->feather, 0.0f, 1.0f);
	return m->opacity * alpha * k;
}
static int ray_intersects_box(orc_vec3 o, orc_vec3 d, orc_vec3 size) {                                                           /* :51-60 */
	orc_vec3 inv = orc_v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	orc_vec3 t0 = orc_v3((-0.5f * size.x - o.x) * inv.x, (-0.5f * size.y - o.y) * inv.y, (-0.5f * size.z - o.z) * inv.z);
	orc_vec3 t1 = orc_v3((0.5f * size.x - o.x) * inv.x, (0.5f * size.y - o.y) * inv.y, (0.5f * size.z - o.z) * inv.z);
	float tmin = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
	float tmax = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
	return tmin <= tmax;
}
static int ray_intersects_sphere(orc_vec3 o, orc_vec3 d, float radius) {                                                         /* :63-67 */
	float od = orc_dot(d, o);
	float a = od * od;
	float b = orc_dot(o, o) - radius * radius;
	return !((a - b) < 0.0f);
}
static int intersect_plane_ray(orc_vec3 o, orc_vec3 d, orc_vec3 n, orc_vec3 p, float* t) {                                       /* :77-84 */
	float denom = orc_dot(n, d);
	if (denom > 1e-6f) { *t = orc_dot(orc_sub(p, o), n) / denom; return *t >= 0.0f; }
	return 0;
}
static int ray_intersects_cylinder(orc_vec3 o, orc_vec3 d, float radius, float height) {                                         /* :86-125 */
	float a = d.x * d.x + d.y * d.y;
	float b = 2.0f * (d.x * o.x + d.y * o.y);
	float c = (o.x * o.x + o.y * o.y) - radius * radius;
	float disc = b * b - 4.0f * a * c;
	if (disc < 0.0f) return 0;
	float d_sqrt = sqrtf(disc), a2 = 2.0f * a, h_2 = 0.5f * height;
	if (a2 > 1e-6f) {
		float t0 = (-b - d_sqrt) / a2, t1 = (-b + d_sqrt) / a2;
		float z0 = o.z + t0 * d.z, z1 = o.z + t1 * d.z;
		if ((z0 >= -h_2 && z0 <= h_2) || (z1 >= -h_2 && z1 <= h_2)) return 1;
	}
	float t = 0.0f;
	if (intersect_plane_ray(o, d, orc_v3(0.0f, 0.0f, 1.0f), orc_v3(0.0f, 0.0f, h_2), &t)) {
		orc_vec3 p = orc_add(o, orc_scale(d, t));
		if (p.x * p.x + p.y * p.y <= radius * radius) return 1;
	}
	if (intersect_plane_ray(o, d, orc_v3(0.0f, 0.0f, -1.0f), orc_v3(0.0f, 0.0f, -h_2), &t)) {
		orc_vec3 p = orc_add(o, orc_scale(d, t));
		if (p.x * p.x + p.y * p.y <= radius * radius) return 1;
	}
	return 0;
}
int orc_mask_intersects_ray(const orc_mask3d* m, const float ro[3], const float rd[3]) {                                         /* :213-246 */
	if (m->mode == 1) return 1;
	if (m->shape == 3) return m->mode == 0;
	orc_vec3 ol = xform_point(m->itransform, ld3(ro));
	orc_vec3 dl = orc_normalized(xform_dir(m->itransform, ld3(rd)));
	float f = 0.5f * m->feather;
	switch (m->shape) {
		case 0: return ray_intersects_box(ol, dl, orc_v3(m->config[0] + f, m->config[1] + f, m->config[2] + f));
		case 1: return ray_intersects_cylinder(ol, dl, m->config[0] + f, m->config[1] + f);
		case 2: return ray_intersects_sphere(ol, dl, m->config[0] + f);
		default: return 1;
	}
}

/* ---- camera_models.cuh ---- */
static void square2disk_shirley(float a, float b, float* ox, float* oy) {                                                        /* random_val.cuh:109-125 */
	const float PI = 3.14159265358979323846f;
	float phi, r;
	if (a * a > b * b) { r = a; phi = (PI / 4.0f) * (b / a); }
	else { r = b; phi = (PI / 2.0f) - (PI / 4.0f) * (a / b); }
	*ox = r * cosf(phi); *oy = r * sinf(phi);
}
static void apply_aperture(uint32_t spp, uint32_t px, uint32_t py, const float* cam, float aperture_size, float focus_z, orc_vec3* origin, orc_vec3* dir) {
	if (aperture_size > 0.0f) {
		orc_vec3 lookat = orc_add(*origin, orc_scale(*dir, focus_z));
		float rv[2];
		orc_ld_random_val_2d(spp, px * 19349663u + py * 96925573u, rv);
		float bx, by;
		square2disk_shirley(rv[0] * 2.0f - 1.0f, rv[1] * 2.0f - 1.0f, &bx, &by);
		bx *= aperture_size; by *= aperture_size;
		*origin = orc_add(*origin, orc_v3(cam[0] * bx + cam[3] * by, cam[1] * bx + cam[4] * by, cam[2] * bx + cam[5] * by));
		*dir = orc_v3((lookat.x - origin->x) / focus_z, (lookat.y - origin->y) / focus_z, (lookat.z - origin->z) / focus_z);
	}
}
static orc_vec3 lerp3(const float* a, const float* b, float t) { return orc_v3(a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), a[2] + t * (b[2] - a[2])); }

void orc_downsample_info_from_mip(const int32_t resolution[2], uint32_t mip, orc_downsample_info* ds) {                          /* common.h:337-355 */
	ds->max_pixels = (uint32_t)(resolution[0] * resolution[1]);
	ds->max_res[0] = resolution[0]; ds->max_res[1] = resolution[1];
	if (mip == 0) {
		ds->scaled_pixels = ds->max_pixels;
		ds->scaled_res[0] = resolution[0]; ds->scaled_res[1] = resolution[1];
		ds->skip[0] = ds->skip[1] = 1;
	} else {
		ds->skip[0] = ds->skip[1] = 1 << mip;
		ds->scaled_res[0] = (resolution[0] + ds->skip[0] - 1) / ds->skip[0];
		ds->scaled_res[1] = (resolution[1] + ds->skip[1] - 1) / ds->skip[1];
		ds->scaled_pixels = (uint32_t)(ds->scaled_res[0] * ds->scaled_res[1]);
	}
}

/* camera_models.cuh: the two extra camera models (model 2 SphericalQuadrilateral, 1 QuadrilateralHexahedron); c = 3x4 column-major camera matrix */
void orc_extra_camera_model_pixel_to_ray(int model, uint32_t spp, uint32_t x, uint32_t y, float rx, float ry, const float* c, float sq_width, float sq_height, float sq_curvature,
                                         const float* qh_front, const float* qh_back, float near_distance, float focus_z, float aperture_size, orc_vec3* origin, orc_vec3* dir) {
	if (model == 2) {     /* spherical_quadrilateral_pixel_to_ray, :162-203 */
			const float PI = 3.14159265358979323846f;
			float max_linear_len = sqrtf(sq_width * sq_width + sq_height * sq_height);
			float ux = 2.0f * (((float)x + 0.5f) / rx - 0.5f), uy = 2.0f * (((float)y + 0.5f) / ry - 0.5f);
			float qx = sq_width * ux, qy = sq_height * uy;
			float a = atan2f(qy, qx), r = sqrtf(qx * qx + qy * qy);
			float wx = 0.0f, wz = 0.0f;
			float arc_t = r / (2.0f * max_linear_len);
			if (!(arc_t == 0.0f || max_linear_len == 0.0f)) {
				if (sq_curvature == 0.0f) { wx = max_linear_len * arc_t; wz = 0.0f; }
				else {
					float tpc = 2.0f * PI * sq_curvature;
					float s_tpc = max_linear_len / tpc;
					wx = s_tpc * sinf(tpc * arc_t); wz = s_tpc * (1.0f - cosf(tpc * arc_t));
				}
			}
			*origin = orc_v3(wx * cosf(a), wx * sinf(a), wz);
			*dir = orc_v3(0.0f, 0.0f, 1.0f);
			if (sq_curvature != 0.0f) {
				orc_vec3 sc = orc_v3(0.0f, 0.0f, max_linear_len / (2.0f * PI * sq_curvature));
				float k = sq_curvature > 0.0f ? 1.0f : -1.0f;
				*dir = orc_scale(orc_normalized(orc_sub(sc, *origin)), k);
			}
			*origin = orc_add(orc_mat3_mul(c, *origin), orc_col(c, 3));
			*dir = orc_mat3_mul(c, *dir);
			apply_aperture(spp, x, y, c, aperture_size, focus_z, origin, dir);
			*origin = orc_add(*origin, orc_scale(*dir, near_distance));
	} else {                          /* quadrilateral_hexahedron_pixel_to_ray, :80-118 */
			float u = ((float)x + 0.5f) / rx, v = ((float)y + 0.5f) / ry;
			const float *f = qh_front, *b = qh_back;
			orc_vec3 f_ab = lerp3(f + 0, f + 3, u), f_dc = lerp3(f + 6, f + 9, u);
			orc_vec3 front_p = orc_add(f_ab, orc_scale(orc_sub(f_dc, f_ab), v));
			orc_vec3 b_ab = lerp3(b + 0, b + 3, u), b_dc = lerp3(b + 6, b + 9, u);
			orc_vec3 back_p = orc_add(b_ab, orc_scale(orc_sub(b_dc, b_ab), v));
			*dir = orc_sub(front_p, back_p);
			*dir = orc_v3(dir->x / dir->z, dir->y / dir->z, dir->z / dir->z);
			*origin = orc_add(orc_mat3_mul(c, back_p), orc_col(c, 3));
			*dir = orc_mat3_mul(c, *dir);
			apply_aperture(spp, x, y, c, aperture_size, focus_z, origin, dir);
			*origin = orc_add(*origin, orc_scale(*dir, near_distance));
		}
}

void orc_multi_init_global_rays(uint32_t sample_index, orc_global_ray* rays, float* depthbuffer, const orc_downsample_info* ds, const orc_render_camera* cam) { /* nerf_renderer.cu:17-94 */
	for (int32_t ty = 0; ty < ds->scaled_res[1]; ++ty) for (int32_t tx = 0; tx < ds->scaled_res[0]; ++tx) {
		uint32_t x = (uint32_t)tx, y = (uint32_t)ty;
		const uint32_t idx = x + (uint32_t)ds->scaled_res[0] * y;
		x *= (uint32_t)ds->skip[0]; y *= (uint32_t)ds->skip[1];
		if (x >= (uint32_t)ds->max_res[0] || y >= (uint32_t)ds->max_res[1]) continue;
		const float rx = (float)ds->max_res[0], ry = (float)ds->max_res[1];
		orc_vec3 origin = orc_v3(0.f, 0.f, 0.f), dir = orc_v3(0.f, 0.f, 1.f);
		const float* c = cam->transform;
		if (cam->model == 0) {            /* perspective_pixel_to_ray, camera_models.cuh:206-241 */
			float off[2];
			orc_ld_random_pixel_offset(sample_index, off);
			float u = ((float)x + off[0]) / rx, v = ((float)y + off[1]) / ry;
			dir = orc_v3((u - 0.5f) * rx / cam->focal_length, (v - 0.5f) * ry / cam->focal_length, 1.0f);
			dir = orc_mat3_mul(c, dir);
			origin = orc_add(orc_mat3_mul(c, orc_v3(0.f, 0.f, 0.f)), orc_col(c, 3));
			apply_aperture(sample_index, x, y, c, cam->aperture_size, cam->focus_z, &origin, &dir);
			origin = orc_add(origin, orc_scale(dir, cam->near_distance));
		} else {
			orc_extra_camera_model_pixel_to_ray(cam->model, sample_index, x, y, rx, ry, c, cam->sq_width, cam->sq_height, cam->sq_curvature, cam->qh_front, cam->qh_back,
			                                    cam->near_distance, cam->focus_z, cam->aperture_size, &origin, &dir);
		}
		depthbuffer[idx] = 1e10f;
		orc_global_ray* ray = &rays[idx];
		dir = orc_normalized(dir);
		ray->origin[0] = origin.x; ray->origin[1] = origin.y; ray->origin[2] = origin.z;
		ray->dir[0] = dir.x; ray->dir[1] = dir.y; ray->dir[2] = dir.z;
		ray->rgba[0] = ray->rgba[1] = ray->rgba[2] = ray->rgba[3] = 0.0f;
		ray->idx = idx; ray->depth = 0.0f; ray->alive = 1; ray->pad_[0] = ray->pad_[1] = ray->pad_[2] = 0;
	}
}

void orc_multi_init_proxy_rays(uint32_t n_elements, const orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, const orc_nerf_props* props) { /* :96-146 */
	for (uint32_t i = 0; i < n_elements; ++i) {
		const orc_global_ray* g = &global_rays[i];
		orc_proxy_ray* p = &proxy_rays[g->idx];
		if (!g->alive) { p->alive = 0; p->origin[0] = p->origin[1] = p->origin[2] = 0.0f; continue; }
		orc_vec3 origin = xform_point(props->itransform, ld3(g->origin));                         /* localized_point, bounding_box.cuh:224-226 */
		orc_vec3 dir = orc_normalized(xform_dir(props->itransform, orc_normalized(ld3(g->dir)))); /* localized_direction, :228-231 */
		p->dir[0] = dir.x; p->dir[1] = dir.y; p->dir[2] = dir.z;
		float tt[2];
		orc_aabb_ray_intersect(&props->render_aabb, origin, dir, tt);
		float t = fmaxf(tt[0], 0.0f) + 1e-5f;
		if (!orc_aabb_contains(&props->render_aabb, orc_add(origin, orc_scale(dir, t)))) { p->alive = 0; continue; }
		int hits = props->n_masks == 0;
		if (!hits) {
			for (uint32_t k = 0; k < props->n_masks; ++k) {
				float o3[3] = {origin.x, origin.y, origin.z}, d3[3] = {dir.x, dir.y, dir.z};
				if (orc_mask_intersects_ray(&props->masks[k], o3, d3)) { hits = 1; break; }
			}
		}
		p->active = 1;
		p->alive = hits ? 1 : 0;
		p->idx = g->idx;
		p->t = 0.0f;
		p->n_steps = 0;
		orc_vec3 o2 = orc_add(origin, orc_scale(dir, t));
		p->origin[0] = o2.x; p->origin[1] = o2.y; p->origin[2] = o2.z;
	}
}

/* :148-208.  The per-mask test inside the loop is dead code in the reference (an unconditional `break` precedes it). */
static int hit_test_and_march(orc_vec3 origin, orc_vec3 dir, orc_vec3 idir, float proxy_t, const orc_nerf_props* props, float* t_out, float* dt_out) {
	float t = proxy_t, dt = 0.0f, prev_t = t;
	while (1) {
		orc_vec3 pos = orc_add(origin, orc_scale(dir, t));
		if (!orc_aabb_contains(&props->render_aabb, pos)) {
			if (t_out) *t_out = prev_t;
			if (dt_out) *dt_out = dt;
			return 0;
		}
		dt = get_dt(t, props->cone_angle, props->min_cone_stepsize, props->max_cone_stepsize);
		int mipi = get_mip_from_dt(dt, pos, props->grid_size, props->nerf_cascades - 1);
		uint32_t mip = (uint32_t)(mipi < 0 ? 0 : mipi);
		if (!props->density_grid_bitfield) break;
		if (get_is_occupied(pos, props->density_grid_bitfield, mip, props->grid_size, props->grid_volume)) break;
		uint32_t res = props->grid_size >> mip;
		prev_t = t;
		t = get_t_advanced_to_next_voxel(t, props->cone_angle, pos, dir, idir, res, props->min_cone_stepsize, props->max_cone_stepsize);
	}
	if (t_out) *t_out = t;
	if (dt_out) *dt_out = dt;
	return 1;
}

void orc_multi_compact_rays(uint32_t n_elements, const orc_global_ray* g_src, orc_global_ray* g_dst, const orc_proxy_ray* p_src, orc_proxy_ray* p_dst, uint32_t n_nerfs,
                            uint32_t stride, orc_global_ray* g_final, uint32_t* alive_counter, uint32_t* final_counter) {       /* :237-268 (order: thread index) */
	for (uint32_t i = 0; i < n_elements; ++i) {
		const orc_global_ray* g = &g_src[i];
		if (g->alive) {
			uint32_t idx = (*alive_counter)++;
			g_dst[idx] = *g;
			for (uint32_t n = 0; n < n_nerfs; ++n) p_dst[idx + n * stride] = p_src[i + n * stride];
		} else if (g->rgba[3] > 0.001f) {
			g_final[(*final_counter)++] = *g;
		}
	}
}

void orc_multi_march_active_rays(uint32_t n_rays_alive, uint32_t n_nerfs, const orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, uint32_t stride, const orc_nerf_props* props) { /* :272-316 */
	for (uint32_t i = 0; i < n_rays_alive; ++i) {
		if (!global_rays[i].alive) continue;
		for (uint32_t n = 0; n < n_nerfs; ++n) {
			orc_proxy_ray* p = &proxy_rays[i + n * stride];
			if (!p->alive || !p->active) continue;
			orc_vec3 origin = ld3(p->origin), dir = ld3(p->dir);
			orc_vec3 idir = orc_v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
			float t = p->t;
			p->alive = hit_test_and_march(origin, dir, idir, t, props + n, &t, NULL) ? 1 : 0;
			p->t = t;
		}
	}
}

void orc_multi_cull_rays(uint32_t n_rays_alive, uint32_t n_nerfs, orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, uint32_t stride, const float cam_pos_in[3], const orc_nerf_props* props) { /* :378-428 */
	const orc_vec3 cam_pos = ld3(cam_pos_in);
	for (uint32_t i = 0; i < n_rays_alive; ++i) {
		orc_global_ray* g = &global_rays[i];
		if (!g->alive) continue;
		float min_d2 = 0.0f;
		int32_t active_idx = -1;
		uint32_t n_proxy_alive = 0;
		for (uint32_t n = 0; n < n_nerfs; ++n) {
			const uint32_t pi = i + n * stride;
			orc_proxy_ray* p = &proxy_rays[pi];
			if (!p->alive) continue;
			++n_proxy_alive;
			orc_vec3 pw = xform_point(props[n].transform, orc_add(ld3(p->origin), orc_scale(ld3(p->dir), p->t)));
			orc_vec3 dlt = orc_sub(pw, cam_pos);
			float d2 = orc_dot(dlt, dlt);
			if (d2 < min_d2 || active_idx == -1) { min_d2 = d2; active_idx = (int32_t)pi; }
			p->active = 0;
		}
		if (active_idx >= 0) proxy_rays[active_idx].active = 1;
		if (n_proxy_alive == 0) g->alive = 0;
	}
}

void orc_multi_generate_next_inputs(uint32_t n_elements, const orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, orc_coord* network_input, uint32_t n_steps, const orc_nerf_props* props) { /* :318-375 */
	for (uint32_t i = 0; i < n_elements; ++i) {
		if (!global_rays[i].alive) continue;
		orc_proxy_ray* p = &proxy_rays[i];
		if (!p->active) continue;
		orc_vec3 origin = ld3(p->origin), dir = ld3(p->dir);
		orc_vec3 idir = orc_v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
		orc_vec3 wd = orc_warp_direction(dir);
		float t = p->t;
		float dt = get_dt(t, props->cone_angle, props->min_cone_stepsize, props->max_cone_stepsize);
		int exited = 0;
		for (uint32_t j = 0; j < n_steps; ++j) {
			orc_vec3 pos = orc_add(origin, orc_scale(dir, t));
			orc_vec3 wp = orc_aabb_relative_pos(&props->train_aabb, pos);
			orc_coord* c = &network_input[i + (size_t)j * n_elements];
			c->pos[0] = wp.x; c->pos[1] = wp.y; c->pos[2] = wp.z;
			c->dt = get_warped_dt(dt, props->min_cone_stepsize, props->nerf_cascades);
			c->dir[0] = wd.x; c->dir[1] = wd.y; c->dir[2] = wd.z;
			if (!hit_test_and_march(origin, dir, idir, t, props, &t, &dt)) { p->n_steps = (uint16_t)j; exited = 1; break; }
			t += dt;
		}
		if (exited) continue;
		p->t = t;
		p->n_steps = (uint16_t)n_steps;
	}
}

void orc_multi_composite(uint32_t n_global_rays, uint32_t current_step, orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, const orc_coord* network_input,
                         const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance,
                         const orc_nerf_props* props) {                                                                         /* :431-510 */
	for (uint32_t i = 0; i < n_global_rays; ++i) {
		orc_global_ray* g = &global_rays[i];
		if (!g->alive) continue;
		orc_proxy_ray* p = &proxy_rays[i];
		if (!p->alive || !p->active) continue;
		float r = g->rgba[0], gg = g->rgba[1], b = g->rgba[2], a = g->rgba[3];
		uint32_t actual_n_steps = p->n_steps;
		uint32_t j = 0;
		for (; j < actual_n_steps; ++j) {
			size_t e = i + (size_t)j * n_global_rays;
			const uint16_t* o = network_output + e * out_stride;
			const orc_coord* in = &network_input[e];
			orc_vec3 pos = orc_unwarp_position(ld3(in->pos), &props->train_aabb);
			float T = 1.f - a;
			float dt = get_unwarped_dt(in->dt, props->min_cone_stepsize, props->nerf_cascades);
			float alpha = 1.f - expf(-orc_network_to_density(orc_h2f(o[3]), density_activation) * dt);
			float weight = alpha * T;
			float cr = orc_network_to_rgb(orc_h2f(o[0]), rgb_activation), cg = orc_network_to_rgb(orc_h2f(o[1]), rgb_activation), cb = orc_network_to_rgb(orc_h2f(o[2]), rgb_activation);
			float mask_weight = 1.f;
			for (uint32_t k = 0; k < props->n_masks; ++k) {
				float p3[3] = {pos.x, pos.y, pos.z};
				mask_weight = orc_clampf(mask_weight + orc_mask_sample(&props->masks[k], p3), 0.0f, 1.0f);
			}
			weight *= mask_weight;
			weight *= props->opacity;
			r += cr * weight; gg += cg * weight; b += cb * weight; a += weight;
			if (a > (1.0f - min_transmittance)) { r /= a; gg /= a; b /= a; a /= a; break; }
		}
		if (j < n_steps) { p->alive = 0; p->n_steps = (uint16_t)(j + current_step); }
		g->rgba[0] = r; g->rgba[1] = gg; g->rgba[2] = b; g->rgba[3] = a;
	}
}

void orc_multi_shade(uint32_t n_rays, const orc_global_ray* rays, int train_in_linear_colors, float* frame_buffer, float* depth_buffer, const orc_downsample_info* ds, int flip_y) { /* :512-563 */
	for (uint32_t i = 0; i < n_rays; ++i) {
		const orc_global_ray* ray = &rays[i];
		uint32_t x = (uint32_t)ds->skip[0] * (ray->idx % (uint32_t)ds->scaled_res[0]);
		uint32_t y = (uint32_t)ds->skip[1] * (ray->idx / (uint32_t)ds->scaled_res[0]);
		if (flip_y) y = (uint32_t)ds->max_res[1] - y - 1;
		float tmp[4] = {ray->rgba[0], ray->rgba[1], ray->rgba[2], ray->rgba[3]};
		if (!train_in_linear_colors) { tmp[0] = orc_srgb_to_linear(tmp[0]); tmp[1] = orc_srgb_to_linear(tmp[1]); tmp[2] = orc_srgb_to_linear(tmp[2]); }
		for (uint32_t u = 0; u < (uint32_t)ds->skip[0]; ++u) for (uint32_t v = 0; v < (uint32_t)ds->skip[1]; ++v) {
			uint32_t idx = (x + u) + (y + v) * (uint32_t)ds->max_res[0];
			if (idx >= ds->max_pixels) continue;
			float* f = frame_buffer + (size_t)idx * 4;
			float k = 1.0f - tmp[3];
			for (int c = 0; c < 4; ++c) f[c] = tmp[c] + f[c] * k;
			if (tmp[3] > 0.2f) depth_buffer[idx] = ray->depth;
		}
	}
}

/* NerfRenderer::render + init_rays_from_camera + march_rays_and_accumulate_colors (:565-791), for n_nerfs networks (nets[n], params[n]).
 * frame_buffer must come in cleared (bl_render_frame clears it, testbed.cu:2679).  Returns the number of network samples consumed. */
uint64_t orc_multi_render(uint32_t n_nerfs, const orc_net* const* nets, const uint16_t* const* params, const orc_nerf_props* props, const int* rgb_activation,
                          const int* density_activation, const float* min_transmittance, const orc_downsample_info* ds, const orc_render_camera* cam, int flip_y,
                          float* frame_buffer, float* depth_buffer) {
	const uint32_t n_pixels = ds->scaled_pixels;
	const uint32_t stride = (n_pixels + 127u) / 128u * 128u;   /* get_n_pixels_padded, render_data_workspace.cuh:63-65 */
	orc_global_ray* global[2]; orc_proxy_ray* proxy[2];
	for (int b = 0; b < 2; ++b) { global[b] = (orc_global_ray*)calloc(stride, sizeof(orc_global_ray)); proxy[b] = (orc_proxy_ray*)calloc((size_t)stride * n_nerfs, sizeof(orc_proxy_ray)); }
	orc_global_ray* hit = (orc_global_ray*)calloc(stride, sizeof(orc_global_ray));
	orc_coord* net_in = (orc_coord*)calloc((size_t)stride * 8, sizeof(orc_coord));
	uint16_t* net_out = (uint16_t*)calloc((size_t)stride * 8 * 4, 2);

	orc_multi_init_global_rays(0, global[0], depth_buffer, ds, cam);
	for (uint32_t n = 0; n < n_nerfs; ++n) orc_multi_init_proxy_rays(n_pixels, global[0], proxy[0] + (size_t)n * stride, props + n);

	uint32_t n_alive = n_pixels, n_hit = 0, i = 1, dbi = 0;
	uint64_t n_samples = 0;
	const float cam_pos[3] = {cam->transform[9], cam->transform[10], cam->transform[11]};
	while (i < 10000) {
		const int tmp = dbi % 2, cur = (dbi + 1) % 2;
		++dbi;
		uint32_t alive = 0;
		orc_multi_compact_rays(n_alive, global[tmp], global[cur], proxy[tmp], proxy[cur], n_nerfs, stride, hit, &alive, &n_hit);
		n_alive = alive;
		if (n_alive == 0) break;
		orc_multi_march_active_rays(n_alive, n_nerfs, global[cur], proxy[cur], stride, props);
		orc_multi_cull_rays(n_alive, n_nerfs, global[cur], proxy[cur], stride, cam_pos, props);
		uint32_t n_steps = n_pixels / n_alive; n_steps = n_steps < 1 ? 1 : (n_steps > 8 ? 8 : n_steps);
		for (uint32_t n = 0; n < n_nerfs; ++n) {
			orc_proxy_ray* pr = proxy[cur] + (size_t)n * stride;
			orc_multi_generate_next_inputs(n_alive, global[cur], pr, net_in, n_steps, props + n);
			/* only the slots the compositor reads are evaluated */
			for (uint32_t r = 0; r < n_alive; ++r) {
				if (!global[cur][r].alive || !pr[r].alive || !pr[r].active) continue;
				for (uint32_t j = 0; j < pr[r].n_steps; ++j) {
					size_t s = r + (size_t)j * n_alive;
					orc_nerf_inference(nets[n], params[n], (const float*)&net_in[s], 7, 1, net_out + s * 4, 4);
					++n_samples;
				}
			}
			orc_multi_composite(n_alive, i, global[cur], pr, net_in, net_out, 4, n_steps, rgb_activation[n], density_activation[n], min_transmittance[n], props + n);
		}
		i += n_steps;
	}
	orc_multi_shade(n_hit, hit, 0, frame_buffer, depth_buffer, ds, flip_y);
	for (int b = 0; b < 2; ++b) { free(global[b]); free(proxy[b]); }
	free(hit); free(net_in); free(net_out);
	return n_samples;
}
