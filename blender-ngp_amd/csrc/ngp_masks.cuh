// ngp_masks.cuh — the fork's SDF crop masks (include/neural-graphics-primitives/nerf/mask_3D.cuh:32-255), shared by the Blender multi-NeRF
// renderer (multi_render.hip) and the stock tracer's render_masks (render.hip; src/testbed_nerf.cu:833-840, 1943-1956).
#pragma once
#include "ngp_device.cuh"

namespace ngp {

// ---- 4x4 column-major helpers (Eigen: M * p.homogeneous() accumulates column by column)
__device__ __forceinline__ v3 xform_point(const float* m, v3 p) {
	return mk(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12], ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13], ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
__device__ __forceinline__ v3 xform_dir(const float* m, v3 d) {
	return mk((m[0] * d.x + m[4] * d.y) + m[8] * d.z, (m[1] * d.x + m[5] * d.y) + m[9] * d.z, (m[2] * d.x + m[6] * d.y) + m[10] * d.z);
}

// ---- mask_3D.cuh
__device__ __forceinline__ float sdf_box(v3 p, v3 b) {
	const v3 d = mk(fabsf(p.x) - 0.5f * b.x, fabsf(p.y) - 0.5f * b.y, fabsf(p.z) - 0.5f * b.z);
	const v3 dm = mk(fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f), fmaxf(d.z, 0.0f));
	return norm(dm) + fminf(fmaxf(d.x, fmaxf(d.y, d.z)), 0.0f);
}
__device__ __forceinline__ float sdf_cylinder(v3 p, float r, float h) {
	const float dx = fabsf(sqrtf(p.y * p.y + p.x * p.x)) - r, dy = fabsf(p.z) - 0.5f * h;
	const float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
	return sqrtf(mx * mx + my * my) + fminf(fmaxf(dx, dy), 0.0f);
}
__device__ __forceinline__ float mask_signed_distance(const NgpMask3D& m, v3 p) {
	const v3 pl = xform_point(m.itransform, p);
	float d = 0.0f;
	switch (m.shape) {
		case 0: d = sdf_box(pl, mk(m.config[0], m.config[1], m.config[2])); break;
		case 1: d = sdf_cylinder(pl, m.config[0], m.config[1]); break;
		case 2: d = norm(pl) - m.config[0]; break;
		default: d = -1.0f; break;
	}
	return d * (m.mode == 0 ? 1.0f : -1.0f);
}
__device__ __forceinline__ float mask_sample(const NgpMask3D& m, v3 p) {
	const float k = m.mode == 0 ? 1.0f : -1.0f;
	if (m.shape == 3) return k;
	const float d = mask_signed_distance(m, p);
	float alpha;
	if (m.feather == 0.0f) alpha = d < 0.0f ? 1.0f : 0.0f;
	else alpha = clampf(0.5f - d / m.feather, 0.0f, 1.0f);
	return m.opacity * alpha * k;
}
__device__ __forceinline__ bool ray_intersects_box(v3 o, v3 d, v3 size) {
	const v3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
	const v3 t0 = mk((-0.5f * size.x - o.x) * inv.x, (-0.5f * size.y - o.y) * inv.y, (-0.5f * size.z - o.z) * inv.z);
	const v3 t1 = mk((0.5f * size.x - o.x) * inv.x, (0.5f * size.y - o.y) * inv.y, (0.5f * size.z - o.z) * inv.z);
	const float tmin = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
	const float tmax = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
	return tmin <= tmax;
}
__device__ __forceinline__ bool ray_intersects_sphere(v3 o, v3 d, float radius) {
	const float od = dot(d, o);
	const float a = od * od;   // powf(x, 2)
	const float b = dot(o, o) - radius * radius;
	return !((a - b) < 0.0f);
}
__device__ __forceinline__ bool intersect_plane_ray(v3 o, v3 d, v3 n, v3 p, float& t) {
	const float denom = dot(n, d);
	if (denom > 1e-6f) { t = dot(p - o, n) / denom; return t >= 0.0f; }
	return false;
}
__device__ __forceinline__ bool ray_intersects_cylinder(v3 o, v3 d, float radius, float height) {
	const float a = d.x * d.x + d.y * d.y;
	const float b = 2.0f * (d.x * o.x + d.y * o.y);
	const float c = (o.x * o.x + o.y * o.y) - radius * radius;
	const float disc = b * b - 4.0f * a * c;
	if (disc < 0.0f) return false;
	const float d_sqrt = sqrtf(disc), a2 = 2.0f * a, h_2 = 0.5f * height;
	if (a2 > 1e-6f) {
		const float t0 = (-b - d_sqrt) / a2, t1 = (-b + d_sqrt) / a2;
		const float z0 = o.z + t0 * d.z, z1 = o.z + t1 * d.z;
		if ((z0 >= -h_2 && z0 <= h_2) || (z1 >= -h_2 && z1 <= h_2)) return true;
	}
	float t = 0.0f;
	if (intersect_plane_ray(o, d, mk(0.0f, 0.0f, 1.0f), mk(0.0f, 0.0f, h_2), t)) {
		const v3 p = o + d * t;
		if (p.x * p.x + p.y * p.y <= radius * radius) return true;
	}
	if (intersect_plane_ray(o, d, mk(0.0f, 0.0f, -1.0f), mk(0.0f, 0.0f, -h_2), t)) {
		const v3 p = o + d * t;
		if (p.x * p.x + p.y * p.y <= radius * radius) return true;
	}
	return false;
}
__device__ __forceinline__ bool mask_intersects_ray(const NgpMask3D& m, v3 ro, v3 rd) {
	if (m.mode == 1) return true;        // subtract masks have infinite additive area around them
	if (m.shape == 3) return m.mode == 0;
	const v3 ol = xform_point(m.itransform, ro);
	const v3 dl = normalized(xform_dir(m.itransform, rd));
	const float f = 0.5f * m.feather;
	switch (m.shape) {
		case 0: return ray_intersects_box(ol, dl, mk(m.config[0] + f, m.config[1] + f, m.config[2] + f));
		case 1: return ray_intersects_cylinder(ol, dl, m.config[0] + f, m.config[1] + f);
		case 2: return ray_intersects_sphere(ol, dl, m.config[0] + f);
		default: return true;
	}
}

}  // namespace ngp
