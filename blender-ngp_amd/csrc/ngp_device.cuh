// Device-side helpers shared by the gfx950 kernels (wave64).  Cited lines are the reference definitions each helper replaces.
// Everything on the integer / index path is written with explicit fp32 operation order and the library is compiled with
// -ffp-contract=off so that results are bit-identical to a plain CPU evaluation (north star: bit-exact occupancy indices
// and sample counts).  Float-tolerance kernels re-enable contraction locally.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ngp_hip.h"

#define NGP_WAVE 64

namespace ngp {

// ---------------------------------------------------------------- error plumbing (host)
void set_last_error(const char* what, hipError_t e);
#define NGP_LAUNCH_CHECK(name)                                  \
	do {                                                        \
		hipError_t e__ = hipGetLastError();                     \
		if (e__ != hipSuccess) { ::ngp::set_last_error(name, e__); return (int)e__; } \
	} while (0)
#define NGP_HIP_TRY(expr)                                       \
	do {                                                        \
		hipError_t e__ = (expr);                                \
		if (e__ != hipSuccess) { ::ngp::set_last_error(#expr, e__); return (int)e__; } \
	} while (0)

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- small vector type
struct v3 { float x, y, z; };
__device__ __forceinline__ v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float norm(v3 a) { return sqrtf(dot(a, a)); }
// Eigen normalized(): v / sqrt(squaredNorm) when squaredNorm > 0
__device__ __forceinline__ v3 normalized(v3 a) {
	float z = dot(a, a);
	if (z > 0.0f) { float n = sqrtf(z); return mk(a.x / n, a.y / n, a.z / n); }
	return a;
}
// column-major 3x3 (or the rotation part of a 3x4) times vector
__device__ __forceinline__ v3 mat3_mul(const float* m, v3 v) {
	return mk(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
__device__ __forceinline__ v3 col(const float* m, int c) { return mk(m[3 * c], m[3 * c + 1], m[3 * c + 2]); }

struct Aabb { v3 mn, mx; };
static inline Aabb aabb_from_host(const NgpAabb* a) {
	Aabb r; r.mn.x = a->min[0]; r.mn.y = a->min[1]; r.mn.z = a->min[2]; r.mx.x = a->max[0]; r.mx.y = a->max[1]; r.mx.z = a->max[2]; return r;
}
struct Mat34 { float m[12]; };
struct Mat33 { float m[9]; };

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------- constants (testbed_nerf.cu:53-73)
__device__ __forceinline__ constexpr float SQRT3() { return 1.73205080757f; }
__device__ __forceinline__ constexpr float STEPSIZE() { return SQRT3() / 1024.0f; }
__device__ __forceinline__ constexpr float MIN_CONE_STEPSIZE() { return STEPSIZE(); }
__device__ __forceinline__ constexpr float MAX_CONE_STEPSIZE() { return STEPSIZE() * 128.0f * 1024.0f / 128.0f; }
__device__ __forceinline__ constexpr float MIN_OPTICAL_THICKNESS() { return 0.01f; }

// ---------------------------------------------------------------- pcg32 (tcnn pcg32.h; random_val.cuh:28-45)
struct Pcg32 {
	uint64_t state, inc;
	__device__ __forceinline__ uint32_t next_uint() {
		uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	__device__ __forceinline__ float next_float() {
		return __uint_as_float((next_uint() >> 9) | 0x3f800000u) - 1.0f;
	}
	__device__ __forceinline__ void advance(uint64_t delta) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta >>= 1;
		}
		state = acc_mult * state + acc_plus;
	}
};

// default_rng_t{seed}: pcg32(initstate = seed, initseq = 1) [tcnn pcg32.h constructor]
__device__ __forceinline__ Pcg32 pcg32_seeded(uint64_t seed) {
	Pcg32 r; r.state = 0u; r.inc = 3u;
	r.next_uint(); r.state += seed; r.next_uint();
	return r;
}

// ---------------------------------------------------------------- LK-scrambled Sobol (random_val.cuh:88-288, dims 0/1 only)
__device__ __forceinline__ uint32_t sobol01(uint32_t index, uint32_t dim) {
	// dim 0: direction[bit] = 0x80000000 >> bit  => bit reversal.
	if (dim == 0) return __brev(index);
	// dim 1: direction numbers of random_val.cuh:99-106
	const uint32_t d1[32] = {
		0x80000000, 0xc0000000, 0xa0000000, 0xf0000000, 0x88000000, 0xcc000000, 0xaa000000, 0xff000000,
		0x80800000, 0xc0c00000, 0xa0a00000, 0xf0f00000, 0x88880000, 0xcccc0000, 0xaaaa0000, 0xffff0000,
		0x80008000, 0xc000c000, 0xa000a000, 0xf000f000, 0x88008800, 0xcc00cc00, 0xaa00aa00, 0xff00ff00,
		0x80808080, 0xc0c0c0c0, 0xa0a0a0a0, 0xf0f0f0f0, 0x88888888, 0xcccccccc, 0xaaaaaaaa, 0xffffffff};
	uint32_t X = 0;
#pragma unroll
	for (uint32_t bit = 0; bit < 32; bit++) X ^= ((index >> bit) & 1u) * d1[bit];
	return X;
}
__device__ __forceinline__ uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
__device__ __forceinline__ uint32_t lk_perm(uint32_t x, uint32_t seed) {
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
__device__ __forceinline__ uint32_t nus_base2(uint32_t x, uint32_t seed) { return __brev(lk_perm(__brev(x), seed)); }
__device__ __forceinline__ float ld_random_val(uint32_t index, uint32_t seed, uint32_t dim = 0) {
	const float S = (float)(1.0 / 4294967296.0);
	index = nus_base2(index, seed);
	return (float)nus_base2(sobol01(index, dim), hash_combine(seed, dim)) * S;
}
__device__ __forceinline__ void ld_random_pixel_offset(uint32_t spp, float& ox, float& oy) {
	float a0 = ld_random_val(0, 0xdeadbeefu, 0), a1 = ld_random_val(0, 0xdeadbeefu, 1);
	float b0 = ld_random_val(spp, 0xdeadbeefu, 0), b1 = ld_random_val(spp, 0xdeadbeefu, 1);
	ox = 0.5f - a0 + b0; ox = ox - floorf(ox);
	oy = 0.5f - a1 + b1; oy = oy - floorf(oy);
}

// ---- depth of field (shared by the stock renderer and the Blender camera models)
__device__ __forceinline__ void square2disk_shirley(float a, float b, float& ox, float& oy) {  // random_val.cuh:109-125
	const float PI = 3.14159265358979323846f;
	float phi, r;
	if (a * a > b * b) { r = a; phi = (PI / 4.0f) * (b / a); }
	else { r = b; phi = (PI / 2.0f) - (PI / 4.0f) * (a / b); }
	ox = r * cosf(phi); oy = r * sinf(phi);
}
__device__ __forceinline__ void apply_aperture(uint32_t spp, uint32_t px, uint32_t py, const float* cam, float aperture_size, float focus_z, v3& origin, v3& dir) {
	if (aperture_size > 0.0f) {
		const v3 lookat = origin + dir * focus_z;
		const uint32_t seed = px * 19349663u + py * 96925573u;
		const float r0 = ld_random_val(spp, seed, 0) * 2.0f - 1.0f, r1 = ld_random_val(spp, seed, 1) * 2.0f - 1.0f;
		float bx, by;
		square2disk_shirley(r0, r1, bx, by);
		bx *= aperture_size; by *= aperture_size;
		origin = origin + mk(cam[0] * bx + cam[3] * by, cam[1] * bx + cam[4] * by, cam[2] * bx + cam[5] * by);
		dir = mk((lookat.x - origin.x) / focus_z, (lookat.y - origin.y) / focus_z, (lookat.z - origin.z) / focus_z);
	}
}

// ---- the fork's extra camera models (camera_models.cuh), shared by the stock renderer and the Blender renderer.  c = 3x4 column-major
// camera-to-world, (x, y) = pixel, (rx, ry) = resolution
__device__ __forceinline__ v3 lerp3(const float* a, const float* b, float t) { return mk(a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), a[2] + t * (b[2] - a[2])); }
// spherical_quadrilateral_pixel_to_ray (camera_models.cuh:162-203) with walk_along_sphere / walk_along_circle (133-160)
__device__ __forceinline__ void spherical_quadrilateral_pixel_to_ray(uint32_t spp, uint32_t x, uint32_t y, float rx, float ry, const float* c, float sq_width, float sq_height, float sq_curvature,
                                                                     float near_distance, float focus_z, float aperture_size, v3& origin, v3& dir) {
	const float PI = 3.14159265358979323846f;
	const float max_linear_len = sqrtf(sq_width * sq_width + sq_height * sq_height);
	const float ux = 2.0f * (((float)x + 0.5f) / rx - 0.5f), uy = 2.0f * (((float)y + 0.5f) / ry - 0.5f);
	const float qx = sq_width * ux, qy = sq_height * uy;
	const float a = atan2f(qy, qx), r = sqrtf(qx * qx + qy * qy);
	float wx = 0.0f, wz = 0.0f;
	const float arc_t = r / (2.0f * max_linear_len);
	if (!(arc_t == 0.0f || max_linear_len == 0.0f)) {
		if (sq_curvature == 0.0f) { wx = max_linear_len * arc_t; wz = 0.0f; }
		else {
			const float tpc = 2.0f * PI * sq_curvature;
			const float s_tpc = max_linear_len / tpc;
			wx = s_tpc * sinf(tpc * arc_t); wz = s_tpc * (1.0f - cosf(tpc * arc_t));
		}
	}
	origin = mk(wx * cosf(a), wx * sinf(a), wz);
	dir = mk(0.0f, 0.0f, 1.0f);
	if (sq_curvature != 0.0f) {
		const v3 sc = mk(0.0f, 0.0f, max_linear_len / (2.0f * PI * sq_curvature));
		const float k = sq_curvature > 0.0f ? 1.0f : -1.0f;
		dir = normalized(sc - origin) * k;
	}
	origin = mat3_mul(c, origin) + col(c, 3);
	dir = mat3_mul(c, dir);
	apply_aperture(spp, x, y, c, aperture_size, focus_z, origin, dir);
	origin = origin + dir * near_distance;
}
// quadrilateral_hexahedron_pixel_to_ray (camera_models.cuh:80-118): front / back = tl, tr, bl, br corners of the two faces
__device__ __forceinline__ void quadrilateral_hexahedron_pixel_to_ray(uint32_t spp, uint32_t x, uint32_t y, float rx, float ry, const float* c, const float* f, const float* b,
                                                                      float near_distance, float focus_z, float aperture_size, v3& origin, v3& dir) {
	const float u = ((float)x + 0.5f) / rx, v = ((float)y + 0.5f) / ry;
	const v3 f_ab = lerp3(f + 0, f + 3, u), f_dc = lerp3(f + 6, f + 9, u);
	const v3 front_p = f_ab + (f_dc - f_ab) * v;
	const v3 b_ab = lerp3(b + 0, b + 3, u), b_dc = lerp3(b + 6, b + 9, u);
	const v3 back_p = b_ab + (b_dc - b_ab) * v;
	dir = front_p - back_p;
	dir = mk(dir.x / dir.z, dir.y / dir.z, dir.z / dir.z);
	origin = mat3_mul(c, back_p) + col(c, 3);
	dir = mat3_mul(c, dir);
	apply_aperture(spp, x, y, c, aperture_size, focus_z, origin, dir);
	origin = origin + dir * near_distance;
}

// ---- error-map importance sampling (testbed_nerf.cu:991-1083; off by default, testbed.h:668-669)
struct ErrorMapCdf { const float* cdf_x_cond_y; const float* cdf_y; const float* cdf_img; int32_t res[2]; };
// common.h:201-224
__device__ __forceinline__ uint32_t binary_search(float val, const float* __restrict__ data, uint32_t length) {
	if (length == 0) return 0;
	uint32_t first = 0, count = length;
	while (count > 0) {
		const uint32_t step = count / 2, it = first + step;
		if (data[it] < val) { first = it + 1; count -= step + 1; }
		else count = step;
	}
	return first < length - 1 ? first : length - 1;
}
// image_idx (1062-1083)
__device__ __forceinline__ uint32_t image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_training_images, const float* __restrict__ cdf, float* pdf) {
	if (cdf) {
		const float sample = ld_random_val(base_idx, 0xdeadbeefu);
		const uint32_t img = binary_search(sample, cdf, n_training_images);
		if (pdf) { const float prev = img > 0 ? cdf[img - 1] : 0.0f; *pdf = (cdf[img] - prev) * (float)n_training_images; }
		return img;
	}
	if (pdf) *pdf = 1.0f;
	return ((base_idx * n_training_images) / n_rays) % n_training_images;   // neighbouring rays read the same image
}
// sample_cdf_2d (991-1022): half of the samples stay uniform
__device__ __forceinline__ void sample_cdf_2d(float& sx, float& sy, uint32_t img, const int32_t res[2], const float* __restrict__ cdf_x_cond_y, const float* __restrict__ cdf_y, float* pdf) {
	constexpr float UNIFORM_SAMPLING_FRACTION = 0.5f;
	if (sx < UNIFORM_SAMPLING_FRACTION) { sx /= UNIFORM_SAMPLING_FRACTION; return; }
	sx = (sx - UNIFORM_SAMPLING_FRACTION) / (1.0f - UNIFORM_SAMPLING_FRACTION);
	cdf_y += (size_t)img * res[1];
	const uint32_t y = binary_search(sy, cdf_y, (uint32_t)res[1]);
	float prev = y > 0 ? cdf_y[y - 1] : 0.0f;
	const float pmf_y = cdf_y[y] - prev;
	sy = (sy - prev) / pmf_y;
	cdf_x_cond_y += (size_t)img * res[1] * res[0] + (size_t)y * res[0];
	const uint32_t x = binary_search(sx, cdf_x_cond_y, (uint32_t)res[0]);
	prev = x > 0 ? cdf_x_cond_y[x - 1] : 0.0f;
	const float pmf_x = cdf_x_cond_y[x] - prev;
	sx = (sx - prev) / pmf_x;
	if (pdf) *pdf = pmf_x * pmf_y * (float)(res[0] * res[1]);
	sx = ((float)x + sx) / (float)res[0];
	sy = ((float)y + sy) / (float)res[1];
}
// nerf_random_image_pos_training (1047-1060)
__device__ __forceinline__ void nerf_random_image_pos_training(Pcg32& rng, const int32_t res[2], int snap_to_pixel_centers, const ErrorMapCdf& cdf, uint32_t img, float& u, float& v, float* pdf) {
	u = rng.next_float(); v = rng.next_float();
	if (pdf) *pdf = 1.0f;
	if (cdf.cdf_x_cond_y) sample_cdf_2d(u, v, img, cdf.res, cdf.cdf_x_cond_y, cdf.cdf_y, pdf);
	if (snap_to_pixel_centers) {
		int px = (int)(u * (float)res[0]), py = (int)(v * (float)res[1]);
		px = px > 0 ? px : 0; px = px < res[0] - 1 ? px : res[0] - 1;
		py = py > 0 ? py : 0; py = py < res[1] - 1 ? py : res[1] - 1;
		u = ((float)px + 0.5f) / (float)res[0];
		v = ((float)py + 0.5f) / (float)res[1];
	}
}
__host__ inline ErrorMapCdf make_error_map_cdf(const NgpErrorMapCdf* h) {
	ErrorMapCdf c; c.cdf_x_cond_y = nullptr; c.cdf_y = nullptr; c.cdf_img = nullptr; c.res[0] = c.res[1] = 0;
	if (h) { c.cdf_x_cond_y = h->cdf_x_cond_y; c.cdf_y = h->cdf_y; c.cdf_img = h->cdf_img; c.res[0] = h->res[0]; c.res[1] = h->res[1]; }
	return c;
}


// ---------------------------------------------------------------- morton (tcnn common_device.h)
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
__device__ __forceinline__ uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249u;
	x = (x | (x >> 2)) & 0xc30c30c3u;
	x = (x | (x >> 4)) & 0x0f00f00fu;
	x = (x | (x >> 8)) & 0xff0000ffu;
	x = (x | (x >> 16)) & 0x0000ffffu;
	return x;
}

// ---------------------------------------------------------------- AABB (bounding_box.cuh:86-88, 163-221)
__device__ __forceinline__ bool aabb_contains(const Aabb& b, v3 p) {
	return p.x >= b.mn.x && p.x <= b.mx.x && p.y >= b.mn.y && p.y <= b.mx.y && p.z >= b.mn.z && p.z <= b.mx.z;
}
__device__ __forceinline__ void aabb_ray_intersect(const Aabb& b, v3 pos, v3 dir, float& tmin_out, float& tmax_out) {
	const float FMAX = 3.402823466e+38f;
	float tmin = (b.mn.x - pos.x) / dir.x, tmax = (b.mx.x - pos.x) / dir.x;
	if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
	float tymin = (b.mn.y - pos.y) / dir.y, tymax = (b.mx.y - pos.y) / dir.y;
	if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
	if (tmin > tymax || tymin > tmax) { tmin_out = tmax_out = FMAX; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (b.mn.z - pos.z) / dir.z, tzmax = (b.mx.z - pos.z) / dir.z;
	if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
	if (tmin > tzmax || tzmin > tmax) { tmin_out = tmax_out = FMAX; return; }
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	tmin_out = tmin; tmax_out = tmax;
}
__device__ __forceinline__ v3 aabb_relative_pos(const Aabb& b, v3 p) {
	return mk((p.x - b.mn.x) / (b.mx.x - b.mn.x), (p.y - b.mn.y) / (b.mx.y - b.mn.y), (p.z - b.mn.z) / (b.mx.z - b.mn.z));
}
__device__ __forceinline__ v3 unwarp_position(v3 p, const Aabb& b) {
	return mk(b.mn.x + p.x * (b.mx.x - b.mn.x), b.mn.y + p.y * (b.mx.y - b.mn.y), b.mn.z + p.z * (b.mx.z - b.mn.z));
}

// ---------------------------------------------------------------- marching helpers (testbed_nerf.cu:96-98, 191-213, 308-342, 449-463)
__device__ __forceinline__ float calc_dt(float t, float cone_angle) { return clampf(t * cone_angle, MIN_CONE_STEPSIZE(), MAX_CONE_STEPSIZE()); }
__device__ __forceinline__ float signf1(float x) { return copysignf(1.0f, x); }
__device__ __forceinline__ float distance_to_next_voxel(v3 pos, v3 dir, v3 idir, uint32_t res) {
	float r = (float)res;
	v3 p = mk(r * pos.x, r * pos.y, r * pos.z);
	float tx = (floorf(p.x + 0.5f + 0.5f * signf1(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * signf1(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * signf1(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	// r is a power of two (128 >> mip): its reciprocal is exact and built from the exponent bits; t * (1/r) == t / r bit for bit
	return fmaxf(t * __uint_as_float(0x7f000000u - __float_as_uint(r)), 0.0f);
}
// CONST_DT: cone_angle == 0 (aabb_scale 1), where calc_dt(t, 0) = clamp(t * 0, MIN, MAX) is MIN_CONE_STEPSIZE for every finite t — the same
// value without the three instructions per step
template <bool CONST_DT>
__device__ __forceinline__ float calc_dt_t(float t, float cone_angle) { return CONST_DT ? MIN_CONE_STEPSIZE() : calc_dt(t, cone_angle); }
template <bool CONST_DT = false>
__device__ __forceinline__ float advance_to_next_voxel(float t, float cone_angle, v3 pos, v3 dir, v3 idir, uint32_t res) {
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	do { t += calc_dt_t<CONST_DT>(t, cone_angle); } while (t < t_target);
	return t;
}
__device__ __forceinline__ float warp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE() * 128.0f;
	return (dt - MIN_CONE_STEPSIZE()) / (max_stepsize - MIN_CONE_STEPSIZE());
}
__device__ __forceinline__ float unwarp_dt(float dt) {
	float max_stepsize = MIN_CONE_STEPSIZE() * 128.0f;
	return dt * (max_stepsize - MIN_CONE_STEPSIZE()) + MIN_CONE_STEPSIZE();
}
__device__ __forceinline__ v3 warp_direction(v3 d) { return mk((d.x + 1.0f) * 0.5f, (d.y + 1.0f) * 0.5f, (d.z + 1.0f) * 0.5f); }

// exponent e of frexpf(x) = m * 2^e, m in [0.5, 1) (x >= 0 and finite here); frexp(0) -> 0.  v_frexp_exp_i32_f32 is exactly that
// (f32 denormals are kept in hipcc's default kernel mode).
__device__ __forceinline__ int frexp_exponent(float x) { return __builtin_amdgcn_frexp_expf(x); }
// cell coordinates of cascaded_grid_idx_at (testbed_nerf.cu:318-337), clamped to [0, 127]
__device__ __forceinline__ void cascaded_grid_coords(v3 pos, uint32_t mip, int& ix, int& iy, int& iz) {
	float mip_scale = __uint_as_float((127u - mip) << 23); // scalbnf(1, -mip)
	pos.x -= 0.5f; pos.y -= 0.5f; pos.z -= 0.5f;
	pos.x *= mip_scale; pos.y *= mip_scale; pos.z *= mip_scale;
	pos.x += 0.5f; pos.y += 0.5f; pos.z += 0.5f;
	ix = clampi((int)(pos.x * 128.0f), 0, 127); iy = clampi((int)(pos.y * 128.0f), 0, 127); iz = clampi((int)(pos.z * 128.0f), 0, 127);
}
__device__ __forceinline__ uint32_t cascaded_grid_idx_at(v3 pos, uint32_t mip) {
	int ix, iy, iz;
	cascaded_grid_coords(pos, mip, ix, iy, iz);
	return morton3D((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
}
__device__ __forceinline__ bool density_grid_occupied_at(v3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip) {
	uint32_t idx = cascaded_grid_idx_at(pos, mip);
	return bitfield[idx / 8 + (NGP_NERF_GRID_N_CELLS * mip) / 8] & (1u << (idx % 8));
}
// The bitfield is in Morton order, so 64 consecutive bits are one 4x4x4 brick of cells.  A marching ray stays inside a brick for
// several lookups (a brick is ~18 minimum steps across); keeping the last brick in two registers turns most of the dependent
// 1-byte loads of the serial march into a compare.  Same bit as density_grid_occupied_at, always.
struct OccBrick { uint32_t id = 0xffffffffu; uint64_t bits = 0; };
__device__ __forceinline__ bool density_grid_occupied_at(v3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip, OccBrick& cache) {
	const uint32_t idx = cascaded_grid_idx_at(pos, mip);
	const uint32_t brick = (idx >> 6) + (NGP_NERF_GRID_N_CELLS / 64u) * mip;
	if (brick != cache.id) { cache.id = brick; cache.bits = ((const uint64_t*)bitfield)[brick]; }
	return (cache.bits >> (idx & 63u)) & 1ull;
}
// ... and with a 4 KiB summary of cascade 0 in LDS (bit b = "brick b has an occupied cell"): in empty space — most iterations of a march —
// the answer is 0 without touching global memory, whose latency every lane of the wave would otherwise wait for.  Same bit, always.
__device__ __forceinline__ bool density_grid_occupied_at(v3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip, OccBrick& cache, const uint32_t* __restrict__ s_brick_any) {
	const uint32_t idx = cascaded_grid_idx_at(pos, mip);
	const uint32_t brick = (idx >> 6) + (NGP_NERF_GRID_N_CELLS / 64u) * mip;
	if (mip == 0 && !((s_brick_any[brick >> 5] >> (brick & 31u)) & 1u)) return false;
	if (brick != cache.id) { cache.id = brick; cache.bits = ((const uint64_t*)bitfield)[brick]; }
	return (cache.bits >> (idx & 63u)) & 1ull;
}
// fills s_brick_any[NGP_NERF_GRID_N_CELLS / 64 / 32] from cascade 0 of the bitfield; all threads of the workgroup call it, then __syncthreads()
__device__ __forceinline__ void load_brick_summary(const uint8_t* __restrict__ bitfield, uint32_t* __restrict__ s_brick_any) {
	constexpr uint32_t N_BRICKS = NGP_NERF_GRID_N_CELLS / 64u;
	const uint64_t* __restrict__ words = (const uint64_t*)bitfield;
	for (uint32_t w = threadIdx.x; w < N_BRICKS / 32u; w += blockDim.x) {
		uint32_t bits = 0;
#pragma unroll 8
		for (uint32_t k = 0; k < 32u; ++k) bits |= (words[w * 32u + k] != 0ull ? 1u : 0u) << k;
		s_brick_any[w] = bits;
	}
}
__device__ __forceinline__ int mip_from_pos(v3 pos, uint32_t max_cascade = NGP_NERF_CASCADES - 1) {
	float maxval = fmaxf(fmaxf(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	int m = frexp_exponent(maxval) + 1;
	m = m < 0 ? 0 : m;
	return (int)max_cascade < m ? (int)max_cascade : m;
}
__device__ __forceinline__ int mip_from_dt(float dt, v3 pos, uint32_t max_cascade = NGP_NERF_CASCADES - 1) {
	int mip = mip_from_pos(pos, max_cascade);
	dt *= 256.0f;
	if (dt < 1.0f) return mip;
	int e = frexp_exponent(dt);
	int m = e > mip ? e : mip;
	return (int)max_cascade < m ? (int)max_cascade : m;
}

// ---------------------------------------------------------------- colour (common_device.cuh:31-77)
__device__ __forceinline__ float srgb_to_linear(float s) { return s <= 0.04045f ? s / 12.92f : powf((s + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float srgb_to_linear_derivative(float s) { return s <= 0.04045f ? 1.0f / 12.92f : 2.4f / 1.055f * powf((s + 0.055f) / 1.055f, 1.4f); }   // common_device.cuh:43-49
__device__ __forceinline__ float linear_to_srgb(float l) { return l < 0.0031308f ? 12.92f * l : 1.055f * powf(l, 0.41666f) - 0.055f; }

// ---------------------------------------------------------------- activations (testbed_nerf.cu:215-257)
__device__ __forceinline__ float logistic(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float network_to_rgb(float v, int act) {
	switch (act) {
		case NGP_ACT_NONE: return v;
		case NGP_ACT_RELU: return v > 0.0f ? v : 0.0f;
		case NGP_ACT_LOGISTIC: return logistic(v);
		default: return __expf(clampf(v, -10.0f, 10.0f));
	}
}
__device__ __forceinline__ float network_to_rgb_derivative(float v, int act) {
	switch (act) {
		case NGP_ACT_NONE: return 1.0f;
		case NGP_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
		case NGP_ACT_LOGISTIC: { float d = logistic(v); return d * (1 - d); }
		default: return __expf(clampf(v, -10.0f, 10.0f));
	}
}
__device__ __forceinline__ float network_to_density(float v, int act) {
	switch (act) {
		case NGP_ACT_NONE: return v;
		case NGP_ACT_RELU: return v > 0.0f ? v : 0.0f;
		case NGP_ACT_LOGISTIC: return logistic(v);
		default: return __expf(v);
	}
}
__device__ __forceinline__ float network_to_density_derivative(float v, int act) {
	switch (act) {
		case NGP_ACT_NONE: return 1.0f;
		case NGP_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
		case NGP_ACT_LOGISTIC: { float d = logistic(v); return d * (1 - d); }
		default: return __expf(clampf(v, -15.0f, 15.0f));
	}
}

// ---------------------------------------------------------------- fp16 helpers
typedef _Float16 half_t;
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(half_t, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (half_t)f); }

// ---------------------------------------------------------------- wave64 primitives
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// inclusive prefix sum over the 64 lanes of a wave (all lanes must participate)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
	const uint32_t l = lane_id();
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		uint32_t n = __shfl_up(v, off, 64);
		if (l >= (uint32_t)off) v += n;
	}
	return v;
}

// ---------------------------------------------------------------- images (common_device.cuh:80-111, 145-200, 621-709)
__device__ __forceinline__ void read_rgba(float x, float y, const int32_t* res, const void* pixels, int type, float out[4]) {
	int px = (int)(x * (float)res[0]), py = (int)(y * (float)res[1]);
	px = px < res[0] - 1 ? px : res[0] - 1; px = px > 0 ? px : 0;
	py = py < res[1] - 1 ? py : res[1] - 1; py = py > 0 ? py : 0;
	uint64_t idx = (uint64_t)px + (uint64_t)py * (uint64_t)res[0];
	if (type == 1) {
		uint32_t raw = ((const uint32_t*)pixels)[idx];
		if (raw == 0x00FF00FFu) { out[0] = out[1] = out[2] = out[3] = -1.0f; return; }
		float alpha = (float)(raw >> 24) * (1.0f / 255.0f);
		out[0] = srgb_to_linear((float)(raw & 0xff) * (1.0f / 255.0f)) * alpha;
		out[1] = srgb_to_linear((float)((raw >> 8) & 0xff) * (1.0f / 255.0f)) * alpha;
		out[2] = srgb_to_linear((float)((raw >> 16) & 0xff) * (1.0f / 255.0f)) * alpha;
		out[3] = alpha;
	} else if (type == 2) {
		const uint16_t* h = (const uint16_t*)pixels + idx * 4;
		for (int i = 0; i < 4; ++i) out[i] = h2f(h[i]);
	} else if (type == 3) {
		const float* f = (const float*)pixels + idx * 4;
		for (int i = 0; i < 4; ++i) out[i] = f[i];
	} else {
		out[0] = 5.0f; out[1] = 0.0f; out[2] = 0.0f; out[3] = 1.0f;
	}
}
// masked-pixel test only (first channel < 0): avoids the sRGB decode in the ray generator
__device__ __forceinline__ bool pixel_is_masked(float x, float y, const int32_t* res, const void* pixels, int type) {
	int px = (int)(x * (float)res[0]), py = (int)(y * (float)res[1]);
	px = px < res[0] - 1 ? px : res[0] - 1; px = px > 0 ? px : 0;
	py = py < res[1] - 1 ? py : res[1] - 1; py = py > 0 ? py : 0;
	uint64_t idx = (uint64_t)px + (uint64_t)py * (uint64_t)res[0];
	if (type == 1) return ((const uint32_t*)pixels)[idx] == 0x00FF00FFu;
	if (type == 2) return h2f(((const uint16_t*)pixels)[idx * 4]) < 0.0f;
	if (type == 3) return ((const float*)pixels)[idx * 4] < 0.0f;
	return false;
}
__device__ __forceinline__ void read_image2(const float* data, int rx, int ry, float px_, float py_, float& o0, float& o1) {
	float pfx = px_ * (float)(rx - 1), pfy = py_ * (float)(ry - 1);
	int tx = (int)pfx, ty = (int)pfy;
	float wx = pfx - (float)tx, wy = pfy - (float)ty;
	float a0 = 0, a1 = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		int x = tx + (k & 1), y = ty + (k >> 1);
		x = x < rx - 1 ? x : rx - 1; x = x > 0 ? x : 0;
		y = y < ry - 1 ? y : ry - 1; y = y > 0 ? y : 0;
		float w = ((k & 1) ? wx : (1 - wx)) * ((k >> 1) ? wy : (1 - wy));
		const float* v = &data[(x + y * rx) * 2];
		if (k == 0) { a0 = w * v[0]; a1 = w * v[1]; } else { a0 = a0 + w * v[0]; a1 = a1 + w * v[1]; }
	}
	o0 = a0; o1 = a1;
}
__device__ __forceinline__ void opencv_distort(const float* p, float u, float v, float& du, float& dv) {
	const float k1 = p[0], k2 = p[1], p1 = p[2], p2 = p[3];
	const float u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2;
	const float radial = k1 * r2 + k2 * r2 * r2;
	du = u * radial + 2.0f * p1 * uv + p2 * (r2 + 2.0f * u2);
	dv = v * radial + 2.0f * p2 * uv + p1 * (r2 + 2.0f * v2);
}
__device__ __forceinline__ void iterative_opencv_lens_undistortion(const float* params, float& u, float& v) {
	const float kMaxStepNorm = 1e-10f, kRelStepSize = 1e-6f, eps = 1.1920928955078125e-07f;
	const float x00 = u, x01 = v;
	float x0 = u, x1 = v;
	for (uint32_t i = 0; i < 100; ++i) {
		const float step0 = fmaxf(eps, fabsf(kRelStepSize * x0));
		const float step1 = fmaxf(eps, fabsf(kRelStepSize * x1));
		float dx0, dx1, b00, b01, f00, f01, b10, b11, f10, f11;
		opencv_distort(params, x0, x1, dx0, dx1);
		opencv_distort(params, x0 - step0, x1, b00, b01);
		opencv_distort(params, x0 + step0, x1, f00, f01);
		opencv_distort(params, x0, x1 - step1, b10, b11);
		opencv_distort(params, x0, x1 + step1, f10, f11);
		float J00 = 1 + (f00 - b00) / (2 * step0);
		float J01 = (f10 - b10) / (2 * step1);
		float J10 = (f01 - b01) / (2 * step0);
		float J11 = 1 + (f11 - b11) / (2 * step1);
		float det = J00 * J11 - J10 * J01;
		float invdet = 1.0f / det;
		float i00 = J11 * invdet, i01 = -J01 * invdet, i10 = -J10 * invdet, i11 = J00 * invdet;
		float r0 = x0 + dx0 - x00, r1 = x1 + dx1 - x01;
		float s0 = i00 * r0 + i01 * r1, s1 = i10 * r0 + i11 * r1;
		x0 -= s0; x1 -= s1;
		if (s0 * s0 + s1 * s1 < kMaxStepNorm) break;
	}
	u = x0; v = x1;
}

// Eigen Quaternionf(Matrix3f) / slerp / normalized / toRotationMatrix as used by get_xform_given_rolling_shutter (common_device.cuh:223-234)
__device__ __forceinline__ void quat_from_mat(const float* m, float q[4]) {
#define NGP_M(r, c) m[(c) * 3 + (r)]
	float t = NGP_M(0, 0) + NGP_M(1, 1) + NGP_M(2, 2);
	if (t > 0.0f) {
		t = sqrtf(t + 1.0f);
		q[3] = 0.5f * t;
		t = 0.5f / t;
		q[0] = (NGP_M(2, 1) - NGP_M(1, 2)) * t;
		q[1] = (NGP_M(0, 2) - NGP_M(2, 0)) * t;
		q[2] = (NGP_M(1, 0) - NGP_M(0, 1)) * t;
	} else {
		int i = 0;
		if (NGP_M(1, 1) > NGP_M(0, 0)) i = 1;
		if (NGP_M(2, 2) > NGP_M(i, i)) i = 2;
		int j = (i + 1) % 3, k = (j + 1) % 3;
		t = sqrtf(NGP_M(i, i) - NGP_M(j, j) - NGP_M(k, k) + 1.0f);
		float qi = 0.5f * t;
		t = 0.5f / t;
		float qw = (NGP_M(k, j) - NGP_M(j, k)) * t;
		float qj = (NGP_M(j, i) + NGP_M(i, j)) * t;
		float qk = (NGP_M(k, i) + NGP_M(i, k)) * t;
		q[3] = qw;
		// static indexing to keep q in registers
		if (i == 0) { q[0] = qi; q[1] = qj; q[2] = qk; }
		else if (i == 1) { q[1] = qi; q[2] = qj; q[0] = qk; }
		else { q[2] = qi; q[0] = qj; q[1] = qk; }
	}
#undef NGP_M
}
__device__ __forceinline__ void get_xform_given_rolling_shutter(const NgpXForm& xf, const float* rs, float u, float v, float motionblur_time, float out[12]) {
	float pixel_t = rs[0] + rs[1] * u + rs[2] * v + rs[3] * motionblur_time;
	v3 s3 = col(xf.start, 3), e3 = col(xf.end, 3);
	v3 pos = s3 + (e3 - s3) * pixel_t;
	float qa[4], qb[4], q[4];
	quat_from_mat(xf.start, qa);
	quat_from_mat(xf.end, qb);
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
#pragma unroll
	for (int i = 0; i < 4; ++i) q[i] = scale0 * qa[i] + scale1 * qb[i];
	float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
	float n = sqrtf(n2);
#pragma unroll
	for (int i = 0; i < 4; ++i) q[i] = q[i] / n;
	const float x = q[0], y = q[1], z = q[2], w = q[3];
	const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
	const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
	out[0] = 1.0f - (tyy + tzz); out[3] = txy - twz;          out[6] = txz + twy;
	out[1] = txy + twz;          out[4] = 1.0f - (txx + tzz); out[7] = tyz - twx;
	out[2] = txz - twy;          out[5] = tyz + twx;          out[8] = 1.0f - (txx + tyy);
	out[9] = pos.x; out[10] = pos.y; out[11] = pos.z;
}
__device__ __forceinline__ v3 f_theta_undistortion(float uvx, float uvy, const float* params, v3 error_direction) {
	float xpix = uvx * params[5], ypix = uvy * params[6];
	float nrm = sqrtf(xpix * xpix + ypix * ypix);
	float alpha = params[0] + nrm * (params[1] + nrm * (params[2] + nrm * (params[3] + nrm * params[4])));
	float sin_alpha = sinf(alpha), cos_alpha = cosf(alpha);
	if (cos_alpha <= 1.17549435e-38f || nrm == 0.f) return error_direction;
	sin_alpha *= 1.f / nrm;
	return mk(sin_alpha * xpix, sin_alpha * ypix, cos_alpha);
}
__device__ __forceinline__ v3 latlong_to_dir(float u, float v) {
	const float PI = 3.14159265358979323846f;
	float theta = (v - 0.5f) * PI, phi = (u - 0.5f) * PI * 2.0f;
	float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
	return mk(sp * ct, st, cp * ct);
}

// ---------------------------------------------------------------- trainable 2-D buffers: distortion map, environment map
// inverse of the 3x3 block of a column-major 3x4 (Eigen's Matrix3f::inverse(): cofactors over the determinant)
__device__ __forceinline__ void mat3_inverse(const float* m, float* inv /* 9, column-major */) {
	const float a = m[0], b = m[3], c = m[6], d = m[1], e = m[4], f = m[7], g = m[2], h = m[5], k = m[8];
	const float A = e * k - f * h, B = -(d * k - f * g), C = d * h - e * g;
	const float invdet = 1.0f / (a * A + b * B + c * C);
	inv[0] = A * invdet; inv[3] = -(b * k - c * h) * invdet; inv[6] = (b * f - c * e) * invdet;
	inv[1] = B * invdet; inv[4] = (a * k - c * g) * invdet;  inv[7] = -(a * f - c * d) * invdet;
	inv[2] = C * invdet; inv[5] = -(a * h - b * g) * invdet; inv[8] = (a * e - b * d) * invdet;
}
// deposit_image_gradient<2> (common_device.cuh:112-143): value * weight into the gradient image, weight into the weight image
__device__ __forceinline__ void deposit_image_gradient2(float v0, float v1, float* __restrict__ gradient, float* __restrict__ gradient_weight, int rx, int ry, float px, float py) {
	const float fx = px * (float)(rx - 1), fy = py * (float)(ry - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		int x = tx + (k & 1), y = ty + (k >> 1);
		x = x < rx - 1 ? x : rx - 1; x = x > 0 ? x : 0;
		y = y < ry - 1 ? y : ry - 1; y = y > 0 ? y : 0;
		const float w = ((k & 1) ? wx : 1 - wx) * ((k >> 1) ? wy : 1 - wy);
		const size_t o = ((size_t)x + (size_t)y * rx) * 2;
		atomicAdd(&gradient[o], v0 * w); atomicAdd(&gradient_weight[o], w);
		atomicAdd(&gradient[o + 1], v1 * w); atomicAdd(&gradient_weight[o + 1], w);
	}
}
// envmap.cuh:29-63 / 65-103: lat-long lookup of a [h][w][4] fp32 map by direction (x wraps, y clamps); dir_to_spherical_unorm: random_val.cuh:64-69
struct EnvmapTap { int idx[4]; float w[4]; };
__device__ __forceinline__ EnvmapTap envmap_taps(int rx, int ry, v3 dir) {
	const float PI = 3.14159265358979323846f;
	const float dx = dir.z, dy = -dir.x, dz = dir.y;
	const float cos_theta = fminf(fmaxf(dz, -1.0f), 1.0f);
	const float theta = acosf(cos_theta);
	const float phi = atan2f(dy, dx);
	const float cx = theta / PI, cy = phi / (2.0f * PI) + 0.5f;
	const float fx = cy * (float)(rx - 1), fy = cx * (float)(ry - 1);
	const int tx = (int)fx, ty = (int)fy;
	const float wx = fx - (float)tx, wy = fy - (float)ty;
	EnvmapTap t;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		int x = tx + (k & 1), y = ty + (k >> 1);
		if (x < 0) x += rx; else if (x >= rx) x -= rx;
		y = y < ry - 1 ? y : ry - 1; y = y > 0 ? y : 0;
		t.idx[k] = (x + y * rx) * 4;
		t.w[k] = ((k & 1) ? wx : 1 - wx) * ((k >> 1) ? wy : 1 - wy);
	}
	return t;
}
__device__ __forceinline__ void read_envmap(const float* __restrict__ data, int rx, int ry, v3 dir, float out[4]) {
	const EnvmapTap t = envmap_taps(rx, ry, dir);
#pragma unroll
	for (int c = 0; c < 4; ++c) out[c] = ((t.w[0] * data[t.idx[0] + c] + t.w[1] * data[t.idx[1] + c]) + t.w[2] * data[t.idx[2] + c]) + t.w[3] * data[t.idx[3] + c];
}
// hsv_to_rgb / to_rgb (common_device.cuh:594-619): the Distortion render mode's colouring of a 2-D offset
__device__ __forceinline__ v3 hsv_to_rgb(float h, float s, float v) {
	if (s == 0.0f) return mk(v, v, v);
	h = fmodf(h, 1.0f) * 6.0f;
	const int i = (int)h;
	const float f = h - (float)i;
	const float p = v * (1.0f - s), q = v * (1.0f - s * f), t = v * (1.0f - s * (1.0f - f));
	switch (i) {
		case 0: return mk(v, t, p);
		case 1: return mk(q, v, p);
		case 2: return mk(p, v, t);
		case 3: return mk(p, q, v);
		case 4: return mk(t, p, v);
		default: return mk(v, p, q);
	}
}
__device__ __forceinline__ v3 offset_to_rgb(float dx, float dy) {
	return hsv_to_rgb(atan2f(dy, dx) / (2.0f * 3.14159265358979323846f) + 0.5f, 1.0f, sqrtf(dx * dx + dy * dy));
}

} // namespace ngp
