/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/README.md).
 *
 * Plain-C CPU restatement of the blender-ngp NeRF hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this; the
 * product path (blender-ngp_amd/) never links, imports or executes it.
 *
 * The reference cannot be compiled here (CUDA + empty tiny-cuda-nn / Eigen
 * submodules) and ships no golden vectors, so this restatement is checked only
 * against (a) constants / parameter counts citable in the reference tree and
 * (b) the python metric helpers of scripts/common.py (tests/golden).
 *
 * Every function cites the reference file:line it follows.  Pieces that live in
 * the absent tiny-cuda-nn submodule (pcg32, morton, grid/MLP arithmetic, Adam)
 * restate that library's published algorithm and are marked [tcnn].
 * Eigen (absent) arithmetic is restated where it matters for rounding [eigen].
 *
 * Float discipline: compile with -ffp-contract=off; all arithmetic is fp32 in
 * the stated order so the HIP kernels (also built -ffp-contract=off for the
 * integer/index paths) can be compared bit-exactly.
 */
#ifndef ORC_CORE_H
#define ORC_CORE_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* constants: src/testbed_nerf.cu:53-73, include/.../nerf.h:24-26       */
/* ------------------------------------------------------------------ */
#define ORC_NERF_GRIDSIZE 128u
#define ORC_NERF_GRID_N_CELLS (128u * 128u * 128u)
#define ORC_NERF_STEPS 1024u
#define ORC_NERF_CASCADES 8u
#define ORC_SQRT3 1.73205080757f
#define ORC_STEPSIZE (ORC_SQRT3 / (float)ORC_NERF_STEPS)
#define ORC_MIN_CONE_STEPSIZE ORC_STEPSIZE
#define ORC_MAX_CONE_STEPSIZE (ORC_STEPSIZE * (float)(1u << (ORC_NERF_CASCADES - 1)) * (float)ORC_NERF_STEPS / (float)ORC_NERF_GRIDSIZE)
#define ORC_N_MAX_RANDOM_SAMPLES_PER_RAY 8u
#define ORC_NERF_MIN_OPTICAL_THICKNESS 0.01f

typedef struct { float x, y, z; } orc_vec3;
typedef struct { orc_vec3 min, max; } orc_aabb;

/* nerf.h:62-107 NerfCoordinate: pos(3) dt dir(3) = 7 floats */
typedef struct { float pos[3]; float dt; float dir[3]; } orc_coord;

/* common.h:169-172 */
typedef struct { orc_vec3 o, d; } orc_ray;

/* common.h:174-177 TrainingXForm: two 3x4 column-major matrices [eigen] */
typedef struct { float start[12]; float end[12]; } orc_xform;

/* nerf_loader.h:30-45 TrainingImageMetadata (fields used on the hot path) */
typedef struct {
	const void* pixels;       /* RGBA8 (type 1), half4 (2) or float4 (3) */
	int32_t image_data_type;  /* common_device.cuh:621-626 EImageDataType */
	int32_t res[2];
	float focal_length[2];
	float principal_point[2];
	float rolling_shutter[4];
	int32_t lens_mode;        /* common.h:179-184: 0 perspective, 1 opencv, 2 ftheta, 3 latlong */
	float lens_params[7];
	const float* depth;
	const orc_ray* rays;
} orc_image_meta;

/* nerf.h:28-36 NerfPayload */
typedef struct {
	orc_vec3 origin;
	orc_vec3 dir;
	float t;
	float max_weight;
	uint32_t idx;
	uint16_t n_steps;
	uint8_t alive;
	uint8_t pad_;
} orc_payload;

/* ---- Blender multi-NeRF renderer records (same byte layout as include/ngp_hip.h so that test buffers can be shared) ---- */
typedef struct { float origin[3]; float dir[3]; float rgba[4]; uint32_t idx; float depth; uint8_t alive; uint8_t pad_[3]; } orc_global_ray;     /* render_data_workspace.cuh:13-20 */
typedef struct { float origin[3]; float dir[3]; float t; uint32_t idx; uint16_t n_steps; uint8_t alive; uint8_t active; float mask_alpha; } orc_proxy_ray; /* :22-31 */
typedef struct { int32_t mode; int32_t shape; float transform[16], itransform[16]; float config[6]; float feather, opacity; } orc_mask3d;       /* mask_3D.cuh:129-137 */
typedef struct {                                                                                                                               /* nerf_props.cuh:14-30 */
	float transform[16], itransform[16];
	const uint8_t* density_grid_bitfield;
	uint32_t grid_size, grid_volume;
	orc_aabb render_aabb, train_aabb;
	const orc_mask3d* masks;
	uint32_t n_masks;
	float cone_angle, min_cone_stepsize, max_cone_stepsize;
	uint32_t nerf_cascades;
	float opacity;
} orc_nerf_props;
typedef struct { int32_t max_res[2], scaled_res[2], skip[2]; uint32_t max_pixels, scaled_pixels; } orc_downsample_info;                          /* common.h:300-355 */
typedef struct {                                                                                                                               /* render_request.cuh:55-103 */
	float transform[12]; int32_t model; float focal_length; float sq_width, sq_height, sq_curvature; float qh_front[12], qh_back[12];
	float near_distance, aperture_size, focus_z;
} orc_render_camera;

/* ------------------------------------------------------------------ */
/* fp16 <-> fp32 (IEEE binary16, round-to-nearest-even)                */
/* ------------------------------------------------------------------ */
static inline uint32_t orc_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float orc_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline uint16_t orc_f2h(float f) {
	uint32_t x = orc_f2u(f);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t absx = x & 0x7fffffffu;
	if (absx >= 0x7f800000u) { /* inf / nan */
		return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0u));
	}
	if (absx >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
		return (uint16_t)(sign | 0x7c00u);
	}
	if (absx < 0x38800000u) { /* subnormal half or zero */
		if (absx < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
		uint32_t e = absx >> 23;
		uint32_t mant = (absx & 0x7fffffu) | 0x800000u;
		uint32_t shift = 126u - e; /* 14..24 */
		uint32_t h = mant >> shift;
		uint32_t rem = mant & ((1u << shift) - 1u);
		uint32_t half = 1u << (shift - 1);
		if (rem > half || (rem == half && (h & 1u))) h++;
		return (uint16_t)(sign | h);
	}
	uint32_t h = ((absx - 0x38000000u) >> 13);
	uint32_t rem = absx & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
	return (uint16_t)(sign | h);
}

static inline float orc_h2f(uint16_t h) {
	uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 0x1fu;
	uint32_t m = h & 0x3ffu;
	if (e == 0) {
		if (m == 0) return orc_u2f(sign);
		/* subnormal: value = m * 2^-24 */
		float v = (float)m * 5.9604644775390625e-08f;
		return sign ? -v : v;
	}
	if (e == 31) return orc_u2f(sign | 0x7f800000u | (m << 13));
	return orc_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* round an fp32 value through fp16 */
static inline float orc_rh(float f) { return orc_h2f(orc_f2h(f)); }

/* ------------------------------------------------------------------ */
/* pcg32 [tcnn: tiny-cuda-nn/include/tiny-cuda-nn/common? pcg32.h]      */
/* PCG32 by M. O'Neill / W. Jakob's pcg32.h, the generator tcnn vendors. */
/* call sites: random_val.cuh:28-45, testbed_nerf.cu:1121, 1379, 470    */
/* ------------------------------------------------------------------ */
#define ORC_PCG32_DEFAULT_STATE 0x853c49e6748fea9bULL
#define ORC_PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define ORC_PCG32_MULT 0x5851f42d4c957f2dULL

typedef struct { uint64_t state, inc; } orc_pcg32;

static inline uint32_t orc_pcg32_next_uint(orc_pcg32* r) {
	uint64_t oldstate = r->state;
	r->state = oldstate * ORC_PCG32_MULT + r->inc;
	uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
	uint32_t rot = (uint32_t)(oldstate >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

static inline void orc_pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq) {
	r->state = 0u;
	r->inc = (initseq << 1u) | 1u;
	orc_pcg32_next_uint(r);
	r->state += initstate;
	orc_pcg32_next_uint(r);
}

/* default_rng_t{seed}: pcg32(initstate = seed, initseq = 1) [tcnn pcg32.h ctor defaults] */
static inline orc_pcg32 orc_pcg32_make(uint64_t seed) {
	orc_pcg32 r;
	orc_pcg32_seed(&r, seed, 1u);
	return r;
}

static inline float orc_pcg32_next_float(orc_pcg32* r) {
	uint32_t u = (orc_pcg32_next_uint(r) >> 9) | 0x3f800000u;
	return orc_u2f(u) - 1.0f;
}

/* O(log delta) LCG skip-ahead (Brown, "Random Number Generation with Arbitrary Stride") */
static inline void orc_pcg32_advance(orc_pcg32* r, int64_t delta_) {
	uint64_t cur_mult = ORC_PCG32_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
	uint64_t delta = (uint64_t)delta_;
	while (delta > 0) {
		if (delta & 1) {
			acc_mult *= cur_mult;
			acc_plus = acc_plus * cur_mult + cur_plus;
		}
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta /= 2;
	}
	r->state = acc_mult * r->state + acc_plus;
}
/* rng.advance() with no argument skips 2^32 draws [tcnn pcg32.h: advance(int64_t delta_ = (1ll<<32))] */
#define ORC_PCG32_DEFAULT_ADVANCE (1ll << 32)

/* ------------------------------------------------------------------ */
/* LK-scrambled Sobol: random_val.cuh:148-288                          */
/* ------------------------------------------------------------------ */
uint32_t orc_sobol(uint32_t index, uint32_t dim);
/* lens models shared by the training sampler and the renderer (common_device.cuh:145-200); defined in orc_sampling.c */
orc_vec3 orc_f_theta_undistortion(float uvx, float uvy, const float* params, orc_vec3 error_direction);
orc_vec3 orc_latlong_to_dir(float u, float v);
static inline uint32_t orc_hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
static inline uint32_t orc_reverse_bits(uint32_t x) {
	x = (((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1));
	x = (((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2));
	x = (((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4));
	x = (((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8));
	return ((x >> 16) | (x << 16));
}
static inline uint32_t orc_lk_perm(uint32_t x, uint32_t seed) {
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
static inline uint32_t orc_nus_base2(uint32_t x, uint32_t seed) {
	x = orc_reverse_bits(x);
	x = orc_lk_perm(x, seed);
	x = orc_reverse_bits(x);
	return x;
}
/* random_val.cuh:273-277 */
static inline float orc_ld_random_val(uint32_t index, uint32_t seed, uint32_t dim) {
	const float S = (float)(1.0 / 4294967296.0);
	index = orc_nus_base2(index, seed);
	return (float)orc_nus_base2(orc_sobol(index, dim), orc_hash_combine(seed, dim)) * S;
}
/* random_val.cuh:267-271 */
static inline void orc_ld_random_val_2d(uint32_t index, uint32_t seed, float out[2]) {
	const float S = (float)(1.0 / 4294967296.0);
	index = orc_nus_base2(index, seed);
	for (uint32_t i = 0; i < 2; ++i) {
		out[i] = (float)orc_nus_base2(orc_sobol(index, i), orc_hash_combine(seed, i)) * S;
	}
}
/* random_val.cuh:317-322; fractf(x) = x - floorf(x) */
static inline void orc_ld_random_pixel_offset(uint32_t spp, float out[2]) {
	float a[2], b[2];
	orc_ld_random_val_2d(0, 0xdeadbeefu, a);
	orc_ld_random_val_2d(spp, 0xdeadbeefu, b);
	for (int i = 0; i < 2; ++i) {
		float o = 0.5f - a[i] + b[i];
		out[i] = o - floorf(o);
	}
}

/* ------------------------------------------------------------------ */
/* morton [tcnn common_device.h: expand_bits / morton3D / morton3D_invert] */
/* call sites: testbed_nerf.cu:330, 381-383, 605-609                    */
/* ------------------------------------------------------------------ */
static inline uint32_t orc_expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
static inline uint32_t orc_morton3D(uint32_t x, uint32_t y, uint32_t z) {
	return orc_expand_bits(x) | (orc_expand_bits(y) << 1) | (orc_expand_bits(z) << 2);
}
static inline uint32_t orc_morton3D_invert(uint32_t x) {
	x = x & 0x49249249u;
	x = (x | (x >> 2)) & 0xc30c30c3u;
	x = (x | (x >> 4)) & 0x0f00f00fu;
	x = (x | (x >> 8)) & 0xff0000ffu;
	x = (x | (x >> 16)) & 0x0000ffffu;
	return x;
}

/* ------------------------------------------------------------------ */
/* vec helpers [eigen fixed-size evaluation order: left to right]       */
/* ------------------------------------------------------------------ */
static inline orc_vec3 orc_v3(float x, float y, float z) { orc_vec3 v = {x, y, z}; return v; }
static inline orc_vec3 orc_add(orc_vec3 a, orc_vec3 b) { return orc_v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline orc_vec3 orc_sub(orc_vec3 a, orc_vec3 b) { return orc_v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline orc_vec3 orc_scale(orc_vec3 a, float s) { return orc_v3(a.x * s, a.y * s, a.z * s); }
static inline float orc_dot(orc_vec3 a, orc_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float orc_norm(orc_vec3 a) { return sqrtf(orc_dot(a, a)); }
/* [eigen] normalized(): z = squaredNorm; z > 0 ? v / sqrt(z) : v */
static inline orc_vec3 orc_normalized(orc_vec3 a) {
	float z = orc_dot(a, a);
	if (z > 0.0f) { float n = sqrtf(z); return orc_v3(a.x / n, a.y / n, a.z / n); }
	return a;
}
/* 3x4 column-major: col c = m[3c..3c+2] */
static inline orc_vec3 orc_col(const float* m, int c) { return orc_v3(m[3 * c], m[3 * c + 1], m[3 * c + 2]); }
/* [eigen] 3x3 * vec3 (lazy product: row dot products, left to right) */
static inline orc_vec3 orc_mat3_mul(const float* m, orc_vec3 v) {
	return orc_v3(
		m[0] * v.x + m[3] * v.y + m[6] * v.z,
		m[1] * v.x + m[4] * v.y + m[7] * v.z,
		m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
static inline float orc_clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); } /* [tcnn] clamp = max(lo, min(hi, v)) — same result for lo<=hi */
static inline int orc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float orc_signf(float x) { return copysignf(1.0f, x); } /* common.h:192-194 */
static inline float orc_logistic(float x) { return 1.0f / (1.0f + expf(-x)); } /* [tcnn] logistic */

/* ------------------------------------------------------------------ */
/* BoundingBox: bounding_box.cuh:86-88, 163-210, 216-221               */
/* ------------------------------------------------------------------ */
static inline int orc_aabb_contains(const orc_aabb* b, orc_vec3 p) {
	return p.x >= b->min.x && p.x <= b->max.x && p.y >= b->min.y && p.y <= b->max.y && p.z >= b->min.z && p.z <= b->max.z;
}
static inline void orc_aabb_ray_intersect(const orc_aabb* b, orc_vec3 pos, orc_vec3 dir, float out[2]) {
	float tmin = (b->min.x - pos.x) / dir.x;
	float tmax = (b->max.x - pos.x) / dir.x;
	if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
	float tymin = (b->min.y - pos.y) / dir.y;
	float tymax = (b->max.y - pos.y) / dir.y;
	if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
	if (tmin > tymax || tymin > tmax) { out[0] = out[1] = 3.402823466e+38f; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (b->min.z - pos.z) / dir.z;
	float tzmax = (b->max.z - pos.z) / dir.z;
	if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
	if (tmin > tzmax || tzmin > tmax) { out[0] = out[1] = 3.402823466e+38f; return; }
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	out[0] = tmin; out[1] = tmax;
}
static inline orc_vec3 orc_aabb_relative_pos(const orc_aabb* b, orc_vec3 p) {
	return orc_v3((p.x - b->min.x) / (b->max.x - b->min.x), (p.y - b->min.y) / (b->max.y - b->min.y), (p.z - b->min.z) / (b->max.z - b->min.z));
}

/* ------------------------------------------------------------------ */
/* marching helpers: testbed_nerf.cu:83-98, 191-213, 267-342, 449-463   */
/* ------------------------------------------------------------------ */
static inline uint32_t orc_grid_mip_offset(uint32_t mip) { return ORC_NERF_GRID_N_CELLS * mip; }
static inline float orc_calc_dt(float t, float cone_angle) { return orc_clampf(t * cone_angle, ORC_MIN_CONE_STEPSIZE, ORC_MAX_CONE_STEPSIZE); }

static inline float orc_distance_to_next_voxel(orc_vec3 pos, orc_vec3 dir, orc_vec3 idir, uint32_t res) {
	float r = (float)res;
	orc_vec3 p = orc_v3(r * pos.x, r * pos.y, r * pos.z);
	float tx = (floorf(p.x + 0.5f + 0.5f * orc_signf(dir.x)) - p.x) * idir.x;
	float ty = (floorf(p.y + 0.5f + 0.5f * orc_signf(dir.y)) - p.y) * idir.y;
	float tz = (floorf(p.z + 0.5f + 0.5f * orc_signf(dir.z)) - p.z) * idir.z;
	float t = fminf(fminf(tx, ty), tz);
	return fmaxf(t / r, 0.0f);
}
static inline float orc_advance_to_next_voxel(float t, float cone_angle, orc_vec3 pos, orc_vec3 dir, orc_vec3 idir, uint32_t res) {
	float t_target = t + orc_distance_to_next_voxel(pos, dir, idir, res);
	do {
		t += orc_calc_dt(t, cone_angle);
	} while (t < t_target);
	return t;
}
static inline float orc_warp_dt(float dt) {
	float max_stepsize = ORC_MIN_CONE_STEPSIZE * (float)(1u << (ORC_NERF_CASCADES - 1));
	return (dt - ORC_MIN_CONE_STEPSIZE) / (max_stepsize - ORC_MIN_CONE_STEPSIZE);
}
static inline float orc_unwarp_dt(float dt) {
	float max_stepsize = ORC_MIN_CONE_STEPSIZE * (float)(1u << (ORC_NERF_CASCADES - 1));
	return dt * (max_stepsize - ORC_MIN_CONE_STEPSIZE) + ORC_MIN_CONE_STEPSIZE;
}
static inline orc_vec3 orc_warp_direction(orc_vec3 d) { return orc_v3((d.x + 1.0f) * 0.5f, (d.y + 1.0f) * 0.5f, (d.z + 1.0f) * 0.5f); }
static inline orc_vec3 orc_unwarp_position(orc_vec3 p, const orc_aabb* b) {
	return orc_v3(b->min.x + p.x * (b->max.x - b->min.x), b->min.y + p.y * (b->max.y - b->min.y), b->min.z + p.z * (b->max.z - b->min.z));
}
static inline uint32_t orc_cascaded_grid_idx_at(orc_vec3 pos, uint32_t mip) {
	float mip_scale = ldexpf(1.0f, -(int)mip);
	pos.x -= 0.5f; pos.y -= 0.5f; pos.z -= 0.5f;
	pos.x *= mip_scale; pos.y *= mip_scale; pos.z *= mip_scale;
	pos.x += 0.5f; pos.y += 0.5f; pos.z += 0.5f;
	int ix = (int)(pos.x * (float)ORC_NERF_GRIDSIZE);
	int iy = (int)(pos.y * (float)ORC_NERF_GRIDSIZE);
	int iz = (int)(pos.z * (float)ORC_NERF_GRIDSIZE);
	return orc_morton3D((uint32_t)orc_clampi(ix, 0, (int)ORC_NERF_GRIDSIZE - 1), (uint32_t)orc_clampi(iy, 0, (int)ORC_NERF_GRIDSIZE - 1), (uint32_t)orc_clampi(iz, 0, (int)ORC_NERF_GRIDSIZE - 1));
}
static inline int orc_density_grid_occupied_at(orc_vec3 pos, const uint8_t* bitfield, uint32_t mip) {
	uint32_t idx = orc_cascaded_grid_idx_at(pos, mip);
	return bitfield[idx / 8 + orc_grid_mip_offset(mip) / 8] & (1u << (idx % 8));
}
static inline int orc_mip_from_pos(orc_vec3 pos, uint32_t max_cascade) {
	int exponent;
	float maxval = fmaxf(fmaxf(fabsf(pos.x - 0.5f), fabsf(pos.y - 0.5f)), fabsf(pos.z - 0.5f));
	frexpf(maxval, &exponent);
	int m = exponent + 1; if (m < 0) m = 0;
	return (int)max_cascade < m ? (int)max_cascade : m;
}
static inline int orc_mip_from_dt(float dt, orc_vec3 pos, uint32_t max_cascade) {
	int mip = orc_mip_from_pos(pos, max_cascade);
	dt *= 2.0f * (float)ORC_NERF_GRIDSIZE;
	if (dt < 1.0f) return mip;
	int exponent;
	frexpf(dt, &exponent);
	int m = exponent > mip ? exponent : mip;
	return (int)max_cascade < m ? (int)max_cascade : m;
}

/* ------------------------------------------------------------------ */
/* colour: common_device.cuh:31-77                                      */
/* ------------------------------------------------------------------ */
static inline float orc_srgb_to_linear(float s) { return s <= 0.04045f ? s / 12.92f : powf((s + 0.055f) / 1.055f, 2.4f); }
static inline float orc_linear_to_srgb(float l) { return l < 0.0031308f ? 12.92f * l : 1.055f * powf(l, 0.41666f) - 0.055f; }
static inline float orc_srgb_to_linear_derivative(float s) { return s <= 0.04045f ? 1.0f / 12.92f : 2.4f / 1.055f * powf((s + 0.055f) / 1.055f, 1.4f); }

/* activations: testbed_nerf.cu:215-257 (Exponential uses __expf; restated with expf) */
enum { ORC_ACT_NONE = 0, ORC_ACT_RELU = 1, ORC_ACT_LOGISTIC = 2, ORC_ACT_EXPONENTIAL = 3 };
static inline float orc_network_to_rgb(float val, int act) {
	switch (act) {
		case ORC_ACT_NONE: return val;
		case ORC_ACT_RELU: return val > 0.0f ? val : 0.0f;
		case ORC_ACT_LOGISTIC: return orc_logistic(val);
		default: return expf(orc_clampf(val, -10.0f, 10.0f));
	}
}
static inline float orc_network_to_rgb_derivative(float val, int act) {
	switch (act) {
		case ORC_ACT_NONE: return 1.0f;
		case ORC_ACT_RELU: return val > 0.0f ? 1.0f : 0.0f;
		case ORC_ACT_LOGISTIC: { float d = orc_logistic(val); return d * (1 - d); }
		default: return expf(orc_clampf(val, -10.0f, 10.0f));
	}
}
static inline float orc_network_to_density(float val, int act) {
	switch (act) {
		case ORC_ACT_NONE: return val;
		case ORC_ACT_RELU: return val > 0.0f ? val : 0.0f;
		case ORC_ACT_LOGISTIC: return orc_logistic(val);
		default: return expf(val);
	}
}
static inline float orc_network_to_density_derivative(float val, int act) {
	switch (act) {
		case ORC_ACT_NONE: return 1.0f;
		case ORC_ACT_RELU: return val > 0.0f ? 1.0f : 0.0f;
		case ORC_ACT_LOGISTIC: { float d = orc_logistic(val); return d * (1 - d); }
		default: return expf(orc_clampf(val, -15.0f, 15.0f));
	}
}

#ifdef __cplusplus
}
#endif
/* error-map CDFs (testbed_nerf.cu:3243-3245: each pointer may be NULL on its own) */
typedef struct { const float* cdf_x_cond_y; const float* cdf_y; const float* cdf_img; int32_t res[2]; } orc_error_map_cdf;

#endif
