// train_samples.hip — training ray generation + occupancy-grid marching for gfx950.
// Replaces src/testbed_nerf.cu:1085-1260 (generate_training_samples_nerf) incl. image_idx (1062-1083),
// nerf_random_image_pos_training (1047-1060) and the camera models of common_device.cuh:145-258.
//
// MI355X notes: the reference reserves output slots with two global atomicAdd's PER RAY (1225, 1232).  Here the 64 lanes
// of a wave reconverge after the counting pass, build an exclusive prefix over their step counts with wave shuffles and the
// kept-ray ballot, and one lane issues ONE atomic per counter per wave.  The per-ray results (numsteps, coords) are
// identical to the reference's; only the slot order differs (it is unordered in the reference as well).
// Index / count arithmetic is exact (-ffp-contract=off): bit-exact against the CPU oracle.
#include "ngp_device.cuh"
#include "ngp_dev_knobs.h"
#include <stdlib.h>

namespace ngp {

// The wave-per-ray marches run BESIDE the backward chain: every register they hold is one the chain's waves cannot have (a march at 137 registers took the binning
// kernels beside it from 35 + 67 us to 64 + 75).  Minimum waves per SIMD the compiler must leave room for = the register cap (4 -> 128, 5 -> 102, 6 -> 85).
#ifndef NGP_MARCH_WAVES_PER_EU
#define NGP_MARCH_WAVES_PER_EU 4
#endif

// occupied runs a ray may have before the write pass falls back to a serial re-march (LDS: 6 B x 256 threads per run slot)
#define NGP_MAX_RUNS 24

struct TrainSampleArgs {
	uint32_t n_rays; Aabb aabb; uint32_t max_samples; Pcg32 rng;
	uint32_t* ray_counter; uint32_t* numsteps_counter; uint32_t* ray_indices_out; NgpRay* rays_out; uint32_t* numsteps_out; NgpCoord* coords_out;
	uint32_t n_training_images; const NgpImageMeta* metadata; const NgpXForm* xforms; const uint8_t* density_grid;
	int max_level_rand_training; float* max_level_ptr; int snap_to_pixel_centers; int train_envmap; float cone_angle_constant;
	const float* distortion_data; int32_t distortion_res[2]; uint32_t ray_offset; uint32_t n_rays_global; ErrorMapCdf cdf;
	const uint32_t* brick_summary;   // optional precomputed s_brick_any
	int dev_variant; // dev-only timing variants (NGP_HIP_GEN_VARIANT): 0 product path, 2 no sample writes, 3 no march
};


// Per-ray setup of generate_training_samples_nerf (testbed_nerf.cu:1107-1200): image, pixel, camera ray, box entry, jittered start.  One copy for the three march kernels.
struct TrainRaySetup { bool pixel_ok; float startt, max_level; v3 ro, rd_unnorm, rd, idir; };
__device__ __forceinline__ TrainRaySetup setup_training_ray(const TrainSampleArgs& a, uint32_t i) {
	TrainRaySetup r;
	r.pixel_ok = false; r.startt = 0.f; r.max_level = 1.0f;
	r.ro = mk(0, 0, 0); r.rd_unnorm = mk(0, 0, 1); r.rd = mk(0, 0, 1); r.idir = mk(1, 1, 1);
	const uint32_t img = image_idx(i, a.n_rays_global, a.n_training_images, a.cdf.cdf_img, nullptr);
	const NgpImageMeta& md = a.metadata[img];
	Pcg32 rng = a.rng;
	rng.advance((uint64_t)(uint32_t)(i * NGP_N_MAX_RANDOM_SAMPLES_PER_RAY));
	float u, v;
	nerf_random_image_pos_training(rng, md.res, a.snap_to_pixel_centers, a.cdf, img, u, v, nullptr);
	if (pixel_is_masked(u, v, md.res, md.pixels, md.image_data_type)) return r;
	r.pixel_ok = true;
	r.max_level = a.max_level_rand_training ? (rng.next_float() * 2.0f) : 1.0f;
	const float motionblur_time = rng.next_float();
	float xform[12];
	get_xform_given_rolling_shutter(a.xforms[img], md.rolling_shutter, u, v, motionblur_time, xform);
	if (md.rays) {
		int px = (int)(u * (float)md.res[0]), py = (int)(v * (float)md.res[1]);
		px = px < md.res[0] - 1 ? px : md.res[0] - 1; px = px > 0 ? px : 0;
		py = py < md.res[1] - 1 ? py : md.res[1] - 1; py = py > 0 ? py : 0;
		const NgpRay ray = md.rays[(uint64_t)px + (uint64_t)py * (uint64_t)md.res[0]];
		r.ro = ld3(ray.o); r.rd_unnorm = ld3(ray.d);
	} else {
		r.ro = col(xform, 3);
		v3 d;
		if (md.lens_mode == 2) d = f_theta_undistortion(u - md.principal_point[0], v - md.principal_point[1], md.lens_params, mk(0.f, 0.f, 1.f));
		else if (md.lens_mode == 3) d = latlong_to_dir(u, v);
		else {
			d = mk((u - md.principal_point[0]) * (float)md.res[0] / md.focal_length[0], (v - md.principal_point[1]) * (float)md.res[1] / md.focal_length[1], 1.0f);
			if (md.lens_mode == 1) iterative_opencv_lens_undistortion(md.lens_params, d.x, d.y);
		}
		if (a.distortion_data) {
			float o0, o1;
			read_image2(a.distortion_data, a.distortion_res[0], a.distortion_res[1], u, v, o0, o1);
			d.x += o0; d.y += o1;
		}
		r.rd_unnorm = mat3_mul(xform, d); // NOT normalized (1189)
	}
	r.rd = normalized(r.rd_unnorm);
	float tmin, tmax;
	aabb_ray_intersect(a.aabb, r.ro, r.rd, tmin, tmax);
	tmin = fmaxf(tmin, 0.0f);
	r.startt = tmin;
	r.startt += calc_dt(r.startt, a.cone_angle_constant) * rng.next_float();   // cone_angle 0: clamp(+-0, MIN, MAX) = MIN_CONE_STEPSIZE
	r.idir = mk(1.0f / r.rd.x, 1.0f / r.rd.y, 1.0f / r.rd.z);
	return r;
}

template <bool CONST_DT>
__global__ void __launch_bounds__(256) generate_training_samples_kernel(const TrainSampleArgs a) {
	const uint32_t li = threadIdx.x + blockIdx.x * blockDim.x;
	const bool in_range = li < a.n_rays;
	const uint32_t i = li + a.ray_offset;

	constexpr uint32_t MAX_RUNS = NGP_MAX_RUNS;
	__shared__ float s_run_t[MAX_RUNS][256];
	__shared__ uint16_t s_run_n[MAX_RUNS][256];
	__shared__ uint32_t s_brick_any[NGP_NERF_GRID_N_CELLS / 64 / 32];
	if (a.brick_summary) { for (uint32_t q = threadIdx.x; q < NGP_NERF_GRID_N_CELLS / 64 / 32; q += blockDim.x) s_brick_any[q] = a.brick_summary[q]; }
	else load_brick_summary(a.density_grid, s_brick_any);
	__syncthreads();
	uint32_t n_runs = 0;

	// ---- per-ray setup (dead lanes keep numsteps = 0 and take part in the wave scan)
	uint32_t numsteps = 0;
	bool keep = false;
	float startt = 0.f, cone_angle = a.cone_angle_constant, max_level = 1.0f;
	v3 ro = mk(0, 0, 0), rd_unnorm = mk(0, 0, 1), rd = mk(0, 0, 1), idir = mk(1, 1, 1);

	if (in_range) {
		const TrainRaySetup rs = setup_training_ray(a, i);
		if (rs.pixel_ok) {
			max_level = rs.max_level; ro = rs.ro; rd_unnorm = rs.rd_unnorm; rd = rs.rd; idir = rs.idir; startt = rs.startt;

			// pass 1: count occupied steps (1204-1219).  The march is a serial, latency-bound chain (one dependent bitfield byte per
			// iteration); instead of repeating it for the write pass, the occupied RUNS (start t, length) are recorded in LDS: inside a
			// run the reference's second pass is exactly `emit at t; t += dt`, so replaying the runs reproduces it bit for bit.
			uint32_t j = 0;
			float t = startt;
			v3 pos;
			if (a.dev_variant == 3) t = 3.0e38f;
			uint32_t run_len = 0;   // length of the open run (registers only; LDS is touched once per run)
			float run_t0 = 0.f;
			OccBrick occ;
			while (aabb_contains(a.aabb, pos = ro + rd * t) && j < NGP_NERF_STEPS) {
				const float dt = calc_dt_t<CONST_DT>(t, cone_angle);
				const uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
				if (density_grid_occupied_at(pos, a.density_grid, mip, occ, s_brick_any)) {
					if (run_len == 0) run_t0 = t;
					++run_len;
					++j; t += dt;
					if (CONST_DT && mip == 0) {
						// The next samples mostly stay in this cell (a cell is ~4.6 minimum steps across) and then need no new decision: same
						// cell => same occupancy bit, inside the box, mip 0.  Membership is tested on the reference's own quantities — the
						// position it would compute, re-centred like cascaded_grid_idx_at ((p - 0.5) + 0.5, mip scale 1); q * 128 is exact, so
						// int(q * 128) == c  <=>  c/128 <= q < (c+1)/128 — hence the same samples, bit for bit, at ~1/6 of the instructions.
						// Only for cells strictly inside the box (no clamped index, no sample on the box faces where mip_from_pos turns 1).
						const float qx = (pos.x - 0.5f) + 0.5f, qy = (pos.y - 0.5f) + 0.5f, qz = (pos.z - 0.5f) + 0.5f;
						const float lx = (float)(int)(qx * 128.0f) * 0.0078125f, ly = (float)(int)(qy * 128.0f) * 0.0078125f, lz = (float)(int)(qz * 128.0f) * 0.0078125f;
						const float hx = lx + 0.0078125f, hy = ly + 0.0078125f, hz = lz + 0.0078125f;
						const float m = 1e-6f;
						const bool interior = lx - a.aabb.mn.x > m && ly - a.aabb.mn.y > m && lz - a.aabb.mn.z > m && a.aabb.mx.x - hx > m && a.aabb.mx.y - hy > m && a.aabb.mx.z - hz > m &&
						                      lx > 0.0f && ly > 0.0f && lz > 0.0f && hx < 1.0f && hy < 1.0f && hz < 1.0f;
						if (interior) {
							while (j < NGP_NERF_STEPS) {
								const v3 pn = ro + rd * t;
								const float nx = (pn.x - 0.5f) + 0.5f, ny = (pn.y - 0.5f) + 0.5f, nz = (pn.z - 0.5f) + 0.5f;
								if (!(nx >= lx && nx < hx && ny >= ly && ny < hy && nz >= lz && nz < hz)) break;
								++run_len;
								++j; t += dt;
							}
						}
					}
				} else {
					if (run_len) {
						if (n_runs < MAX_RUNS) { s_run_t[n_runs][threadIdx.x] = run_t0; s_run_n[n_runs][threadIdx.x] = (uint16_t)run_len; }
						++n_runs;
						run_len = 0;
					}
					t = advance_to_next_voxel<CONST_DT>(t, cone_angle, pos, rd, idir, NGP_NERF_GRIDSIZE >> mip);
				}
			}
			if (run_len) {
				if (n_runs < MAX_RUNS) { s_run_t[n_runs][threadIdx.x] = run_t0; s_run_n[n_runs][threadIdx.x] = (uint16_t)run_len; }
				++n_runs;
			}
			numsteps = j;
			keep = !(j == 0 && !a.train_envmap);
			if (!keep) numsteps = 0;
		}
	}

	// ---- wave-aggregated slot reservation: one atomic per counter per wave
	const uint32_t lane = lane_id();
	const uint32_t incl = wave_inclusive_scan(numsteps);
	const uint32_t wave_total = __shfl(incl, 63, 64);
	uint32_t wave_base = 0;
	if (lane == 63 && wave_total) wave_base = atomicAdd(a.numsteps_counter, wave_total);
	wave_base = __shfl(wave_base, 63, 64);
	const uint32_t base = wave_base + incl - numsteps;
	// a ray whose run would overflow is dropped AFTER the counter was bumped (1225-1228)
	if (keep && base + numsteps > a.max_samples) keep = false;
	const unsigned long long kept_mask = __ballot(keep);
	const uint32_t n_kept = (uint32_t)__popcll(kept_mask);
	uint32_t ray_base = 0;
	if (lane == 0 && n_kept) ray_base = atomicAdd(a.ray_counter, n_kept);
	ray_base = __shfl(ray_base, 0, 64);
	if (!keep) return;
	const uint32_t ray_idx = ray_base + (uint32_t)__popcll(kept_mask & ((1ull << lane) - 1ull));

	a.ray_indices_out[ray_idx] = i;
	NgpRay ray_out; ray_out.o[0] = ro.x; ray_out.o[1] = ro.y; ray_out.o[2] = ro.z; ray_out.d[0] = rd_unnorm.x; ray_out.d[1] = rd_unnorm.y; ray_out.d[2] = rd_unnorm.z;
	a.rays_out[ray_idx] = ray_out;
	a.numsteps_out[ray_idx * 2 + 0] = numsteps;
	a.numsteps_out[ray_idx * 2 + 1] = base;

	// pass 2 is a separate, wave-per-ray kernel (expand_training_samples_kernel): this thread only leaves the recipe — start t,
	// number of runs and the (t0, length) pairs — in the first bytes of the ray's own output slots, which that kernel reads
	// before it overwrites them.  Every run has >= 1 sample, so the 28 B x numsteps of slots always hold 8 B + 8 B x n_runs.
	if (numsteps == 0) return;
	uint32_t* stash = (uint32_t*)(a.coords_out + base);
	stash[0] = n_runs;
	stash[1] = __float_as_uint(startt);
	const uint32_t n_store = n_runs <= MAX_RUNS ? n_runs : 0;
	for (uint32_t r = 0; r < n_store; ++r) {
		stash[2 + 2 * r] = __float_as_uint(s_run_t[r][threadIdx.x]);
		stash[3 + 2 * r] = s_run_n[r][threadIdx.x];
	}
	if (a.max_level_rand_training) {
		float* ml = a.max_level_ptr + base;
		for (uint32_t j = 0; j < numsteps; ++j) ml[j] = max_level;
	}
}

// Pass 2 (testbed_nerf.cu:1239-1253) with one WAVE per kept ray: lane k of a 64-sample chunk rebuilds t_k by the same sequence
// of `t += calc_dt(t)` additions the reference performs (bit-identical), and the 28-byte records of consecutive samples are
// written by consecutive lanes (coalesced) instead of one lane dribbling out its whole ray.
__global__ void __launch_bounds__(256) expand_training_samples_kernel(const TrainSampleArgs a) {
	constexpr uint32_t MAX_RUNS = NGP_MAX_RUNS;
	__shared__ float s_t[4][MAX_RUNS];
	__shared__ uint32_t s_n[4][MAX_RUNS];
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t slot = blockIdx.x * 4 + w;
	if (slot >= *a.ray_counter) return;
	const uint32_t numsteps = a.numsteps_out[slot * 2 + 0], base = a.numsteps_out[slot * 2 + 1];
	if (numsteps == 0) return;
	NgpCoord* co = a.coords_out + base;
	const uint32_t* stash = (const uint32_t*)co;
	const uint32_t n_runs = stash[0];
	const float startt = __uint_as_float(stash[1]);
	// lane r fetches run r of the recipe into LDS before any lane overwrites the slots (n_runs <= numsteps, so it always fits)
	if (n_runs <= MAX_RUNS && lane < n_runs) { s_t[w][lane] = __uint_as_float(stash[2 + 2 * lane]); s_n[w][lane] = stash[3 + 2 * lane]; }
	const NgpRay ray = a.rays_out[slot];
	const v3 ro = ld3(ray.o), rd = normalized(ld3(ray.d));
	const v3 warped_dir = warp_direction(rd);
	const float cone_angle = a.cone_angle_constant;
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the recipe is out of the slots before any lane overwrites them

	if (n_runs <= MAX_RUNS) {
		uint32_t j0 = 0;
		for (uint32_t r = 0; r < n_runs; ++r) {
			const uint32_t cnt = s_n[w][r];
			float t_chunk = s_t[w][r];
			for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
				float t = t_chunk;
				// lane k replays k additions; lanes past the end of the run replay none (runs are short: the wave loops max(k) times)
				const uint32_t n_add = c0 + lane < cnt ? lane : 0u;
				for (uint32_t k = 0; k < n_add; ++k) t += calc_dt(t, cone_angle);
				const float dt = calc_dt(t, cone_angle);
				if (c0 + lane < cnt) {
					const v3 wp = aabb_relative_pos(a.aabb, ro + rd * t);
					NgpCoord c;
					c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z; c.dt = warp_dt(dt);
					c.dir[0] = warped_dir.x; c.dir[1] = warped_dir.y; c.dir[2] = warped_dir.z;
					co[j0 + c0 + lane] = c;
				}
				t_chunk = __shfl(t + dt, 63, 64);   // only read when the run continues, i.e. when lane 63 was inside it
			}
			j0 += cnt;
		}
		return;
	}
	// more runs than recipe slots (very fragmented occupancy): the reference's full second march, one lane
	if (lane != 0) return;
	const v3 idir = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
	float t = startt;
	v3 pos;
	uint32_t j = 0;
	OccBrick occ;
	while (j < numsteps && aabb_contains(a.aabb, pos = ro + rd * t)) {
		const float dt = calc_dt(t, cone_angle);
		const uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
		if (density_grid_occupied_at(pos, a.density_grid, mip, occ)) {
			const v3 wp = aabb_relative_pos(a.aabb, pos);
			NgpCoord c;
			c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z; c.dt = warp_dt(dt);
			c.dir[0] = warped_dir.x; c.dir[1] = warped_dir.y; c.dir[2] = warped_dir.z;
			co[j] = c;
			++j;
			t += dt;
		} else {
			t = advance_to_next_voxel(t, cone_angle, pos, rd, idir, NGP_NERF_GRIDSIZE >> mip);
		}
	}
}


// ------------------------------------------------------------------------------------------------------------------------------
// Wave-per-ray march for the constant-step case (cone_angle == 0: aabb_scale <= 1, where calc_dt is MIN_CONE_STEPSIZE for every t).
//
// Every t the reference's march visits lies on ONE additive sequence t_{k+1} = fl(t_k + dt) from the jittered start: an occupied sample
// steps by dt, and advance_to_next_voxel (testbed_nerf.cu:201-213) walks the SAME additions until t >= t_target.  The march is therefore
// a walk over candidate INDICES: at candidate k, "occupied -> emit, go to k + 1", "empty -> go to the first k' > k with t_k' >= t_k +
// distance_to_next_voxel(pos_k)".  Inside one binade [2^e, 2^(e+1)) every t is a multiple of u = 2^(e-23) and fl(t + dt) = t + inc * u
// with inc = round(dt / u) (no tie) as long as the sum stays below 2^(e+1): there t_k = fma(k - k_begin, inc * u, t_begin) EXACTLY (the
// true value is representable, so the fused multiply-add rounds nothing).  A ray is a handful of such segments (one per binade it
// crosses, the crossing steps done by real fp32 additions); with t_k in closed form the 64 lanes of a wave evaluate 64 consecutive
// candidates at once (position, box test, mip, occupancy bit, skip target by exact arithmetic on the segment) and the walk over them is
// a SCALAR loop on the two ballot masks (v_readlane for the skip targets; runs of occupied candidates by count-trailing-ones).  Same
// visited candidates, same t, same samples as the serial march, bit for bit — without its ~450-trip dependent chain per wave.
// One workgroup = 4 waves x 4 rays; sample slots are reserved once per workgroup (one atomic per counter), then every wave writes the
// 28-byte records of its rays from the per-window emit masks (consecutive samples by consecutive lanes).
constexpr uint32_t WM_WAVES = 4, WM_RAYS_PER_WAVE = 4, WM_RAYS_PER_WG = WM_WAVES * WM_RAYS_PER_WAVE;
constexpr uint32_t WM_MAX_SEGS = 20;       // binades from t ~ 2^-18 up to 4 (a start at t = 0 takes one degenerate segment, then e = -10 ... 1)
constexpr uint32_t WM_MAX_WINDOWS = 32;    // windows that emitted samples; a ray that needs more takes the serial path
constexpr uint32_t WM_MAX_CANDIDATES = 1u << 20;

struct WmSeg { uint32_t k_begin, k_last; float t_begin, delta, inv_delta; };   // t_k = fma(k - k_begin, delta, t_begin) for k_begin <= k <= k_last

// segments of the additive sequence from `startt` until it has left the box; false = more than WM_MAX_SEGS (serial path)
__device__ __forceinline__ bool wm_build_segments(float startt, v3 ro, v3 rd, const Aabb& aabb, WmSeg* __restrict__ segs, uint32_t& n_segs) {
	const float dt = MIN_CONE_STEPSIZE();
	float t = startt;
	uint32_t k = 0;
	n_segs = 0;
	for (;;) {
		if (n_segs == WM_MAX_SEGS || k >= WM_MAX_CANDIDATES) return false;
		WmSeg s; s.k_begin = k; s.k_last = k; s.t_begin = t; s.delta = 0.f; s.inv_delta = 0.f;
		const uint32_t bits = __float_as_uint(t);
		const int e = (int)(bits >> 23) - 127;
		if (e >= -60 && e <= 60) {   // (t == 0, denormals, huge values: a degenerate segment, every step a real addition)
			const float x = dt * __uint_as_float((uint32_t)(127 + 23 - e) << 23);   // dt / u, exact
			if (x < 16777216.0f) {
				const float q = floorf(x), frac = x - q;
				const uint32_t inc = (uint32_t)q + (frac > 0.5f ? 1u : 0u);
				if (frac != 0.5f && inc != 0u) {   // a tie rounds to even, i.e. alternates: no closed form in that binade
					const uint32_t m = (bits & 0x007fffffu) | 0x00800000u;
					const uint32_t n = (0x00ffffffu - m) / inc;   // steps that stay below 2^(e+1)
					s.k_last = k + n;
					s.delta = (float)inc * __uint_as_float((uint32_t)(127 - 23 + e) << 23);
					s.inv_delta = 1.0f / s.delta;
				}
			}
		}
		segs[n_segs++] = s;
		const float t_last = __builtin_fmaf((float)(s.k_last - s.k_begin), s.delta, t);
		// positions move monotonically along every axis: once a candidate is outside the (convex) box, all later ones are
		if (!aabb_contains(aabb, ro + rd * t_last)) return true;
		const float t_next = t_last + dt;   // the crossing step
		if (!(t_next > t_last)) return true;   // dt below half an ulp (t > 2^14) or a non-finite t: the sequence does not advance; treat the rest as outside
		t = t_next;
		k = s.k_last + 1;
	}
}

// wave-uniform copy of lane `l`'s value (v_readlane_b32: the result lives in a scalar register)
__device__ __forceinline__ float wm_lane_f(float v, uint32_t l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)l)); }
__device__ __forceinline__ uint32_t wm_lane_u(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ v3 wm_lane_v3(v3 v, uint32_t l) { return mk(wm_lane_f(v.x, l), wm_lane_f(v.y, l), wm_lane_f(v.z, l)); }

// the reference's own march, one lane (rays whose bookkeeping does not fit: > WM_MAX_SEGS binades or > WM_MAX_WINDOWS sample windows)
template <bool CONST_DT = true>
__device__ __forceinline__ uint32_t wm_serial_march(const TrainSampleArgs& a, v3 ro, v3 rd, v3 idir, float startt, NgpCoord* __restrict__ co, v3 warped_dir, uint32_t limit) {
	uint32_t j = 0;
	float t = startt;
	v3 pos;
	OccBrick occ;
	while (aabb_contains(a.aabb, pos = ro + rd * t) && j < limit) {
		const float dt = calc_dt_t<CONST_DT>(t, a.cone_angle_constant);
		const uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
		if (density_grid_occupied_at(pos, a.density_grid, mip, occ)) {
			if (co) {
				const v3 wp = aabb_relative_pos(a.aabb, pos);
				NgpCoord c;
				c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z; c.dt = warp_dt(dt);
				c.dir[0] = warped_dir.x; c.dir[1] = warped_dir.y; c.dir[2] = warped_dir.z;
				co[j] = c;
			}
			++j; t += dt;
		} else {
			t = advance_to_next_voxel<CONST_DT>(t, a.cone_angle_constant, pos, rd, idir, NGP_NERF_GRIDSIZE >> mip);
		}
	}
	return j;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NGP_MARCH_WAVES_PER_EU, 8))) generate_training_samples_wave_kernel(const TrainSampleArgs a) {
	__shared__ uint32_t s_brick_any[NGP_NERF_GRID_N_CELLS / 64 / 32];
	__shared__ WmSeg s_segs[WM_RAYS_PER_WG][WM_MAX_SEGS];
	__shared__ uint64_t s_win_mask[WM_RAYS_PER_WG][WM_MAX_WINDOWS];
	__shared__ uint32_t s_win_k0[WM_RAYS_PER_WG][WM_MAX_WINDOWS];
	__shared__ uint32_t s_n_windows[WM_RAYS_PER_WG], s_numsteps[WM_RAYS_PER_WG], s_base[WM_RAYS_PER_WG], s_slot[WM_RAYS_PER_WG];
	__shared__ uint8_t s_serial[WM_RAYS_PER_WG];
	__shared__ uint32_t s_spread[128];   // expand_bits of the 7-bit cell coordinates (morton3D by three LDS reads)
	if (threadIdx.x < 128u) s_spread[threadIdx.x] = expand_bits(threadIdx.x);

	if (a.brick_summary) { for (uint32_t q = threadIdx.x; q < NGP_NERF_GRID_N_CELLS / 64 / 32; q += blockDim.x) s_brick_any[q] = a.brick_summary[q]; }
	else load_brick_summary(a.density_grid, s_brick_any);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	const float dt = MIN_CONE_STEPSIZE();

	// persistent workgroups: the grid size is the throttle (the march shares the chip with the step's backward pass, whose 256-register kernel
	// loses a resident wave on every SIMD that also hosts march waves)
	const uint32_t n_groups = (a.n_rays + WM_RAYS_PER_WG - 1) / WM_RAYS_PER_WG;
	for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
		// ---- per-ray setup on lanes 0..3 of every wave (identical to the serial kernel's)
		const uint32_t li = group * WM_RAYS_PER_WG + w * WM_RAYS_PER_WAVE + lane;
		const bool setup_lane = lane < WM_RAYS_PER_WAVE;
		const bool in_range = setup_lane && li < a.n_rays;
		const uint32_t i = li + a.ray_offset;
		bool valid = false;   // the ray exists, its pixel is not masked and its first candidate lies inside the box
		bool pixel_ok = false;
		float startt = 0.f, max_level = 1.0f;
		v3 ro = mk(0, 0, 0), rd_unnorm = mk(0, 0, 1), rd = mk(0, 0, 1), idir = mk(1, 1, 1);
		uint32_t n_segs = 0;
		bool serial = false;
		if (in_range) {
			const TrainRaySetup rs = setup_training_ray(a, i);
			if (rs.pixel_ok) {
				pixel_ok = true;
				max_level = rs.max_level; ro = rs.ro; rd_unnorm = rs.rd_unnorm; rd = rs.rd; idir = rs.idir; startt = rs.startt;
				valid = aabb_contains(a.aabb, ro + rd * startt);   // otherwise the reference's loop ends before its first iteration
				if (valid && a.dev_variant != 3) serial = !wm_build_segments(startt, ro, rd, a.aabb, s_segs[w * WM_RAYS_PER_WAVE + lane], n_segs);
			}
		}
		const v3 warped_dir = warp_direction(rd);

		// ---- march: the wave takes its rays one after the other; ray parameters become scalars
	#pragma unroll 1
		for (uint32_t q = 0; q < WM_RAYS_PER_WAVE; ++q) {
			const uint32_t rl = w * WM_RAYS_PER_WAVE + q;
			const bool r_valid = wm_lane_u(valid, q) != 0 && a.dev_variant != 3;
			const bool r_serial = wm_lane_u(serial, q) != 0;
			uint32_t j = 0, n_windows = 0;
			bool overflow = false;
			if (r_valid && !r_serial) {
				const v3 o = wm_lane_v3(ro, q), d = wm_lane_v3(rd, q), id = wm_lane_v3(idir, q);
				const uint32_t ns = wm_lane_u(n_segs, q);
				const WmSeg* __restrict__ segs = s_segs[rl];
				uint32_t k0 = 0, s_first = 0, s_loaded = 0xffffffffu;
				WmSeg A = segs[0], B = A, C = A;   // the segment of the window's first candidate and its two successors (wave-uniform values)
				bool has_b = false, has_c = false;
				bool done = false;
				while (!done) {
					if (s_first != s_loaded) {
						s_loaded = s_first;
						A = segs[s_first];
						has_b = s_first + 1 < ns; has_c = s_first + 2 < ns;
						B = segs[has_b ? s_first + 1 : s_first];
						C = segs[has_c ? s_first + 2 : s_first];
					}
					// ---- 64 candidates at once.  Common case: the window lies in A and B (segments are hundreds of candidates long except next to t = 0)
					const uint32_t k = k0 + lane;
					const bool fast = !has_c || k0 + 63u <= (uint32_t)__builtin_amdgcn_readfirstlane((int)B.k_last);
					WmSeg sg; uint32_t s = s_first;
					if (fast) {
						const bool in_b = has_b && k > A.k_last;
						sg.k_begin = in_b ? B.k_begin : A.k_begin; sg.k_last = in_b ? B.k_last : A.k_last;
						sg.t_begin = in_b ? B.t_begin : A.t_begin; sg.delta = in_b ? B.delta : A.delta; sg.inv_delta = in_b ? B.inv_delta : A.inv_delta;
						s = s_first + (in_b ? 1u : 0u);
					} else {
						while (s + 1 < ns && k > segs[s].k_last) ++s;
						sg = segs[s];
					}
					const float t = __builtin_fmaf((float)(k - sg.k_begin), sg.delta, sg.t_begin);
					const v3 pos = o + d * t;
					const bool inside = k <= sg.k_last && aabb_contains(a.aabb, pos);
					const uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
					int ix, iy, iz;
					cascaded_grid_coords(pos, mip, ix, iy, iz);
					const uint32_t idx = s_spread[ix] | (s_spread[iy] << 1) | (s_spread[iz] << 2);   // morton3D
					const uint32_t brick = (idx >> 6) + (NGP_NERF_GRID_N_CELLS / 64u) * mip;
					bool occ = false;
					if (inside && !(mip == 0 && !((s_brick_any[brick >> 5] >> (brick & 31u)) & 1u))) occ = (((const uint64_t*)a.density_grid)[brick] >> (idx & 63u)) & 1ull;
					uint32_t nxt = k + 1;
					if (inside && !occ) {
						// advance_to_next_voxel: the first k' > k with t_k' >= t_target, by exact arithmetic on the segments
						const float t_target = t + distance_to_next_voxel(pos, d, id, NGP_NERF_GRIDSIZE >> mip);
						uint32_t ss = s, kcur = k;
						WmSeg g = sg;
						float tcur = t;
						for (;;) {
							const float t_last = __builtin_fmaf((float)(g.k_last - g.k_begin), g.delta, g.t_begin);
							if (t_target <= t_last) {
								// same binade: t_target - tcur is exact; the estimate is within one of the answer, settled on the exact values
								uint32_t n = 0;
								const float diff = t_target - tcur;
								if (diff > 0.0f) {
									n = (uint32_t)ceilf(diff * g.inv_delta);
									if (n > 0u && __builtin_fmaf((float)(n - 1u), g.delta, tcur) >= t_target) --n;
									else if (__builtin_fmaf((float)n, g.delta, tcur) < t_target) ++n;
								}
								if (kcur == k && n == 0u) n = 1u;   // do { t += dt } while (t < t_target): at least one step
								nxt = kcur + n;
								break;
							}
							if (ss + 1 >= ns) { nxt = g.k_last + 1u; break; }   // beyond every segment: outside the box
							++ss;
							if (fast && ss == s_first + 1u) g = B; else if (fast && ss == s_first + 2u) g = C; else g = segs[ss];
							kcur = g.k_begin; tcur = g.t_begin;
							if (tcur >= t_target) { nxt = kcur; break; }
						}
					}
					// ---- the walk over this window by pointer doubling: reach = candidates on the path from this lane, jmp = where the path stands after
					// 2^i hops (64 = it left the window or the ray ended).  Lane 0's reach after <= 6 rounds is the visited set.
					uint32_t jmp = 64u;
					if (inside) { jmp = nxt - k0; jmp = jmp < 64u ? jmp : 64u; }
					uint32_t r_lo = lane < 32u ? (1u << lane) : 0u, r_hi = lane >= 32u ? (1u << (lane - 32u)) : 0u;
	#pragma unroll 1
					for (int it = 0; it < 6; ++it) {
						if (wm_lane_u(jmp, 0) >= 64u) break;   // only lane 0's path matters
						const bool live = jmp < 64u;
						const int src = (int)((live ? jmp : lane) << 2);
						const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r_lo), pb = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r_hi);
						const uint32_t pj = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)jmp);
						if (live) { r_lo |= pa; r_hi |= pb; jmp = pj; }
					}
					const uint64_t visited = (uint64_t)wm_lane_u(r_lo, 0) | ((uint64_t)wm_lane_u(r_hi, 0) << 32);
					const uint32_t last = 63u - (uint32_t)__builtin_clzll(visited);   // lane 0 is always visited
					const uint64_t in_m = __ballot(inside), occ_m = __ballot(occ);
					uint64_t emit = visited & occ_m;
					if (!((in_m >> last) & 1ull)) done = true;   // the walk reached a candidate outside the box (1204)
					const uint32_t n_emit = (uint32_t)__popcll(emit);
					if (j + n_emit >= NGP_NERF_STEPS) {   // j < NERF_STEPS (1204): the ray ends with its 1024th sample
						uint32_t room = NGP_NERF_STEPS - j;
						while ((uint32_t)__popcll(emit) > room) emit &= ~(1ull << (63u - (uint32_t)__builtin_clzll(emit)));
						done = true;
					}
					j += (uint32_t)__popcll(emit);
					if (emit) {
						if (n_windows < WM_MAX_WINDOWS) { if (lane == 0) { s_win_mask[rl][n_windows] = emit; s_win_k0[rl][n_windows] = k0; } }
						else overflow = true;
						++n_windows;
					}
					k0 = wm_lane_u(nxt, last);   // an occupied last candidate: k + 1
					if (k0 >= WM_MAX_CANDIDATES) done = true;
					while (s_first + 1 < ns && k0 > segs[s_first].k_last) ++s_first;
					s_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_first);
				}
			}
			bool ray_serial = r_serial || overflow;
			if (r_valid && ray_serial) {
				// count with the reference's own loop (one lane); the write pass repeats it
				uint32_t cnt = 0;
				if (lane == q) cnt = wm_serial_march(a, ro, rd, idir, startt, nullptr, warped_dir, NGP_NERF_STEPS);
				j = wm_lane_u(cnt, q);
			}
			const bool r_pixel_ok = wm_lane_u(pixel_ok, q) != 0;
			if (lane == 0) { s_numsteps[rl] = j; s_n_windows[rl] = n_windows; s_serial[rl] = (ray_serial ? 1 : 0) | (r_pixel_ok ? 2 : 0); }
		}
		__syncthreads();

		// ---- slot reservation, once per workgroup (the reference: two atomics per ray, 1225 / 1232)
		if (w == 0) {
			const bool has = lane < WM_RAYS_PER_WG;
			uint32_t numsteps = has ? s_numsteps[lane] : 0u;
			// a ray past n_rays or on a masked pixel returns before both atomics in the reference (1115-1135); one without samples unless an envmap trains (1221)
			bool keep = has && (s_serial[lane < WM_RAYS_PER_WG ? lane : 0] & 2) && !(numsteps == 0 && !a.train_envmap);
			if (!keep) numsteps = 0;
			const uint32_t incl = wave_inclusive_scan(numsteps);
			const uint32_t total = __shfl(incl, 63, 64);
			uint32_t wg_base = 0;
			if (lane == 63 && total) wg_base = atomicAdd(a.numsteps_counter, total);
			wg_base = __shfl(wg_base, 63, 64);
			const uint32_t base = wg_base + incl - numsteps;
			if (keep && base + numsteps > a.max_samples) keep = false;   // dropped AFTER the counter was bumped (1225-1228)
			const unsigned long long kept_mask = __ballot(keep);
			const uint32_t n_kept = (uint32_t)__popcll(kept_mask);
			uint32_t ray_base = 0;
			if (lane == 0 && n_kept) ray_base = atomicAdd(a.ray_counter, n_kept);
			ray_base = __shfl(ray_base, 0, 64);
			if (has) { s_base[lane] = base; s_slot[lane] = keep ? ray_base + (uint32_t)__popcll(kept_mask & ((1ull << lane) - 1ull)) : 0xffffffffu; }
		}
		__syncthreads();

		// ---- write pass
	#pragma unroll 1
		for (uint32_t q = 0; q < WM_RAYS_PER_WAVE; ++q) {
			const uint32_t rl = w * WM_RAYS_PER_WAVE + q;
			const uint32_t slot = s_slot[rl];
			if (slot == 0xffffffffu) continue;
			const uint32_t numsteps = s_numsteps[rl], base = s_base[rl];
			if (lane == q) {
				a.ray_indices_out[slot] = i;
				NgpRay ray_out; ray_out.o[0] = ro.x; ray_out.o[1] = ro.y; ray_out.o[2] = ro.z; ray_out.d[0] = rd_unnorm.x; ray_out.d[1] = rd_unnorm.y; ray_out.d[2] = rd_unnorm.z;
				a.rays_out[slot] = ray_out;
				a.numsteps_out[slot * 2 + 0] = numsteps;
				a.numsteps_out[slot * 2 + 1] = base;
			}
			if (numsteps == 0 || a.dev_variant == 2) continue;
			NgpCoord* __restrict__ co = a.coords_out + base;
			const float ml = wm_lane_f(max_level, q);
			if (s_serial[rl] & 1) {
				if (lane == q) wm_serial_march(a, ro, rd, idir, startt, co, warped_dir, numsteps);
				if (a.max_level_rand_training) for (uint32_t jj = lane; jj < numsteps; jj += 64u) a.max_level_ptr[base + jj] = ml;
				continue;
			}
			const v3 o = wm_lane_v3(ro, q);
			const v3 d = wm_lane_v3(rd, q);
			const v3 wd = wm_lane_v3(warped_dir, q);
			const uint32_t ns = wm_lane_u(n_segs, q);
			const WmSeg* __restrict__ segs = s_segs[rl];
			const uint32_t n_windows = s_n_windows[rl];
			uint32_t j0 = 0;
			for (uint32_t wi = 0; wi < n_windows; ++wi) {
				const uint64_t mask = s_win_mask[rl][wi];
				const uint32_t k = s_win_k0[rl][wi] + lane;
				if ((mask >> lane) & 1ull) {
					uint32_t s = 0;
					while (s + 1 < ns && k > segs[s].k_last) ++s;
					const WmSeg sg = segs[s];
					const float t = __builtin_fmaf((float)(k - sg.k_begin), sg.delta, sg.t_begin);
					const v3 wp = aabb_relative_pos(a.aabb, o + d * t);
					const uint32_t jj = j0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
					NgpCoord c;
					c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z; c.dt = warp_dt(dt);
					c.dir[0] = wd.x; c.dir[1] = wd.y; c.dir[2] = wd.z;
					co[jj] = c;
					if (a.max_level_rand_training) a.max_level_ptr[base + jj] = ml;
				}
				j0 += (uint32_t)__popcll(mask);
			}
		}
	}
}


// ------------------------------------------------------------------------------------------------------------------------------
// Wave-per-ray march for CONE stepping (cone_angle != 0: every dataset with aabb_scale > 1, testbed_nerf.cu:2730 — the fox photographs, every real capture).
//
// dt = clamp(t * cone_angle, MIN, MAX) has no closed form, but the candidate sequence t_{k+1} = fl(t_k + calc_dt(t_k)) is still ONE sequence per ray,
// independent of the occupancy grid: an occupied sample steps by calc_dt(t), and advance_to_next_voxel (testbed_nerf.cu:201-213) walks the very same additions
// until t >= t_target.  So the sequence is GENERATED instead of written down: the ray's own lane runs the recurrence once (three dependent VALU
// instructions per step and no memory access: ~5 us for a 700-candidate ray, all rays of the workgroup at the same time) and leaves every 8th t in
// LDS (1 KiB per ray: 2048 candidates, more than a ray through an aabb_scale-128 box has).  After that the wave handles its rays one after the other, 64
// CONSECUTIVE candidates at once: lane l takes checkpoint (64 w + l) / 8 and replays l mod 8 additions — the same fp32 additions in the same order, so the
// same bits — then evaluates position, box test, mip (per candidate: it depends on dt), occupancy bit and, for an empty candidate, the skip target
// t + distance_to_next_voxel.  The successor "first k' > k with t_k' >= t_target" is found among the window's own t by an estimate ((t_target - t) / dt)
// settled on the neighbours' exact values (wave shuffles); a target beyond the window is carried into the next window as `pending` (candidates below it are
// passed over — whole windows by one checkpoint compare).  The walk over the window is the pointer doubling of the constant-step kernel, entered at the
// first candidate that reaches `pending`.  Windows are therefore always the aligned blocks [64 w, 64 w + 64): at most 32 per ray, no overflow path.
// Same visited candidates, same t, same samples as the serial march, bit for bit (tests/test_sampling_gpu.py runs every cone case on both).
constexpr uint32_t CW_STRIDE = 8, CW_MAX_CP = 256;   // checkpoints: t of every 8th candidate; CW_STRIDE * CW_MAX_CP = 2048 candidates = WM_MAX_WINDOWS windows of 64

// t_{k + n} from t_k, n < CW_STRIDE: the reference's own additions
__device__ __forceinline__ float cw_replay(float t, uint32_t n, float cone_angle) {
#pragma unroll
	for (uint32_t i = 0; i + 1 < CW_STRIDE; ++i) if (i < n) t += calc_dt(t, cone_angle);
	return t;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NGP_MARCH_WAVES_PER_EU, 8))) generate_training_samples_cone_wave_kernel(const TrainSampleArgs a) {
	__shared__ uint32_t s_brick_any[NGP_NERF_GRID_N_CELLS / 64 / 32];
	__shared__ float s_cp[WM_RAYS_PER_WG][CW_MAX_CP];
	__shared__ uint64_t s_win_mask[WM_RAYS_PER_WG][WM_MAX_WINDOWS];
	__shared__ uint8_t s_win_w[WM_RAYS_PER_WG][WM_MAX_WINDOWS];
	__shared__ uint32_t s_n_windows[WM_RAYS_PER_WG], s_numsteps[WM_RAYS_PER_WG], s_base[WM_RAYS_PER_WG], s_slot[WM_RAYS_PER_WG];
	__shared__ uint8_t s_serial[WM_RAYS_PER_WG];
	__shared__ uint32_t s_spread[128];   // expand_bits of the 7-bit cell coordinates (morton3D by three LDS reads)
	static_assert(CW_STRIDE * CW_MAX_CP == 64u * WM_MAX_WINDOWS, "one emit-mask slot per window of the checkpoint table");
	if (threadIdx.x < 128u) s_spread[threadIdx.x] = expand_bits(threadIdx.x);
	if (a.brick_summary) { for (uint32_t q = threadIdx.x; q < NGP_NERF_GRID_N_CELLS / 64 / 32; q += blockDim.x) s_brick_any[q] = a.brick_summary[q]; }
	else load_brick_summary(a.density_grid, s_brick_any);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
	const float cone = a.cone_angle_constant;
	const float INF = __uint_as_float(0x7f800000u);

	const uint32_t n_groups = (a.n_rays + WM_RAYS_PER_WG - 1) / WM_RAYS_PER_WG;
	for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
		// ---- per-ray setup + generation of the candidate sequence on lanes 0..3 of every wave
		const uint32_t li = group * WM_RAYS_PER_WG + w * WM_RAYS_PER_WAVE + lane;
		const bool setup_lane = lane < WM_RAYS_PER_WAVE;
		const bool in_range = setup_lane && li < a.n_rays;
		const uint32_t i = li + a.ray_offset;
		bool valid = false, pixel_ok = false, serial = false;
		float startt = 0.f, max_level = 1.0f;
		v3 ro = mk(0, 0, 0), rd_unnorm = mk(0, 0, 1), rd = mk(0, 0, 1), idir = mk(1, 1, 1);
		uint32_t n_cp = 0;
		if (in_range) {
			const TrainRaySetup rs = setup_training_ray(a, i);
			if (rs.pixel_ok) {
				pixel_ok = true;
				max_level = rs.max_level; ro = rs.ro; rd_unnorm = rs.rd_unnorm; rd = rs.rd; idir = rs.idir; startt = rs.startt;
				valid = aabb_contains(a.aabb, ro + rd * startt);   // otherwise the reference's loop ends before its first iteration
				if (valid && a.dev_variant != 3) {
					// positions move monotonically along every axis: once a candidate is outside the (convex) box all later ones are, so the table ends with the
					// first block whose first candidate is outside
					float* __restrict__ cp = s_cp[w * WM_RAYS_PER_WAVE + lane];
					float t = startt;
					for (;;) {
						if (n_cp == CW_MAX_CP) { serial = true; break; }   // still inside after 2048 candidates: the reference's own loop, one lane
						cp[n_cp++] = t;
#pragma unroll
						for (uint32_t s = 0; s < CW_STRIDE; ++s) t += calc_dt(t, cone);
						if (!aabb_contains(a.aabb, ro + rd * t)) break;
					}
				}
			}
		}
		const v3 warped_dir = warp_direction(rd);

		// ---- march: the wave takes its rays one after the other; ray parameters become scalars
	#pragma unroll 1
		for (uint32_t q = 0; q < WM_RAYS_PER_WAVE; ++q) {
			const uint32_t rl = w * WM_RAYS_PER_WAVE + q;
			const bool r_valid = wm_lane_u(valid, q) != 0 && a.dev_variant != 3;
			const bool r_serial = wm_lane_u(serial, q) != 0;
			uint32_t j = 0, n_windows = 0;
			if (r_valid && !r_serial) {
				const v3 o = wm_lane_v3(ro, q), d = wm_lane_v3(rd, q), id = wm_lane_v3(idir, q);
				const uint32_t ncp = wm_lane_u(n_cp, q);
				const float* __restrict__ cp = s_cp[rl];
				const uint32_t n_win = (ncp + 7u) >> 3;   // windows that hold at least one checkpoint
				float pending = -INF;                     // skip target carried over from the previous window (wave-uniform)
				bool done = false;
	#pragma unroll 1
				for (uint32_t win = 0; win < n_win && !done; ++win) {
					// a whole window below the pending target: its successor's first t decides
					if ((win + 1u) * 8u < ncp && cp[(win + 1u) * 8u] < pending) continue;
					const uint32_t k = win * 64u + lane;
					const uint32_t ci = k >> 3;
					const bool have = ci < ncp;
					const float t = cw_replay(cp[have ? ci : 0u], k & 7u, cone);
					const float dt = calc_dt(t, cone);
					const v3 pos = o + d * t;
					const bool inside = have && aabb_contains(a.aabb, pos);
					const uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
					int ix, iy, iz;
					cascaded_grid_coords(pos, mip, ix, iy, iz);
					const uint32_t idx = s_spread[ix] | (s_spread[iy] << 1) | (s_spread[iz] << 2);   // morton3D
					const uint32_t brick = (idx >> 6) + (NGP_NERF_GRID_N_CELLS / 64u) * mip;
					bool occ = false;
					if (inside && !(mip == 0 && !((s_brick_any[brick >> 5] >> (brick & 31u)) & 1u))) occ = (((const uint64_t*)a.density_grid)[brick] >> (idx & 63u)) & 1ull;
					// the walk is entered at the first candidate that reaches the pending target; a candidate outside the box ends the ray whatever its t
					// (everything behind it is outside as well), so outside candidates count as t = +inf everywhere below
					const float te = inside ? t : INF;
					const uint64_t reach_m = __ballot(te >= pending);
					if (!reach_m) continue;   // (cannot happen after the checkpoint test above unless the table ended inside the window; then lanes past it are +inf)
					const uint32_t entry = (uint32_t)__builtin_ctzll(reach_m);
					// ---- successor of every candidate, window-relative: occupied -> lane + 1; empty -> first lane r > lane with te_r >= t_target (64 = beyond the window)
					float t_target = 0.0f;
					uint32_t r = lane + 1u;
					const bool skip = inside && !occ;
					if (skip) {
						const float dist = distance_to_next_voxel(pos, d, id, NGP_NERF_GRIDSIZE >> mip);
						t_target = t + dist;
						const float est = ceilf(dist * __builtin_amdgcn_rcpf(dt));   // do { t += dt } while (t < t_target): at least one step
						uint32_t n = est >= 1.0f ? (est < 64.0f ? (uint32_t)est : 64u) : 1u;
						r = lane + n; r = r < 64u ? r : 64u;
					}
					bool settled = !skip;
	#pragma unroll 1
					while (__ballot(!settled)) {
						const float ta = __shfl(te, (int)((r - 1u) & 63u), 64), tb = __shfl(te, (int)(r & 63u), 64);
						if (!settled) {
							if (r > lane + 1u && ta >= t_target) --r;
							else if (r < 64u && tb < t_target) ++r;
							else settled = true;
						}
					}
					// ---- the walk by pointer doubling: reach = candidates on the path from this lane, jmp = where the path stands after 2^i hops (64 = it left the window or the ray ended)
					uint32_t jmp = inside ? r : 64u;
					uint32_t r_lo = lane < 32u ? (1u << lane) : 0u, r_hi = lane >= 32u ? (1u << (lane - 32u)) : 0u;
	#pragma unroll 1
					for (int it = 0; it < 6; ++it) {
						if (wm_lane_u(jmp, entry) >= 64u) break;   // only the entry lane's path matters
						const bool live = jmp < 64u;
						const int src = (int)((live ? jmp : lane) << 2);
						const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r_lo), pb = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)r_hi);
						const uint32_t pj = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)jmp);
						if (live) { r_lo |= pa; r_hi |= pb; jmp = pj; }
					}
					const uint64_t visited = (uint64_t)wm_lane_u(r_lo, entry) | ((uint64_t)wm_lane_u(r_hi, entry) << 32);
					const uint32_t last = 63u - (uint32_t)__builtin_clzll(visited);   // the entry lane is always visited
					const uint64_t in_m = __ballot(inside), occ_m = __ballot(occ);
					uint64_t emit = visited & occ_m;
					if (!((in_m >> last) & 1ull)) done = true;   // the walk reached a candidate outside the box (1204)
					if (j + (uint32_t)__popcll(emit) >= NGP_NERF_STEPS) {   // j < NERF_STEPS (1204): the ray ends with its 1024th sample
						const uint32_t room = NGP_NERF_STEPS - j;
						while ((uint32_t)__popcll(emit) > room) emit &= ~(1ull << (63u - (uint32_t)__builtin_clzll(emit)));
						done = true;
					}
					j += (uint32_t)__popcll(emit);
					if (emit) {
						if (lane == 0) { s_win_mask[rl][n_windows] = emit; s_win_w[rl][n_windows] = (uint8_t)win; }
						++n_windows;
					}
					// an occupied last candidate is lane 63 (otherwise the walk went on): the next window starts fresh; an empty one carries its target
					pending = ((occ_m >> last) & 1ull) ? -INF : wm_lane_f(t_target, last);
				}
			}
			if (r_valid && r_serial) {
				// count with the reference's own loop (one lane); the write pass repeats it
				uint32_t cnt = 0;
				if (lane == q) cnt = wm_serial_march<false>(a, ro, rd, idir, startt, nullptr, warped_dir, NGP_NERF_STEPS);
				j = wm_lane_u(cnt, q);
			}
			const bool r_pixel_ok = wm_lane_u(pixel_ok, q) != 0;
			if (lane == 0) { s_numsteps[rl] = j; s_n_windows[rl] = n_windows; s_serial[rl] = (r_serial ? 1 : 0) | (r_pixel_ok ? 2 : 0); }
		}
		__syncthreads();

		// ---- slot reservation, once per workgroup (the reference: two atomics per ray, 1225 / 1232)
		if (w == 0) {
			const bool has = lane < WM_RAYS_PER_WG;
			uint32_t numsteps = has ? s_numsteps[lane] : 0u;
			bool keep = has && (s_serial[lane < WM_RAYS_PER_WG ? lane : 0] & 2) && !(numsteps == 0 && !a.train_envmap);
			if (!keep) numsteps = 0;
			const uint32_t incl = wave_inclusive_scan(numsteps);
			const uint32_t total = __shfl(incl, 63, 64);
			uint32_t wg_base = 0;
			if (lane == 63 && total) wg_base = atomicAdd(a.numsteps_counter, total);
			wg_base = __shfl(wg_base, 63, 64);
			const uint32_t base = wg_base + incl - numsteps;
			if (keep && base + numsteps > a.max_samples) keep = false;   // dropped AFTER the counter was bumped (1225-1228)
			const unsigned long long kept_mask = __ballot(keep);
			const uint32_t n_kept = (uint32_t)__popcll(kept_mask);
			uint32_t ray_base = 0;
			if (lane == 0 && n_kept) ray_base = atomicAdd(a.ray_counter, n_kept);
			ray_base = __shfl(ray_base, 0, 64);
			if (has) { s_base[lane] = base; s_slot[lane] = keep ? ray_base + (uint32_t)__popcll(kept_mask & ((1ull << lane) - 1ull)) : 0xffffffffu; }
		}
		__syncthreads();

		// ---- write pass
	#pragma unroll 1
		for (uint32_t q = 0; q < WM_RAYS_PER_WAVE; ++q) {
			const uint32_t rl = w * WM_RAYS_PER_WAVE + q;
			const uint32_t slot = s_slot[rl];
			if (slot == 0xffffffffu) continue;
			const uint32_t numsteps = s_numsteps[rl], base = s_base[rl];
			if (lane == q) {
				a.ray_indices_out[slot] = i;
				NgpRay ray_out; ray_out.o[0] = ro.x; ray_out.o[1] = ro.y; ray_out.o[2] = ro.z; ray_out.d[0] = rd_unnorm.x; ray_out.d[1] = rd_unnorm.y; ray_out.d[2] = rd_unnorm.z;
				a.rays_out[slot] = ray_out;
				a.numsteps_out[slot * 2 + 0] = numsteps;
				a.numsteps_out[slot * 2 + 1] = base;
			}
			if (numsteps == 0 || a.dev_variant == 2) continue;
			NgpCoord* __restrict__ co = a.coords_out + base;
			const float ml = wm_lane_f(max_level, q);
			if (s_serial[rl] & 1) {
				if (lane == q) wm_serial_march<false>(a, ro, rd, idir, startt, co, warped_dir, numsteps);
				if (a.max_level_rand_training) for (uint32_t jj = lane; jj < numsteps; jj += 64u) a.max_level_ptr[base + jj] = ml;
				continue;
			}
			const v3 o = wm_lane_v3(ro, q);
			const v3 d = wm_lane_v3(rd, q);
			const v3 wd = wm_lane_v3(warped_dir, q);
			const float* __restrict__ cp = s_cp[rl];
			const uint32_t n_windows = s_n_windows[rl];
			uint32_t j0 = 0;
			for (uint32_t wi = 0; wi < n_windows; ++wi) {
				const uint64_t mask = s_win_mask[rl][wi];
				const uint32_t k = (uint32_t)s_win_w[rl][wi] * 64u + lane;
				if ((mask >> lane) & 1ull) {
					const float t = cw_replay(cp[k >> 3], k & 7u, cone);
					const float dt = calc_dt(t, cone);
					const v3 wp = aabb_relative_pos(a.aabb, o + d * t);
					const uint32_t jj = j0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
					NgpCoord c;
					c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z; c.dt = warp_dt(dt);
					c.dir[0] = wd.x; c.dir[1] = wd.y; c.dir[2] = wd.z;
					co[jj] = c;
					if (a.max_level_rand_training) a.max_level_ptr[base + jj] = ml;
				}
				j0 += (uint32_t)__popcll(mask);
			}
		}
		__syncthreads();   // the next group's setup lanes overwrite s_cp / the per-ray slots
	}
}

} // namespace ngp

using namespace ngp;

namespace ngp {
__global__ void brick_summary_kernel(const uint8_t* __restrict__ bitfield, uint32_t* __restrict__ out) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= NGP_NERF_GRID_N_CELLS / 64u / 32u) return;
	const uint64_t* __restrict__ words = (const uint64_t*)bitfield;
	uint32_t bits = 0;
	for (uint32_t k = 0; k < 32u; ++k) bits |= (words[w * 32u + k] != 0ull ? 1u : 0u) << k;   // as load_brick_summary
	out[w] = bits;
}
}

extern "C" int ngp_hip_bitfield_brick_summary(void* stream, const uint8_t* bitfield, uint32_t* summary_out) {
	hipLaunchKernelGGL(ngp::brick_summary_kernel, dim3(NGP_NERF_GRID_N_CELLS / 64u / 32u / 64u), dim3(64), 0, (hipStream_t)stream, bitfield, summary_out);
	NGP_LAUNCH_CHECK("brick_summary_kernel");
	return 0;
}

static int generate_training_samples_impl(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint32_t max_samples, uint64_t rng_state, uint64_t rng_inc,
	uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, NgpRay* rays_out_unnormalized, uint32_t* numsteps_out,
	NgpCoord* coords_out, uint32_t n_training_images, const NgpImageMeta* metadata, const NgpXForm* xforms, const uint8_t* density_grid,
	int max_level_rand_training, float* max_level_ptr, int snap_to_pixel_centers, int train_envmap, float cone_angle_constant,
	const float* distortion_data, const int32_t* distortion_resolution_host, uint32_t ray_offset, uint32_t n_rays_global,
	const NgpErrorMapCdf* cdf_host, const uint32_t* brick_summary, uint32_t march_mode) {
	if (!n_rays) return 0;
	TrainSampleArgs a;
	a.brick_summary = brick_summary;
	a.cdf = make_error_map_cdf(cdf_host);
	a.n_rays = n_rays; a.aabb = aabb_from_host(aabb_host); a.max_samples = max_samples; a.rng.state = rng_state; a.rng.inc = rng_inc;
	a.ray_counter = ray_counter; a.numsteps_counter = numsteps_counter; a.ray_indices_out = ray_indices_out; a.rays_out = rays_out_unnormalized;
	a.numsteps_out = numsteps_out; a.coords_out = coords_out; a.n_training_images = n_training_images; a.metadata = metadata; a.xforms = xforms;
	a.density_grid = density_grid; a.max_level_rand_training = max_level_rand_training; a.max_level_ptr = max_level_ptr;
	a.snap_to_pixel_centers = snap_to_pixel_centers; a.train_envmap = train_envmap; a.cone_angle_constant = cone_angle_constant;
	a.distortion_data = distortion_data;
	a.distortion_res[0] = distortion_resolution_host ? distortion_resolution_host[0] : 0;
	a.distortion_res[1] = distortion_resolution_host ? distortion_resolution_host[1] : 0;
	a.ray_offset = ray_offset; a.n_rays_global = n_rays_global ? n_rays_global : n_rays;
	static const int dev_variant = (int)ngp_dev_knob_u32("NGP_HIP_GEN_VARIANT", 0);   // dev: phases of the lane-per-ray pair knocked out (ngp_dev_knobs.h: folds to 0 in the product library)
	a.dev_variant = dev_variant;
	// cone_angle == 0 and not NGP_MARCH_LANE_PER_RAY: wave-per-ray march on the closed-form step sequence
	static const int mode_env = (int)ngp_dev_knob_u32("NGP_HIP_GEN_MODE", 0);   // dev / A-B: overrides the caller's choice
	if (mode_env) march_mode = (uint32_t)mode_env;
	if (cone_angle_constant == 0.0f && march_mode != NGP_MARCH_LANE_PER_RAY) {
		// all workgroups resident at once (4 per CU): the kernel has the chip to itself in this mode
		// NGP_MARCH_WAVE_PER_RAY: all workgroups resident at once (the kernel has the chip to itself).  NGP_MARCH_WAVE_PER_RAY_SHARED: 2.5 persistent
		// workgroups per CU — beside the step's backward pass every further march wave costs that pass more than it gains the march.  The optimum moves with the
		// balance of the step's two chains (round 2, backward group 245 us: sweep 192 ... 4096 workgroups, step 0.75 / 0.61 at 512 / 0.64 ms; round 3, group 210 us:
		// 512 -> 0.572-0.595, 640 -> 0.546-0.557, 768 -> 0.555-0.562, 1024 -> 0.552-0.555 ms: the march had become the longer chain)
		static const uint32_t wg_cap_env = ngp_dev_knob_u32("NGP_HIP_GEN_WGS", 0u);   // dev: sweep (round 5, profiles/r05_launch_constants.md: fox 384 ... 4096 within 1 %, lego's optimum stays at 640)
		const uint32_t n_groups = div_up(n_rays, WM_RAYS_PER_WG), wg_cap = wg_cap_env ? wg_cap_env : (march_mode == NGP_MARCH_WAVE_PER_RAY_SHARED ? 640u : 4096u);
		hipLaunchKernelGGL(generate_training_samples_wave_kernel, dim3(n_groups < wg_cap ? n_groups : wg_cap), dim3(256), 0, (hipStream_t)stream, a);
		NGP_LAUNCH_CHECK("generate_training_samples_wave_kernel");
		return 0;
	}
	if (march_mode != NGP_MARCH_LANE_PER_RAY) {
		// cone stepping (every aabb_scale > 1 dataset): wave-per-ray on the generated candidate sequence, same throttle as above
		static const uint32_t wg_cap_env = ngp_dev_knob_u32("NGP_HIP_GEN_WGS", 0u);   // dev: sweep (round 5, profiles/r05_launch_constants.md: fox 384 ... 4096 within 1 %, lego's optimum stays at 640)
		const uint32_t n_groups = div_up(n_rays, WM_RAYS_PER_WG), wg_cap = wg_cap_env ? wg_cap_env : (march_mode == NGP_MARCH_WAVE_PER_RAY_SHARED ? 640u : 4096u);
		hipLaunchKernelGGL(generate_training_samples_cone_wave_kernel, dim3(n_groups < wg_cap ? n_groups : wg_cap), dim3(256), 0, (hipStream_t)stream, a);
		NGP_LAUNCH_CHECK("generate_training_samples_cone_wave_kernel");
		return 0;
	}
	if (cone_angle_constant == 0.0f) hipLaunchKernelGGL(generate_training_samples_kernel<true>, dim3(div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, a);
	else hipLaunchKernelGGL(generate_training_samples_kernel<false>, dim3(div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, a);
	NGP_LAUNCH_CHECK("generate_training_samples_kernel");
	if (a.dev_variant != 2) {
		hipLaunchKernelGGL(expand_training_samples_kernel, dim3(div_up(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, a);
		NGP_LAUNCH_CHECK("expand_training_samples_kernel");
	}
	return 0;
}

extern "C" int ngp_hip_generate_training_samples(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint32_t max_samples, uint64_t rng_state, uint64_t rng_inc,
	uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, NgpRay* rays_out_unnormalized, uint32_t* numsteps_out,
	NgpCoord* coords_out, uint32_t n_training_images, const NgpImageMeta* metadata, const NgpXForm* xforms, const uint8_t* density_grid,
	int max_level_rand_training, float* max_level_ptr, int snap_to_pixel_centers, int train_envmap, float cone_angle_constant,
	const float* distortion_data, const int32_t* distortion_resolution_host, uint32_t ray_offset, uint32_t n_rays_global,
	const NgpErrorMapCdf* cdf_host, const uint32_t* brick_summary, uint32_t march_mode) {
	if (march_mode > NGP_MARCH_WAVE_PER_RAY_SHARED) { set_last_error("ngp_hip_generate_training_samples: unknown march_mode", hipErrorInvalidValue); return -1; }
	return generate_training_samples_impl(stream, n_rays, aabb_host, max_samples, rng_state, rng_inc, ray_counter, numsteps_counter, ray_indices_out, rays_out_unnormalized, numsteps_out,
	                                      coords_out, n_training_images, metadata, xforms, density_grid, max_level_rand_training, max_level_ptr, snap_to_pixel_centers, train_envmap,
	                                      cone_angle_constant, distortion_data, distortion_resolution_host, ray_offset, n_rays_global, cdf_host, brick_summary, march_mode);
}
