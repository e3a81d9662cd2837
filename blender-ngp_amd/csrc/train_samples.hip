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
#include <stdlib.h>

namespace ngp {

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
		const uint32_t img = image_idx(i, a.n_rays_global, a.n_training_images, a.cdf.cdf_img, nullptr);
		const NgpImageMeta& md = a.metadata[img];
		Pcg32 rng = a.rng;
		rng.advance((uint64_t)(uint32_t)(i * NGP_N_MAX_RANDOM_SAMPLES_PER_RAY));
		float u, v;
		nerf_random_image_pos_training(rng, md.res, a.snap_to_pixel_centers, a.cdf, img, u, v, nullptr);
		if (!pixel_is_masked(u, v, md.res, md.pixels, md.image_data_type)) {
			max_level = a.max_level_rand_training ? (rng.next_float() * 2.0f) : 1.0f;
			const float motionblur_time = rng.next_float();
			float xform[12];
			get_xform_given_rolling_shutter(a.xforms[img], md.rolling_shutter, u, v, motionblur_time, xform);
			if (md.rays) {
				int px = (int)(u * (float)md.res[0]), py = (int)(v * (float)md.res[1]);
				px = px < md.res[0] - 1 ? px : md.res[0] - 1; px = px > 0 ? px : 0;
				py = py < md.res[1] - 1 ? py : md.res[1] - 1; py = py > 0 ? py : 0;
				const NgpRay r = md.rays[(uint64_t)px + (uint64_t)py * (uint64_t)md.res[0]];
				ro = ld3(r.o); rd_unnorm = ld3(r.d);
			} else {
				ro = col(xform, 3);
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
				rd_unnorm = mat3_mul(xform, d); // NOT normalized (1189)
			}
			rd = normalized(rd_unnorm);
			float tmin, tmax;
			aabb_ray_intersect(a.aabb, ro, rd, tmin, tmax);
			tmin = fmaxf(tmin, 0.0f);
			startt = tmin;
			startt += calc_dt(startt, cone_angle) * rng.next_float();
			idir = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);

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

extern "C" int ngp_hip_generate_training_samples(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint32_t max_samples, uint64_t rng_state, uint64_t rng_inc,
	uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, NgpRay* rays_out_unnormalized, uint32_t* numsteps_out,
	NgpCoord* coords_out, uint32_t n_training_images, const NgpImageMeta* metadata, const NgpXForm* xforms, const uint8_t* density_grid,
	int max_level_rand_training, float* max_level_ptr, int snap_to_pixel_centers, int train_envmap, float cone_angle_constant,
	const float* distortion_data, const int32_t* distortion_resolution_host, uint32_t ray_offset, uint32_t n_rays_global,
	const NgpErrorMapCdf* cdf_host, const uint32_t* brick_summary) {
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
	const char* var = getenv("NGP_HIP_GEN_VARIANT");
	a.dev_variant = var ? atoi(var) : 0;
	if (cone_angle_constant == 0.0f) hipLaunchKernelGGL(generate_training_samples_kernel<true>, dim3(div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, a);
	else hipLaunchKernelGGL(generate_training_samples_kernel<false>, dim3(div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, a);
	NGP_LAUNCH_CHECK("generate_training_samples_kernel");
	if (a.dev_variant != 2) {
		hipLaunchKernelGGL(expand_training_samples_kernel, dim3(div_up(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, a);
		NGP_LAUNCH_CHECK("expand_training_samples_kernel");
	}
	return 0;
}
