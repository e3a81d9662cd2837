// render.hip — single-NeRF renderer kernels (Testbed::NerfTracer) + frame accumulate / tonemap for gfx950.
// Replaces src/testbed_nerf.cu:612-664 (advance_pos_nerf), 705-765 (generate_next_nerf_network_inputs), 767-989
// (composite_kernel_nerf; Shade mode, no masks / glow), 1748-1781 (shade_kernel_nerf), 1784-1807 (compact_kernel_nerf),
// 1809-1978 (init_rays_with_payload_kernel_nerf; Perspective / OpenCV / FTheta / LatLong lenses, rolling shutter, depth of field, slice plane) and src/render_buffer.cu:235-272, 274-348, 540-567.
// Compaction uses wave64 ballots: one atomic per wave per counter instead of one per ray.
#include "ngp_device.cuh"
#include "ngp_masks.cuh"

namespace ngp {

// What the reference's kernels take beyond the round-1 argument lists (NgpRenderExtras): crop masks, glow, envmap background, distortion map, quilting.
struct RenderExtras {
	const NgpMask3D* render_masks; uint32_t n_render_masks; int glow_mode; float glow_y_cutoff;
	const float* envmap; int envmap_res[2]; const float* distortion; int distortion_res[2]; int quilting_dims[2]; int render_mode; float4* frame_buffer;
	int row_begin, row_end;   // the rows this launch sets up (a shard of the frame, or 0 .. res[1])
	int tile_order;           // payload slots in 8 x 8 pixel tiles (one wave = one tile) instead of row-major
};

struct InitRaysArgs {
	uint32_t sample_index; NgpPayload* payloads; int32_t res[2]; float focal_length[2]; Mat34 cam0, cam1; float rolling_shutter[4];
	float screen_center[2]; float parallax_shift[3]; int snap_to_pixel_centers; Aabb render_aabb; Mat33 to_local; float near_distance;
	int lens_mode; float lens_params[7]; float* depthbuffer; float plane_z, aperture_size;
	int camera_model; float sq_width, sq_height, sq_curvature; float qh_front[12], qh_back[12];   // camera_models.cuh (0 = Perspective)
	RenderExtras ex;
};

__global__ void __launch_bounds__(128) init_rays_kernel(const InitRaysArgs a) {
	uint32_t x = threadIdx.x + blockDim.x * blockIdx.x, y = threadIdx.y + blockDim.y * blockIdx.y + (uint32_t)a.ex.row_begin;
	if (x >= (uint32_t)a.res[0] || y >= (uint32_t)a.ex.row_end) return;
	// idx: the pixel in the WHOLE frame (frame / depth buffer slot, key of every per-pixel random number); slot: its payload in this launch's ray array
	uint32_t idx = x + (uint32_t)a.res[0] * y, slot = idx - (uint32_t)a.res[0] * (uint32_t)a.ex.row_begin;
	// tile order: the launch's blocks are 8 x 8 pixels = one wave; slot = block-linear, so that a wave of every later kernel holds a SQUARE of neighbouring pixels (until
	// compaction thins it): their samples share hash-grid cells on the coarse and middle levels, a 64 x 1 strip of pixels does not
	if (a.ex.tile_order) slot = (blockIdx.y * gridDim.x + blockIdx.x) * 64u + threadIdx.y * 8u + threadIdx.x;
	float parallax_shift[3] = {a.parallax_shift[0], a.parallax_shift[1], a.parallax_shift[2]};
	const int qx = a.ex.quilting_dims[0], qy = a.ex.quilting_dims[1];
	if (qx != 1 || qy != 1) {   // apply_quilting (common_device.cuh:541-560): the pixel inside its panel, the panel's parallax
		const float resx = (float)a.res[0] / (float)qx, resy = (float)a.res[1] / (float)qy;
		const int panelx = (int)floorf((float)x / resx), panely = (int)floorf((float)y / resy);
		x = (uint32_t)((float)x - (float)panelx * resx);
		y = (uint32_t)((float)y - (float)panely * resy);
		const int pidx = panelx + qx * panely;
		if (qx == 2 && qy == 1) {
			parallax_shift[0] = pidx ? (-0.5f * parallax_shift[0]) : (0.5f * parallax_shift[0]);
		} else {
			const float max_parallax_angle = 17.5f;
			const float parallax_angle = max_parallax_angle * 3.14159265358979323846f / 180.f * (((float)pidx + 0.5f) * 2.f / (float)(qy * qx) - 1.f);
			parallax_shift[0] = atanf(parallax_angle) / parallax_shift[2];
		}
	}
	const float u = ((float)x + 0.5f) * (1.f / (float)a.res[0]), v = ((float)y + 0.5f) * (1.f / (float)a.res[1]);
	const float ray_time = a.rolling_shutter[0] + a.rolling_shutter[1] * u + a.rolling_shutter[2] * v + a.rolling_shutter[3] * ld_random_val(a.sample_index, idx * 72239731u);
	float cam[12];
#pragma unroll
	for (int k = 0; k < 12; ++k) cam[k] = a.cam0.m[k] * ray_time + a.cam1.m[k] * (1.f - ray_time);
	const int rx = a.res[0] / qx, ry = a.res[1] / qy;   // resolution.cwiseQuotient(quilting_dims) (1863)
	const float frx = (float)rx, fry = (float)ry;

	// pixel_to_ray (common_device.cuh:260-317)
	const float aperture_size = a.plane_z < 0 ? 0.0f : a.aperture_size;   // 1849-1851
	float ox, oy;
	ld_random_pixel_offset(a.snap_to_pixel_centers ? 0 : a.sample_index, ox, oy);
	const float pu = ((float)x + ox) / frx, pv = ((float)y + oy) / fry;
	v3 dir, origin;
	bool outside = false;
	if (a.camera_model != 0) {       // the fork's extra camera models (1868-1908); they ignore lens, parallax shift and screen centre
		if (a.camera_model == 2) spherical_quadrilateral_pixel_to_ray(a.sample_index, x, y, frx, fry, cam, a.sq_width, a.sq_height, a.sq_curvature, a.near_distance, a.plane_z, aperture_size, origin, dir);
		else quadrilateral_hexahedron_pixel_to_ray(a.sample_index, x, y, frx, fry, cam, a.qh_front, a.qh_back, a.near_distance, a.plane_z, aperture_size, origin, dir);
	} else {
	if (a.lens_mode == 2) {          // FTheta
		dir = f_theta_undistortion(pu - a.screen_center[0], pv - a.screen_center[1], a.lens_params, mk(1000.f, 0.f, 0.f));
		outside = dir.x == 1000.f;   // the reference returns a ray from (1000, 0, 0): outside the aabb, the pixel is not rendered
	} else if (a.lens_mode == 3) {   // LatLong
		dir = latlong_to_dir(pu, pv);
	} else {
		dir = mk((pu - a.screen_center[0]) * frx / a.focal_length[0], (pv - a.screen_center[1]) * fry / a.focal_length[1], 1.0f);
		if (a.lens_mode == 1) iterative_opencv_lens_undistortion(a.lens_params, dir.x, dir.y);
	}
	if (outside) {
		origin = mk(1000.f, 0.f, 0.f); dir = mk(0.f, 0.f, 1.f);
	} else {
		if (a.ex.distortion) {   // common_device.cuh:297-299
			float d0, d1;
			read_image2(a.ex.distortion, a.ex.distortion_res[0], a.ex.distortion_res[1], pu, pv, d0, d1);
			dir.x += d0; dir.y += d1;
		}
		const v3 head_pos = mk(parallax_shift[0], parallax_shift[1], 0.f);
		dir = dir - head_pos * parallax_shift[2];
		dir = mat3_mul(cam, dir);
		origin = mat3_mul(cam, head_pos) + col(cam, 3);
		apply_aperture(a.sample_index, x, y, cam, aperture_size, a.plane_z, origin, dir);   // depth of field (307-312)
		origin = origin + dir * a.near_distance;
	}
	}

	NgpPayload p = a.payloads[slot];
	p.max_weight = 0.0f;
	if (a.plane_z < 0) {   // slice plane (1913-1923): the ray stops at depth -plane_z along the view axis
		const float n = norm(dir);
		const v3 dn = dir * (1.0f / n);
		p.origin[0] = origin.x; p.origin[1] = origin.y; p.origin[2] = origin.z;
		p.dir[0] = dn.x; p.dir[1] = dn.y; p.dir[2] = dn.z;
		p.t = -a.plane_z * n; p.idx = idx; p.n_steps = 0; p.alive = 0;
		a.depthbuffer[idx] = -a.plane_z;
		a.payloads[slot] = p;
		return;
	}
	a.depthbuffer[idx] = 1e10f;
	dir = normalized(dir);
	if (a.ex.envmap) {   // 1931-1933
		float e[4];
		read_envmap(a.ex.envmap, a.ex.envmap_res[0], a.ex.envmap_res[1], dir, e);
		a.ex.frame_buffer[idx] = make_float4(e[0], e[1], e[2], e[3]);
	}
	float tmin, tmax;
	aabb_ray_intersect(a.render_aabb, mat3_mul(a.to_local.m, origin), mat3_mul(a.to_local.m, dir), tmin, tmax);
	const float t = fmaxf(tmin, 0.0f) + 1e-6f;
	p.origin[0] = origin.x; p.origin[1] = origin.y; p.origin[2] = origin.z;
	if (!aabb_contains(a.render_aabb, mat3_mul(a.to_local.m, origin + dir * t))) {
		p.alive = 0;
		a.payloads[slot] = p;
		return;
	}
	bool ray_intersects_any_mask = a.ex.n_render_masks == 0;   // 1943-1956
	for (uint32_t k = 0; k < a.ex.n_render_masks && !ray_intersects_any_mask; ++k) ray_intersects_any_mask = mask_intersects_ray(a.ex.render_masks[k], origin, dir);
	if (!ray_intersects_any_mask) {
		p.alive = 0;
		a.payloads[slot] = p;
		return;
	}
	if (a.ex.render_mode == 5) {   // Distortion (1959-1970): paint the map's offset at the pixel centre, the ray is done
		float d0 = 0.0f, d1 = 0.0f;
		if (a.ex.distortion) read_image2(a.ex.distortion, a.ex.distortion_res[0], a.ex.distortion_res[1], ((float)x + 0.5f) / (float)a.res[0], ((float)y + 0.5f) / (float)a.res[1], d0, d1);
		const v3 c = offset_to_rgb(d0 * 50.0f, d1 * 50.0f);
		a.ex.frame_buffer[idx] = make_float4(c.x, c.y, c.z, 1.0f);
		a.depthbuffer[idx] = 1.0f;
		const v3 far = origin + dir * 10000.0f;
		p.origin[0] = far.x; p.origin[1] = far.y; p.origin[2] = far.z;
		p.alive = 0;
		a.payloads[slot] = p;
		return;
	}
	p.dir[0] = dir.x; p.dir[1] = dir.y; p.dir[2] = dir.z;
	p.t = t; p.idx = idx; p.n_steps = 0; p.alive = 1;
	a.payloads[slot] = p;
}

struct Payload40 { uint32_t w[10]; };
static_assert(sizeof(NgpPayload) == sizeof(Payload40), "payload");

// compact_kernel_nerf (src/testbed_nerf.cu:1784-1807) at the END of the kernel that decides a ray's fate (advance_pos, composite) instead of as a pass of its own: the
// ray's payload, colour and depth are still in registers, so they are written once — to their compacted slot (alive), to the finished list (dead with alpha > 0.001)
// or nowhere — and the tracer loses a launch and a 60-byte read + write per ray and pass.  One atomic per wave and counter (ballots).  counter == NULL: off.
struct CompactOut { float4* dst_rgba; float* dst_depth; Payload40* dst_payloads; float4* fin_rgba; float* fin_depth; Payload40* fin_payloads; uint32_t* counter; uint32_t* final_counter;
                    uint32_t* blocks_done; uint32_t* host_mailbox; uint32_t sequence; };
// (every thread of the workgroup calls this; at most COMPACT_MAX_WAVES waves per workgroup.  The two counters are device-scope atomics on ONE address each, which the
// memory side serialises: one atomic per workgroup and counter — wave ballots summed through LDS — instead of one per wave.)
constexpr uint32_t COMPACT_MAX_WAVES = 4;
__device__ __forceinline__ void compact_store(bool alive, bool hit, const NgpPayload& payload, float4 c, float d, const CompactOut& co) {
	__shared__ uint32_t s_count[2][COMPACT_MAX_WAVES], s_base[2];
	const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63u) >> 6;
	const unsigned long long am = __ballot(alive), hm = __ballot(hit);
	if (lane == 0) { s_count[0][wave] = (uint32_t)__popcll(am); s_count[1][wave] = (uint32_t)__popcll(hm); }
	__syncthreads();
	if (threadIdx.x < 2) {
		uint32_t total = 0;
		for (uint32_t w = 0; w < n_waves; ++w) total += s_count[threadIdx.x][w];
		uint32_t base = total ? atomicAdd(threadIdx.x == 0 ? co.counter : co.final_counter, total) : 0u;
		s_base[threadIdx.x] = base;
		// The last workgroup to get here posts the pass's alive count to the caller's mailbox in host memory — one 8-byte store {count, sequence number the caller waits
		// for} — so the host learns it while the kernel drains, without a copy command and a stream synchronisation between two passes.  No fences (a device-scope release
		// writes the XCD's L2 back): the counter update above has returned, i.e. happened at the memory side, before the ticket is drawn, and all three are device-scope atomics.
		if (threadIdx.x == 0 && co.host_mailbox) {
			// The ticket's operand is computed FROM the counter atomic's return value by an instruction the compiler cannot fold (`base & 0u` it could): a true data
			// dependency, so the ticket is not issued before this workgroup's counter add has been performed at the memory side, and whoever draws the last ticket reads
			// a counter that holds every workgroup's add
			uint32_t zero;
			asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(base));
			if (__hip_atomic_fetch_add(co.blocks_done, 1u + zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
				const uint32_t n_alive = __hip_atomic_load(co.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(co.blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store((unsigned long long*)co.host_mailbox, (unsigned long long)n_alive | ((unsigned long long)co.sequence << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
	__syncthreads();
	uint32_t abase = s_base[0], hbase = s_base[1];
	for (uint32_t w = 0; w < wave; ++w) { abase += s_count[0][w]; hbase += s_count[1][w]; }
	const unsigned long long below = (1ull << lane) - 1ull;
	const Payload40& p = *(const Payload40*)&payload;
	if (alive) {
		const uint32_t idx = abase + (uint32_t)__popcll(am & below);
		co.dst_payloads[idx] = p; co.dst_rgba[idx] = c; co.dst_depth[idx] = d;
	} else if (hit) {
		const uint32_t idx = hbase + (uint32_t)__popcll(hm & below);
		co.fin_payloads[idx] = p; co.fin_rgba[idx] = c; co.fin_depth[idx] = d;
	}
}

// The 4 KiB cascade-0 brick summary (ngp_hip_bitfield_brick_summary) staged in LDS when the caller has one: the walk through empty space — most iterations of
// advance_pos, and the exit walk of every ray in generate_next_inputs — answers "empty" without a dependent global load per brick.  Same bits, same samples.
__device__ __forceinline__ void stage_brick_summary(const uint32_t* __restrict__ brick_summary, uint32_t* __restrict__ s_brick_any) {
	if (!brick_summary) return;   // (uniform)
	const uint4* src = (const uint4*)brick_summary;
	for (uint32_t q = threadIdx.x; q < NGP_NERF_GRID_N_CELLS / 64u / 32u / 4u; q += blockDim.x) ((uint4*)s_brick_any)[q] = src[q];
	__syncthreads();
}

template <bool CONST_DT>   // cone_angle == 0: see calc_dt_t
__global__ void __launch_bounds__(256) advance_pos_kernel(uint32_t n_elements, Aabb render_aabb, Mat33 to_local, uint32_t sample_index, NgpPayload* __restrict__ payloads,
                                   const uint8_t* __restrict__ density_grid, uint32_t min_mip, float cone_angle_constant, const CompactOut co, const uint32_t* __restrict__ brick_summary) {
	__shared__ __attribute__((aligned(16))) uint32_t s_brick_any[NGP_NERF_GRID_N_CELLS / 64 / 32];
	stage_brick_summary(brick_summary, s_brick_any);
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	NgpPayload payload;
	payload.alive = 0;
	if (i < n_elements) payload = payloads[i];
	const bool was_alive = i < n_elements && payload.alive;
	if (was_alive) {
		const v3 origin = ld3(payload.origin), dir = ld3(payload.dir);
		const v3 idir = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
		const float cone_angle = cone_angle_constant;
		float t = payload.t;
		float dt = calc_dt(t, cone_angle);
		t += ld_random_val(sample_index, payload.idx * 786433u) * dt;   // keyed by the pixel's index in the whole frame (= the ray's index when the frame is traced at once, :625): shards and tile-ordered rays jitter like the whole frame
		v3 pos;
		OccBrick occ;
		while (1) {
			pos = origin + dir * t;
			if (!aabb_contains(render_aabb, mat3_mul(to_local.m, pos))) { payload.alive = 0; break; }
			dt = calc_dt_t<CONST_DT>(t, cone_angle);
			uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
			mip = mip < min_mip ? min_mip : mip;
			if (!density_grid || (brick_summary ? density_grid_occupied_at(pos, density_grid, mip, occ, s_brick_any) : density_grid_occupied_at(pos, density_grid, mip, occ))) break;
			t = advance_to_next_voxel<CONST_DT>(t, cone_angle, pos, dir, idir, NGP_NERF_GRIDSIZE >> mip);
		}
		payload.t = t;
	}
	if (!co.counter) { if (was_alive) payloads[i] = payload; return; }
	// (a ray has no colour yet: zeros; a ray that died here has alpha 0 and is dropped)
	compact_store(was_alive && payload.alive, false, payload, make_float4(0.f, 0.f, 0.f, 0.f), 0.f, co);
}

__global__ void __launch_bounds__(256) compact_rays_kernel(uint32_t n_elements, const float4* __restrict__ src_rgba, const float* __restrict__ src_depth, const Payload40* __restrict__ src_payloads,
                                                           float4* __restrict__ dst_rgba, float* __restrict__ dst_depth, Payload40* __restrict__ dst_payloads,
                                                           float4* __restrict__ fin_rgba, float* __restrict__ fin_depth, Payload40* __restrict__ fin_payloads,
                                                           uint32_t* counter, uint32_t* final_counter) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const bool in_range = i < n_elements;
	Payload40 p; float4 c = make_float4(0, 0, 0, 0); float d = 0;
	bool alive = false, hit = false;
	if (in_range) {
		p = src_payloads[i]; c = src_rgba[i]; d = src_depth[i];
		alive = ((const NgpPayload*)&p)->alive != 0;
		hit = !alive && c.w > 0.001f;
	}
	const uint32_t lane = lane_id();
	const unsigned long long am = __ballot(alive), hm = __ballot(hit);
	uint32_t abase = 0, hbase = 0;
	if (lane == 0) {
		if (am) abase = atomicAdd(counter, (uint32_t)__popcll(am));
		if (hm) hbase = atomicAdd(final_counter, (uint32_t)__popcll(hm));
	}
	abase = __shfl(abase, 0, 64); hbase = __shfl(hbase, 0, 64);
	const unsigned long long below = (1ull << lane) - 1ull;
	if (alive) {
		const uint32_t idx = abase + (uint32_t)__popcll(am & below);
		dst_payloads[idx] = p; dst_rgba[idx] = c; dst_depth[idx] = d;
	} else if (hit) {
		const uint32_t idx = hbase + (uint32_t)__popcll(hm & below);
		fin_payloads[idx] = p; fin_rgba[idx] = c; fin_depth[idx] = d;
	}
}

// One flat loop per ray instead of the reference's "for every step: march until occupied": every iteration is one DDA iteration — emit a
// sample and step, or skip an empty voxel.  The per-ray sequence (and so every bit of the output) is the same, but a lane that leaves
// the object and walks ~150 empty voxels to the box boundary no longer holds the other 63 lanes of its wave at every one of the
// n_steps steps: the wave runs max-over-lanes(n_steps + skips) iterations, not sum-over-steps(max-over-lanes skips).
template <bool CONST_DT>
__global__ void __launch_bounds__(128) generate_next_inputs_kernel(uint32_t n_elements, Aabb render_aabb, Aabb train_aabb, NgpPayload* __restrict__ payloads, NgpCoord* __restrict__ network_input,
                                            uint32_t n_steps, const uint8_t* __restrict__ density_grid, uint32_t min_mip, float cone_angle_constant,
                                            const uint32_t* __restrict__ brick_summary, uint32_t* __restrict__ zero_word, uint32_t max_skips) {
	__shared__ __attribute__((aligned(16))) uint32_t s_brick_any[NGP_NERF_GRID_N_CELLS / 64 / 32];
	stage_brick_summary(brick_summary, s_brick_any);
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (zero_word && i == 0) *zero_word = 0;   // the caller's next compaction counter (stream order puts this before the kernel that bumps it): saves it a memset per pass
	if (i >= n_elements) return;
	NgpPayload& payload = payloads[i];
	if (!payload.alive) return;
	const v3 origin = ld3(payload.origin), dir = ld3(payload.dir);
	const v3 idir = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	const v3 wd = warp_direction(dir);
	const float cone_angle = cone_angle_constant;
	float t = payload.t;
	OccBrick occ;
	uint32_t j = 0, skips = 0;
	while (j < n_steps) {
		const v3 pos = origin + dir * t;
		if (!aabb_contains(render_aabb, pos)) { payload.n_steps = (uint16_t)j; return; }
		// A ray that has spent its allowance of empty voxels for this pass hands in the j samples it has and rests (alive = NGP_RAY_PAUSED, t kept): composite leaves it
		// alive and the next pass resumes the same march.  Without this every pass lasts as long as the longest walk to the box boundary among its rays.
		if (max_skips && skips >= max_skips) { payload.t = t; payload.n_steps = (uint16_t)j; payload.alive = NGP_RAY_PAUSED; return; }
		const float dt = calc_dt_t<CONST_DT>(t, cone_angle);
		uint32_t mip = (uint32_t)mip_from_dt(dt, pos);
		mip = mip < min_mip ? min_mip : mip;
		if (!density_grid || (brick_summary ? density_grid_occupied_at(pos, density_grid, mip, occ, s_brick_any) : density_grid_occupied_at(pos, density_grid, mip, occ))) {
			const v3 wp = aabb_relative_pos(train_aabb, pos);
			NgpCoord c;
			c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z; c.dt = warp_dt(dt); c.dir[0] = wd.x; c.dir[1] = wd.y; c.dir[2] = wd.z;
			network_input[i + (size_t)j * n_elements] = c;
			t += dt;
			++j;
		} else {
			t = advance_to_next_voxel<CONST_DT>(t, cone_angle, pos, dir, idir, NGP_NERF_GRIDSIZE >> mip);
			++skips;
		}
	}
	payload.t = t;
	payload.n_steps = (uint16_t)n_steps;
}

typedef uint16_t us4 __attribute__((ext_vector_type(4)));

// Glow / grid-line visualisation of composite_kernel_nerf (src/testbed_nerf.cu:843-939; the "#if 0" variant there is dead code)
__device__ __forceinline__ float glow_lines(v3 pos) {
	float line = 0.0f;
#pragma unroll
	for (int o = 0; o < 4; ++o) {
		const float f = (float)(2 << o);   // 2, 4, 8, 16
		line += fmaxf(0.f, cosf(pos.y * f * 3.141592653589793f * 16.f) - 0.975f);
		line += fmaxf(0.f, cosf(pos.x * f * 3.141592653589793f * 16.f) - 0.975f);
		line += fmaxf(0.f, cosf(pos.z * f * 3.141592653589793f * 16.f) - 0.975f);
	}
	return line;
}
__device__ __forceinline__ void glow_shading(int glow_mode, float glow_y_cutoff, v3 pos, v3 cam_pos, float& r, float& g, float& b, float& weight) {
	float glow = 0.f;
	const bool green_grid = glow_mode & 1, green_cutline = glow_mode & 2, mask_to_alpha = glow_mode & 4, radial_mode = glow_mode & 8, grid_mode = glow_mode & 16;
	float dist;
	if (radial_mode) {
		dist = norm(pos - cam_pos);
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
		if (mask_to_alpha) weight *= mask;
	}
	if (glow > 0.f) {
		const float line = glow_lines(pos);
		if (grid_mode) {
			glow = glow * line * 15.f;
			g = glow; b = glow * 0.5f; r = glow * 0.25f;
		} else {
			glow = glow * glow * 0.25f + glow * line * 15.f;
			g += glow; b += glow * 0.5f; r += glow * 0.25f;
		}
	}
}

__global__ void composite_kernel(uint32_t n_elements, uint32_t current_step, Aabb aabb, Mat34 camera_matrix, float4* __restrict__ rgba, float* __restrict__ depth,
                                 NgpPayload* __restrict__ payloads, const NgpCoord* __restrict__ network_input, const uint16_t* __restrict__ network_output, uint32_t out_stride,
                                 uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance, int render_mode, float depth_scale, int show_accel,
                                 const RenderExtras ex, const CompactOut co) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	NgpPayload payload;
	payload.alive = 0;
	if (i < n_elements) payload = payloads[i];
	const bool was_alive = i < n_elements && payload.alive;
	const bool paused = payload.alive == NGP_RAY_PAUSED;   // (ngp_hip_generate_next_inputs with an allowance of empty voxels)
	float4 local_rgba = make_float4(0.f, 0.f, 0.f, 0.f);
	float local_depth = 0.f;
	if (i < n_elements && (was_alive || co.counter)) { local_rgba = rgba[i]; local_depth = depth[i]; }
	if (was_alive) {
	const v3 ray_origin = ld3(payload.origin);
	const v3 cam_fwd = col(camera_matrix.m, 2), cam_pos = col(camera_matrix.m, 3);
	const uint32_t actual_n_steps = payload.n_steps;
	float max_weight = payload.max_weight;
	uint32_t j = 0;
	// The samples of a ray are one n_elements-strided load apart and the loop may stop at any of them, so the compiler issues each step's loads after the previous step's
	// exit test — n_steps dependent trips to memory.  Four steps' loads are issued together here (all inside [0, actual_n_steps): nothing is read that the plain loop could
	// not have read) and then consumed in order; same arithmetic, same exit.
	bool done = false, saturated = false;
	for (uint32_t j0 = 0; j0 < actual_n_steps && !done; j0 += 4) {
		us4 lo4[4]; NgpCoord in4[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			if (j0 + k < actual_n_steps) {
				const size_t s = (size_t)i + (size_t)(j0 + k) * n_elements;
				lo4[k] = __builtin_nontemporal_load((const us4*)(network_output + s * out_stride));
				in4[k] = network_input[s];
			}
		}
#pragma unroll
	for (uint32_t k = 0; k < 4; ++k) {
		if (j0 + k >= actual_n_steps) { done = true; break; }
		const us4 lo = lo4[k];
		const NgpCoord in = in4[k];
		const v3 pos = unwarp_position(mk(in.pos[0], in.pos[1], in.pos[2]), aabb);
		const float T = 1.f - local_rgba.w;
		const float dt = unwarp_dt(in.dt);
		float alpha = 1.f - __expf(-network_to_density(h2f(lo[3]), density_activation) * dt);
		if (show_accel >= 0) alpha = 1.f;   // 827-829
		float weight = alpha * T;
		float cr = network_to_rgb(h2f(lo[0]), rgb_activation), cg = network_to_rgb(h2f(lo[1]), rgb_activation), cb = network_to_rgb(h2f(lo[2]), rgb_activation);
		if (ex.n_render_masks) {   // crop masks (833-840)
			float mask_weight = 1.f;
			for (uint32_t k = 0; k < ex.n_render_masks; ++k) mask_weight = clampf(mask_weight + mask_sample(ex.render_masks[k], pos), 0.0f, 1.0f);
			weight *= mask_weight;
		}
		if (ex.glow_mode) glow_shading(ex.glow_mode, ex.glow_y_cutoff, pos, cam_pos, cr, cg, cb, weight);   // 843-939
		if (render_mode != 1) {   // visualisation modes (941-968); Cost / Slice composite like Shade
			if (render_mode == 2) {           // Normals: in.pos holds d(density output)/d(pos) (ngp_hip_nerf_input_gradient)
				const float k = -network_to_density_derivative(h2f(lo[3]), density_activation);
				const v3 nrm = normalized(mk(k * in.pos[0], k * in.pos[1], k * in.pos[2]));
				cr = nrm.x; cg = nrm.y; cb = nrm.z;
			} else if (render_mode == 8) {    // EncodingVis: in.pos holds the visualised activation (ngp_hip_nerf_visualize_activation)
				cr = in.pos[0]; cg = in.pos[1]; cb = in.pos[2];
			} else if (render_mode == 3) {    // Positions
				if (show_accel >= 0) {
					const int mp = mip_from_pos(pos);
					const uint32_t mip = (uint32_t)(show_accel > mp ? show_accel : mp);
					const uint32_t res = NGP_NERF_GRIDSIZE >> mip;
					const int ix = (int)(pos.x * (float)res), iy = (int)(pos.y * (float)res), iz = (int)(pos.z * (float)res);
					Pcg32 rng = pcg32_seeded((uint64_t)(int64_t)(ix + iy * 232323 + iz * 727272));
					cr = 1.f - (float)mip * (1.f / (float)(NGP_NERF_CASCADES - 1));
					cg = rng.next_float();
					cb = rng.next_float();
				} else {
					cr = (pos.x - 0.5f) / 2.0f + 0.5f; cg = (pos.y - 0.5f) / 2.0f + 0.5f; cb = (pos.z - 0.5f) / 2.0f + 0.5f;
				}
			} else if (render_mode == 4) {    // Depth
				cr = cg = cb = dot(cam_fwd, pos - ray_origin) * depth_scale;
			} else if (render_mode == 0) {    // AO
				cr = cg = cb = alpha;
			}
		}
		local_rgba.x += cr * weight;
		local_rgba.y += cg * weight;
		local_rgba.z += cb * weight;
		local_rgba.w += weight;
		if (weight > max_weight) { max_weight = weight; local_depth = dot(cam_fwd, pos - cam_pos); }
		if (local_rgba.w > (1.0f - min_transmittance)) {
			const float w = local_rgba.w;
			local_rgba.x /= w; local_rgba.y /= w; local_rgba.z /= w; local_rgba.w /= w;
			done = saturated = true;
			break;
		}
		++j;
	}
	}
	payload.max_weight = max_weight;
	if (saturated || (j < n_steps && !paused)) { payload.alive = 0; payload.n_steps = (uint16_t)(j + current_step); }
	else payload.alive = 1;
	}
	if (!co.counter) {
		if (was_alive) { payloads[i] = payload; rgba[i] = local_rgba; depth[i] = local_depth; }
		return;
	}
	const bool alive = i < n_elements && payload.alive;
	compact_store(alive, i < n_elements && !alive && local_rgba.w > 0.001f, payload, local_rgba, local_depth, co);
}

__global__ void shade_kernel(uint32_t n_elements, const float4* __restrict__ rgba, const float* __restrict__ depth, const NgpPayload* __restrict__ payloads,
                             bool train_in_linear_colors, float4* __restrict__ frame_buffer, float* __restrict__ depth_buffer, int render_mode) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	float4 tmp = rgba[i];
	if (render_mode == 2) {   // Normals (1764-1767): accumulated normal -> unit length -> [0, 1], premultiplied
		const v3 n = normalized(mk(tmp.x, tmp.y, tmp.z));
		tmp.x = (0.5f * n.x + 0.5f) * tmp.w; tmp.y = (0.5f * n.y + 0.5f) * tmp.w; tmp.z = (0.5f * n.z + 0.5f) * tmp.w;
	} else if (render_mode == 6) { const float c = (float)payloads[i].n_steps / 128; tmp = make_float4(c, c, c, 1.0f); }   // Cost
	if (!train_in_linear_colors && (render_mode == 1 || render_mode == 7)) { tmp.x = srgb_to_linear(tmp.x); tmp.y = srgb_to_linear(tmp.y); tmp.z = srgb_to_linear(tmp.z); }
	const uint32_t idx = payloads[i].idx;
	const float4 fb = frame_buffer[idx];
	const float k = 1.0f - tmp.w;
	frame_buffer[idx] = make_float4(tmp.x + fb.x * k, tmp.y + fb.y * k, tmp.z + fb.z * k, tmp.w + fb.w * k);
	if (render_mode != 7 && tmp.w > 0.2f) depth_buffer[idx] = depth[i];
}

// Slice mode (2445-2476): the network evaluated where every ray meets the slice plane
__global__ void generate_inputs_at_current_position_kernel(uint32_t n_elements, Aabb aabb, const NgpPayload* __restrict__ payloads, NgpCoord* __restrict__ network_input) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const v3 dir = ld3(payloads[i].dir), origin = ld3(payloads[i].origin);
	const v3 wp = aabb_relative_pos(aabb, origin + dir * payloads[i].t), wd = warp_direction(dir);
	NgpCoord c;
	c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z; c.dt = warp_dt(MIN_CONE_STEPSIZE()); c.dir[0] = wd.x; c.dir[1] = wd.y; c.dir[2] = wd.z;
	network_input[i] = c;
}
__global__ void compute_nerf_rgba_kernel(uint32_t n_elements, const uint16_t* __restrict__ network_output, uint32_t out_stride, float4* __restrict__ rgba, int rgb_activation,
                                         int density_activation, float depth, bool density_as_alpha) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const us4 lo = *(const us4*)(network_output + (size_t)i * out_stride);
	const float density = network_to_density(h2f(lo[3]), density_activation);
	float alpha = 1.f, w;
	if (density_as_alpha) w = density;
	else w = alpha = clampf(1.f - __expf(-density * depth), 0.0f, 1.0f);
	rgba[i] = make_float4(network_to_rgb(h2f(lo[0]), rgb_activation) * alpha, network_to_rgb(h2f(lo[1]), rgb_activation) * alpha, network_to_rgb(h2f(lo[2]), rgb_activation) * alpha, w);
}

__global__ void accumulate_kernel(uint32_t n, const float4* __restrict__ frame_buffer, float4* __restrict__ accumulate_buffer, float sample_count, int color_space) {
	const uint32_t idx = threadIdx.x + blockIdx.x * blockDim.x;
	if (idx >= n) return;
	float4 color = frame_buffer[idx];
	float4 tmp = accumulate_buffer[idx];
	if (color_space == NGP_COLOR_SRGB) { color.x = linear_to_srgb(color.x); color.y = linear_to_srgb(color.y); color.z = linear_to_srgb(color.z); }
	tmp.x = (tmp.x * sample_count + color.x) / (sample_count + 1);
	tmp.y = (tmp.y * sample_count + color.y) / (sample_count + 1);
	tmp.z = (tmp.z * sample_count + color.z) / (sample_count + 1);
	tmp.w = (tmp.w * sample_count + color.w) / (sample_count + 1);
	accumulate_buffer[idx] = tmp;
}

__device__ __forceinline__ void tonemap_curve(float x[3], int curve) {
	if (curve == NGP_TONEMAP_IDENTITY) return;
#pragma unroll
	for (int c = 0; c < 3; ++c) x[c] = fmaxf(x[c], 0.f);
	float k0, k1, k2, k3, k4, k5;
	if (curve == NGP_TONEMAP_ACES) {
		k0 = 0.6f * 0.6f * 2.51f; k1 = 0.6f * 0.03f; k2 = 0.0f; k3 = 0.6f * 0.6f * 2.43f; k4 = 0.6f * 0.59f; k5 = 0.14f;
	} else if (curve == NGP_TONEMAP_HABLE) {
		const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
		k0 = A * F - A * E; k1 = C * B * F - B * E; k2 = 0.0f; k3 = A * F; k4 = B * F; k5 = D * F * F;
		const float W = 11.2f;
		const float nom = k0 * (W * W) + k1 * W + k2, denom = k3 * (W * W) + k4 * W + k5;
		const float white_scale = denom / nom;
		k0 = 4.0f * k0 * white_scale; k1 = 2.0f * k1 * white_scale; k2 = k2 * white_scale; k3 = 4.0f * k3; k4 = 2.0f * k4;
	} else {
		const float Y = 0.2126f * x[0] + 0.7152f * x[1] + 0.0722f * x[2];
#pragma unroll
		for (int c = 0; c < 3; ++c) x[c] = x[c] * (1.f / (Y + 1.0f));
		return;
	}
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const float sq = x[c] * x[c];
		x[c] = (sq * k0 + k1 * x[c] + k2) / (k3 * sq + k4 * x[c] + k5);
	}
}

__global__ void tonemap_kernel(uint32_t n, float exposure, float4 background_color, const float4* __restrict__ accumulate_buffer, int color_space, int output_color_space,
                               int curve, bool clamp_output_color, float4* __restrict__ surface) {
	const uint32_t idx = threadIdx.x + blockIdx.x * blockDim.x;
	if (idx >= n) return;
	if (color_space != NGP_COLOR_SRGB) { background_color.x = srgb_to_linear(background_color.x); background_color.y = srgb_to_linear(background_color.y); background_color.z = srgb_to_linear(background_color.z); }
	float4 color = accumulate_buffer[idx];
	const float weight = (1 - color.w) * background_color.w;
	float c3[3] = {color.x + background_color.x * weight, color.y + background_color.y * weight, color.z + background_color.z * weight};
	color.w += weight;
	if (color_space == NGP_COLOR_SRGB) { for (int c = 0; c < 3; ++c) c3[c] = srgb_to_linear(c3[c]); }
	const float e = powf(2.0f, exposure);
	for (int c = 0; c < 3; ++c) c3[c] *= e;
	tonemap_curve(c3, curve);
	if (output_color_space == NGP_COLOR_SRGB) { for (int c = 0; c < 3; ++c) c3[c] = linear_to_srgb(c3[c]); }
	if (clamp_output_color) { for (int c = 0; c < 3; ++c) c3[c] = fminf(fmaxf(c3[c], 0.0f), 1.0f); color.w = fminf(fmaxf(color.w, 0.0f), 1.0f); }
	surface[idx] = make_float4(c3[0], c3[1], c3[2], color.w);
}

static CompactOut compact_from_host(const NgpCompactOut* c) {
	CompactOut o{};
	if (!c || !c->counter) return o;
	o.dst_rgba = (float4*)c->dst_rgba; o.dst_depth = c->dst_depth; o.dst_payloads = (Payload40*)c->dst_payloads;
	o.fin_rgba = (float4*)c->dst_final_rgba; o.fin_depth = c->dst_final_depth; o.fin_payloads = (Payload40*)c->dst_final_payloads;
	o.counter = c->counter; o.final_counter = c->final_counter;
	o.blocks_done = c->blocks_done; o.host_mailbox = c->blocks_done ? c->host_mailbox : nullptr; o.sequence = c->sequence;
	return o;
}
static Mat34 mat34_from_host(const float* m) { Mat34 r; for (int i = 0; i < 12; ++i) r.m[i] = m[i]; return r; }
static Mat33 mat33_from_host(const float* m) { Mat33 r; for (int i = 0; i < 9; ++i) r.m[i] = m ? m[i] : ((i % 4 == 0) ? 1.0f : 0.0f); return r; }

} // namespace ngp

using namespace ngp;

extern "C" {

static int extras_from_host(const NgpRenderExtras* e, RenderExtras& x, const char* who) {
	x.render_masks = nullptr; x.n_render_masks = 0; x.glow_mode = 0; x.glow_y_cutoff = 0.f; x.envmap = nullptr; x.envmap_res[0] = x.envmap_res[1] = 0;
	x.distortion = nullptr; x.distortion_res[0] = x.distortion_res[1] = 0; x.quilting_dims[0] = x.quilting_dims[1] = 1; x.render_mode = 1; x.frame_buffer = nullptr;
	x.row_begin = x.row_end = 0; x.tile_order = 0;
	if (!e) return 0;
	x.row_begin = e->row_begin; x.row_end = e->row_end; x.tile_order = e->tile_order;
	x.render_masks = e->n_render_masks ? e->render_masks : nullptr; x.n_render_masks = e->render_masks ? e->n_render_masks : 0;
	x.glow_mode = e->glow_mode; x.glow_y_cutoff = e->glow_y_cutoff;
	if (e->envmap && e->envmap_res[0] > 0 && e->envmap_res[1] > 0) { x.envmap = e->envmap; x.envmap_res[0] = e->envmap_res[0]; x.envmap_res[1] = e->envmap_res[1]; }
	if (e->distortion && e->distortion_res[0] > 0 && e->distortion_res[1] > 0) { x.distortion = e->distortion; x.distortion_res[0] = e->distortion_res[0]; x.distortion_res[1] = e->distortion_res[1]; }
	if (e->quilting_dims[0] > 0 && e->quilting_dims[1] > 0) { x.quilting_dims[0] = e->quilting_dims[0]; x.quilting_dims[1] = e->quilting_dims[1]; }
	x.render_mode = e->render_mode; x.frame_buffer = (float4*)e->frame_buffer;
	if ((x.envmap || x.render_mode == 5) && !x.frame_buffer) { set_last_error(who, hipErrorInvalidValue); return -1; }
	return 0;
}

int ngp_hip_init_rays(void* stream, uint32_t sample_index, NgpPayload* payloads, const int32_t* res_host, const float* focal_length_host,
                      const float* camera_matrix0_host, const float* camera_matrix1_host, const float* rolling_shutter_host,
                      const float* screen_center_host, const float* parallax_shift_host, int snap_to_pixel_centers, const NgpAabb* render_aabb_host,
                      const float* render_aabb_to_local_host, float near_distance, int lens_mode, const float* lens_params_host, float* depthbuffer,
                      float plane_z, float aperture_size, const NgpRenderCamera* camera_models_host, const NgpRenderExtras* extras_host) {
	InitRaysArgs a;
	if (extras_from_host(extras_host, a.ex, "ngp_hip_init_rays: an envmap or the Distortion mode needs extras->frame_buffer")) return -1;
	if (a.ex.row_begin == 0 && a.ex.row_end == 0) a.ex.row_end = res_host[1];
	if (a.ex.row_begin < 0 || a.ex.row_end > res_host[1] || a.ex.row_begin > a.ex.row_end) { set_last_error("ngp_hip_init_rays: row range outside the frame", hipErrorInvalidValue); return -1; }
	if (a.ex.row_begin == a.ex.row_end) return 0;
	a.plane_z = plane_z; a.aperture_size = aperture_size;
	a.camera_model = camera_models_host ? camera_models_host->model : 0;
	a.sq_width = a.sq_height = a.sq_curvature = 0.f;
	for (int i = 0; i < 12; ++i) { a.qh_front[i] = 0.f; a.qh_back[i] = 0.f; }
	if (camera_models_host) {
		a.sq_width = camera_models_host->sq_width; a.sq_height = camera_models_host->sq_height; a.sq_curvature = camera_models_host->sq_curvature;
		for (int i = 0; i < 12; ++i) { a.qh_front[i] = camera_models_host->qh_front[i]; a.qh_back[i] = camera_models_host->qh_back[i]; }
	}
	a.sample_index = sample_index; a.payloads = payloads; a.res[0] = res_host[0]; a.res[1] = res_host[1];
	a.focal_length[0] = focal_length_host[0]; a.focal_length[1] = focal_length_host[1];
	a.cam0 = mat34_from_host(camera_matrix0_host); a.cam1 = mat34_from_host(camera_matrix1_host);
	for (int i = 0; i < 4; ++i) a.rolling_shutter[i] = rolling_shutter_host ? rolling_shutter_host[i] : 0.0f;
	a.screen_center[0] = screen_center_host[0]; a.screen_center[1] = screen_center_host[1];
	for (int i = 0; i < 3; ++i) a.parallax_shift[i] = parallax_shift_host ? parallax_shift_host[i] : 0.0f;
	a.snap_to_pixel_centers = snap_to_pixel_centers; a.render_aabb = aabb_from_host(render_aabb_host); a.to_local = mat33_from_host(render_aabb_to_local_host);
	a.near_distance = near_distance; a.lens_mode = lens_mode;
	for (int i = 0; i < 7; ++i) a.lens_params[i] = lens_params_host ? lens_params_host[i] : 0.0f;
	a.depthbuffer = depthbuffer;
	if (a.ex.tile_order && ((res_host[0] & 7) || ((a.ex.row_end - a.ex.row_begin) & 7))) { set_last_error("ngp_hip_init_rays: tile_order needs a width and a row count that are multiples of 8", hipErrorInvalidValue); return -1; }
	const dim3 threads = a.ex.tile_order ? dim3(8, 8, 1) : dim3(16, 8, 1);
	const dim3 blocks(div_up((uint32_t)res_host[0], threads.x), div_up((uint32_t)(a.ex.row_end - a.ex.row_begin), 8), 1);
	hipLaunchKernelGGL(init_rays_kernel, blocks, threads, 0, (hipStream_t)stream, a);
	NGP_LAUNCH_CHECK("init_rays_kernel");
	return 0;
}

int ngp_hip_advance_pos(void* stream, uint32_t n_elements, const NgpAabb* render_aabb_host, const float* render_aabb_to_local_host, uint32_t sample_index,
                        NgpPayload* payloads, const uint8_t* density_grid, uint32_t min_mip, float cone_angle_constant, const NgpCompactOut* compact_host, const uint32_t* brick_summary) {
	if (!n_elements) return 0;
	const CompactOut co = compact_from_host(compact_host);
	if (cone_angle_constant == 0.0f) hipLaunchKernelGGL(advance_pos_kernel<true>, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, aabb_from_host(render_aabb_host), mat33_from_host(render_aabb_to_local_host), sample_index, payloads, density_grid, min_mip, cone_angle_constant, co, brick_summary);
	else hipLaunchKernelGGL(advance_pos_kernel<false>, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, aabb_from_host(render_aabb_host), mat33_from_host(render_aabb_to_local_host), sample_index, payloads, density_grid, min_mip, cone_angle_constant, co, brick_summary);
	NGP_LAUNCH_CHECK("advance_pos_kernel");
	return 0;
}

int ngp_hip_compact_rays(void* stream, uint32_t n_elements, const float* src_rgba, const float* src_depth, const NgpPayload* src_payloads, float* dst_rgba, float* dst_depth,
                         NgpPayload* dst_payloads, float* dst_final_rgba, float* dst_final_depth, NgpPayload* dst_final_payloads, uint32_t* counter, uint32_t* final_counter) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(compact_rays_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, (const float4*)src_rgba, src_depth, (const Payload40*)src_payloads,
	                   (float4*)dst_rgba, dst_depth, (Payload40*)dst_payloads, (float4*)dst_final_rgba, dst_final_depth, (Payload40*)dst_final_payloads, counter, final_counter);
	NGP_LAUNCH_CHECK("compact_rays_kernel");
	return 0;
}

int ngp_hip_generate_next_inputs(void* stream, uint32_t n_elements, const NgpAabb* render_aabb_host, const NgpAabb* train_aabb_host, NgpPayload* payloads, NgpCoord* network_input,
                                 uint32_t n_steps, const uint8_t* density_grid, uint32_t min_mip, float cone_angle_constant, uint32_t max_skips_per_pass, const uint32_t* brick_summary, uint32_t* zero_word) {
	if (!n_elements) { if (zero_word) NGP_HIP_TRY(hipMemsetAsync(zero_word, 0, 4, (hipStream_t)stream)); return 0; }
	if (cone_angle_constant == 0.0f) hipLaunchKernelGGL(generate_next_inputs_kernel<true>, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, aabb_from_host(render_aabb_host), aabb_from_host(train_aabb_host), payloads, network_input, n_steps, density_grid, min_mip, cone_angle_constant, brick_summary, zero_word, max_skips_per_pass);
	else hipLaunchKernelGGL(generate_next_inputs_kernel<false>, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, aabb_from_host(render_aabb_host), aabb_from_host(train_aabb_host), payloads, network_input, n_steps, density_grid, min_mip, cone_angle_constant, brick_summary, zero_word, max_skips_per_pass);
	NGP_LAUNCH_CHECK("generate_next_inputs_kernel");
	return 0;
}

int ngp_hip_composite(void* stream, uint32_t n_elements, uint32_t current_step, const NgpAabb* aabb_host, const float* camera_matrix_host, float* rgba, float* depth,
                      NgpPayload* payloads, const NgpCoord* network_input, const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation,
                      int density_activation, float min_transmittance, int render_mode, float depth_scale, int show_accel, const NgpRenderExtras* extras_host, const NgpCompactOut* compact_host) {
	if (render_mode < 0 || render_mode > 8) { set_last_error("ngp_hip_composite: render mode out of range (ERenderMode 0..7, EncodingVis 8)", hipErrorInvalidValue); return -1; }
	RenderExtras ex;
	extras_from_host(nullptr, ex, "");
	if (extras_host) { RenderExtras t; extras_from_host(extras_host, t, ""); ex.render_masks = t.render_masks; ex.n_render_masks = t.n_render_masks; ex.glow_mode = t.glow_mode; ex.glow_y_cutoff = t.glow_y_cutoff; }
	if (!n_elements) return 0;
	hipLaunchKernelGGL(composite_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, current_step, aabb_from_host(aabb_host), mat34_from_host(camera_matrix_host), (float4*)rgba, depth,
	                   payloads, network_input, network_output, out_stride, n_steps, rgb_activation, density_activation, min_transmittance, render_mode, depth_scale, show_accel, ex, compact_from_host(compact_host));
	NGP_LAUNCH_CHECK("composite_kernel");
	return 0;
}

int ngp_hip_shade(void* stream, uint32_t n_elements, const float* rgba, const float* depth, const NgpPayload* payloads, int train_in_linear_colors, float* frame_buffer, float* depth_buffer,
                  int render_mode) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(shade_kernel, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, (const float4*)rgba, depth, payloads, train_in_linear_colors != 0, (float4*)frame_buffer, depth_buffer, render_mode);
	NGP_LAUNCH_CHECK("shade_kernel");
	return 0;
}

int ngp_hip_generate_inputs_at_current_position(void* stream, uint32_t n_elements, const NgpAabb* aabb_host, const NgpPayload* payloads, NgpCoord* network_input) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(generate_inputs_at_current_position_kernel, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, aabb_from_host(aabb_host), payloads, network_input);
	NGP_LAUNCH_CHECK("generate_inputs_at_current_position_kernel");
	return 0;
}
int ngp_hip_compute_nerf_rgba(void* stream, uint32_t n_elements, const uint16_t* network_output, uint32_t out_stride, float* rgba, int rgb_activation, int density_activation,
                              float depth, int density_as_alpha) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(compute_nerf_rgba_kernel, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, network_output, out_stride, (float4*)rgba, rgb_activation, density_activation,
	                   depth, density_as_alpha != 0);
	NGP_LAUNCH_CHECK("compute_nerf_rgba_kernel");
	return 0;
}

int ngp_hip_accumulate(void* stream, const int32_t* res_host, const float* frame_buffer, float* accumulate_buffer, float sample_count, int color_space) {
	const uint32_t n = (uint32_t)res_host[0] * (uint32_t)res_host[1];
	if (!n) return 0;
	hipLaunchKernelGGL(accumulate_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, (const float4*)frame_buffer, (float4*)accumulate_buffer, sample_count, color_space);
	NGP_LAUNCH_CHECK("accumulate_kernel");
	return 0;
}

int ngp_hip_tonemap(void* stream, const int32_t* res_host, float exposure, const float* background_color_host, const float* accumulate_buffer, int color_space, int output_color_space,
                    int tonemap_curve, int clamp_output_color, float* surface) {
	const uint32_t n = (uint32_t)res_host[0] * (uint32_t)res_host[1];
	if (!n) return 0;
	const float4 bg = make_float4(background_color_host[0], background_color_host[1], background_color_host[2], background_color_host[3]);
	hipLaunchKernelGGL(tonemap_kernel, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, exposure, bg, (const float4*)accumulate_buffer, color_space, output_color_space, tonemap_curve, clamp_output_color != 0, (float4*)surface);
	NGP_LAUNCH_CHECK("tonemap_kernel");
	return 0;
}

} // extern "C"
