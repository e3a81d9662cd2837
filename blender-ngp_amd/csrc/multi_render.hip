// multi_render.hip — the Blender add-on's multi-NeRF renderer kernels for gfx950.
// Replaces src/nerf_renderer.cu:17-563 (init_global_rays, init_proxy_rays, hit_test_and_march, compact_rays, march_active_rays,
// march_proxy_rays_and_generate_next_network_inputs, cull_global_rays_and_set_proxy_rays_active, composite_proxy_ray_colors,
// shade_buffer_with_rays), src/nerf_utils.cu (runtime-parameter marching helpers), include/.../nerf/mask_3D.cuh (SDF masks) and
// include/.../camera_models.cuh (perspective / spherical-quadrilateral / quadrilateral-hexahedron cameras).
// One thread per ray like the reference; compaction uses wave64 ballots (one atomic per wave and counter).
#include "ngp_device.cuh"
#include "ngp_masks.cuh"

namespace ngp {

__device__ __forceinline__ Aabb aabb_of(const NgpAabb& a) { Aabb b; b.mn = ld3(a.min); b.mx = ld3(a.max); return b; }

// ---- nerf_utils.cu
__device__ __forceinline__ float get_dt(float t, float cone_angle, float min_step, float max_step) { return clampf(t * cone_angle, min_step, max_step); }
__device__ __forceinline__ int get_mip_from_dt(float dt, v3 pos, uint32_t grid_size, uint32_t max_cascade) {
	int mip = mip_from_pos(pos, max_cascade);
	dt *= (float)(2u * grid_size);
	if (dt < 1.f) return mip;
	const int e = frexp_exponent(dt);
	const int m = e > mip ? e : mip;
	return (int)max_cascade < m ? (int)max_cascade : m;
}
__device__ __forceinline__ uint32_t get_cascaded_grid_idx_at(v3 pos, uint32_t mip, uint32_t grid_size) {
	const float mip_scale = __uint_as_float((127u - mip) << 23);
	pos.x -= 0.5f; pos.y -= 0.5f; pos.z -= 0.5f;
	pos.x *= mip_scale; pos.y *= mip_scale; pos.z *= mip_scale;
	pos.x += 0.5f; pos.y += 0.5f; pos.z += 0.5f;
	const float g = (float)grid_size;
	const int ix = (int)(pos.x * g), iy = (int)(pos.y * g), iz = (int)(pos.z * g);
	const int hi = (int)grid_size - 1;
	return morton3D((uint32_t)clampi(ix, 0, hi), (uint32_t)clampi(iy, 0, hi), (uint32_t)clampi(iz, 0, hi));
}
// same bit as bitfield[idx / 8 + grid_volume * mip / 8] & (1 << idx % 8), read through the last 4x4x4 Morton brick (64 consecutive bits)
// kept in two registers: consecutive steps of a ray mostly stay inside one brick, and the march is a chain of dependent loads otherwise
__device__ __forceinline__ bool get_is_occupied(v3 pos, const uint8_t* __restrict__ bitfield, uint32_t mip, uint32_t grid_size, uint32_t grid_volume, OccBrick& cache) {
	const uint32_t idx = get_cascaded_grid_idx_at(pos, mip, grid_size);
	const uint32_t brick = (idx >> 6) + (grid_volume / 64u) * mip;
	if (brick != cache.id) { cache.id = brick; cache.bits = ((const uint64_t*)bitfield)[brick]; }
	return (cache.bits >> (idx & 63u)) & 1ull;
}
__device__ __forceinline__ float get_t_advanced_to_next_voxel(float t, float cone_angle, v3 pos, v3 dir, v3 idir, uint32_t res, float min_step, float max_step) {
	const float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	do { t += get_dt(t, cone_angle, min_step, max_step); } while (t < t_target);
	return t;
}
__device__ __forceinline__ float get_warped_dt(float dt, float min_step, uint32_t nerf_cascades) {
	const float max_stepsize = min_step * (float)(1u << (nerf_cascades - 1));
	return (dt - min_step) / (max_stepsize - min_step);
}
__device__ __forceinline__ float get_unwarped_dt(float dt, float min_step, uint32_t nerf_cascades) {
	const float max_stepsize = min_step * (float)(1u << (nerf_cascades - 1));
	return dt * (max_stepsize - min_step) + min_step;
}


// ---- camera_models.cuh

__global__ void __launch_bounds__(128) init_global_rays_kernel(uint32_t sample_index, NgpGlobalRay* __restrict__ rays, float* __restrict__ depthbuffer, const NgpDownsampleInfo ds, const NgpRenderCamera cam) {
	uint32_t x = threadIdx.x + blockDim.x * blockIdx.x, y = threadIdx.y + blockDim.y * blockIdx.y;
	const uint32_t idx = x + (uint32_t)ds.scaled_res[0] * y;
	x *= (uint32_t)ds.skip[0]; y *= (uint32_t)ds.skip[1];
	if (x >= (uint32_t)ds.max_res[0] || y >= (uint32_t)ds.max_res[1]) return;
	const float rx = (float)ds.max_res[0], ry = (float)ds.max_res[1];
	v3 origin = mk(0.f, 0.f, 0.f), dir = mk(0.f, 0.f, 1.f);
	const float* c = cam.transform;
	if (cam.model == 0) {  // perspective_pixel_to_ray (camera_models.cuh:206-241)
		float ox, oy;
		ld_random_pixel_offset(sample_index, ox, oy);
		const float u = ((float)x + ox) / rx, v = ((float)y + oy) / ry;
		dir = mk((u - 0.5f) * rx / cam.focal_length, (v - 0.5f) * ry / cam.focal_length, 1.0f);
		dir = mat3_mul(c, dir);
		origin = mat3_mul(c, mk(0.f, 0.f, 0.f)) + col(c, 3);
		apply_aperture(sample_index, x, y, c, cam.aperture_size, cam.focus_z, origin, dir);
		origin = origin + dir * cam.near_distance;
	} else if (cam.model == 2) {
		spherical_quadrilateral_pixel_to_ray(sample_index, x, y, rx, ry, c, cam.sq_width, cam.sq_height, cam.sq_curvature, cam.near_distance, cam.focus_z, cam.aperture_size, origin, dir);
	} else {
		quadrilateral_hexahedron_pixel_to_ray(sample_index, x, y, rx, ry, c, cam.qh_front, cam.qh_back, cam.near_distance, cam.focus_z, cam.aperture_size, origin, dir);
	}
	depthbuffer[idx] = 1e10f;
	NgpGlobalRay ray;
	dir = normalized(dir);
	ray.origin[0] = origin.x; ray.origin[1] = origin.y; ray.origin[2] = origin.z;
	ray.dir[0] = dir.x; ray.dir[1] = dir.y; ray.dir[2] = dir.z;
	ray.rgba[0] = ray.rgba[1] = ray.rgba[2] = ray.rgba[3] = 0.0f;
	ray.idx = idx; ray.depth = 0.0f; ray.alive = 1; ray.pad_[0] = ray.pad_[1] = ray.pad_[2] = 0;
	rays[idx] = ray;
}

__global__ void init_proxy_rays_kernel(uint32_t n_elements, const NgpGlobalRay* __restrict__ global_rays, NgpProxyRay* __restrict__ proxy_rays, const NgpNerfProps* __restrict__ props) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const NgpGlobalRay& g = global_rays[i];
	NgpProxyRay& p = proxy_rays[g.idx];
	if (!g.alive) { p.alive = 0; p.origin[0] = p.origin[1] = p.origin[2] = 0.0f; return; }
	const Aabb render_aabb = aabb_of(props->render_aabb);
	const v3 origin = xform_point(props->itransform, ld3(g.origin));
	const v3 dir = normalized(xform_dir(props->itransform, normalized(ld3(g.dir))));
	p.dir[0] = dir.x; p.dir[1] = dir.y; p.dir[2] = dir.z;
	float tmin, tmax;
	aabb_ray_intersect(render_aabb, origin, dir, tmin, tmax);
	const float t = fmaxf(tmin, 0.0f) + 1e-5f;
	if (!aabb_contains(render_aabb, origin + dir * t)) { p.alive = 0; return; }
	bool hits = props->n_masks == 0;
	if (!hits) {
		for (uint32_t k = 0; k < props->n_masks; ++k) {
			if (mask_intersects_ray(props->masks[k], origin, dir)) { hits = true; break; }
		}
	}
	p.active = 1;
	p.alive = hits ? 1 : 0;
	p.idx = g.idx;
	p.t = 0.0f;
	p.n_steps = 0;
	const v3 o2 = origin + dir * t;
	p.origin[0] = o2.x; p.origin[1] = o2.y; p.origin[2] = o2.z;
}

// nerf_renderer.cu:148-208.  The mask test inside the loop is unreachable in the reference (a `break` precedes it), so occupancy alone decides.
__device__ __forceinline__ bool hit_test_and_march(v3 origin, v3 dir, v3 idir, float proxy_t, const NgpNerfProps* __restrict__ props, float* t_out, float* dt_out, OccBrick& occ) {
	const Aabb render_aabb = aabb_of(props->render_aabb);
	const float cone = props->cone_angle, mn = props->min_cone_stepsize, mx = props->max_cone_stepsize;
	const uint8_t* __restrict__ bitfield = props->density_grid_bitfield;
	const uint32_t grid_size = props->grid_size, grid_volume = props->grid_volume, max_cascade = props->nerf_cascades - 1;
	float t = proxy_t, dt = 0.0f, prev_t = t;
	while (1) {
		const v3 pos = origin + dir * t;
		if (!aabb_contains(render_aabb, pos)) {
			if (t_out) *t_out = prev_t;
			if (dt_out) *dt_out = dt;
			return false;
		}
		dt = get_dt(t, cone, mn, mx);
		int mipi = get_mip_from_dt(dt, pos, grid_size, max_cascade);
		const uint32_t mip = (uint32_t)(mipi < 0 ? 0 : mipi);
		if (!bitfield) break;
		if (get_is_occupied(pos, bitfield, mip, grid_size, grid_volume, occ)) break;
		const uint32_t res = grid_size >> mip;
		prev_t = t;
		t = get_t_advanced_to_next_voxel(t, cone, pos, dir, idir, res, mn, mx);
	}
	if (t_out) *t_out = t;
	if (dt_out) *dt_out = dt;
	return true;
}

__global__ void multi_compact_rays_kernel(uint32_t n_elements, const NgpGlobalRay* __restrict__ g_src, NgpGlobalRay* __restrict__ g_dst, const NgpProxyRay* __restrict__ p_src,
                                          NgpProxyRay* __restrict__ p_dst, uint32_t n_nerfs, uint32_t stride, NgpGlobalRay* __restrict__ g_final, uint32_t* alive_counter, uint32_t* final_counter) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	const bool in = i < n_elements;
	NgpGlobalRay g;
	bool alive = false, fin = false;
	if (in) { g = g_src[i]; alive = g.alive != 0; fin = !alive && g.rgba[3] > 0.001f; }
	const uint64_t m_alive = __ballot(alive), m_fin = __ballot(fin);
	const uint32_t lane = lane_id();
	uint32_t base_a = 0, base_f = 0;
	if (lane == 0) {
		if (m_alive) base_a = atomicAdd(alive_counter, (uint32_t)__popcll(m_alive));
		if (m_fin) base_f = atomicAdd(final_counter, (uint32_t)__popcll(m_fin));
	}
	base_a = __shfl(base_a, 0, 64); base_f = __shfl(base_f, 0, 64);
	const uint64_t below = (1ull << lane) - 1ull;
	if (alive) {
		const uint32_t idx = base_a + (uint32_t)__popcll(m_alive & below);
		g_dst[idx] = g;
		for (uint32_t n = 0; n < n_nerfs; ++n) p_dst[idx + n * stride] = p_src[i + n * stride];
	} else if (fin) {
		g_final[base_f + (uint32_t)__popcll(m_fin & below)] = g;
	}
}

__global__ void march_active_rays_kernel(uint32_t n_rays_alive, uint32_t n_nerfs, const NgpGlobalRay* __restrict__ global_rays, NgpProxyRay* __restrict__ proxy_rays, uint32_t stride,
                                         const NgpNerfProps* __restrict__ props) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_rays_alive) return;
	if (!global_rays[i].alive) return;
	for (uint32_t n = 0; n < n_nerfs; ++n) {
		NgpProxyRay& p = proxy_rays[i + n * stride];
		if (!p.alive || !p.active) continue;
		const v3 origin = ld3(p.origin), dir = ld3(p.dir);
		const v3 idir = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
		float t = p.t;
		OccBrick occ;
		p.alive = hit_test_and_march(origin, dir, idir, t, props + n, &t, nullptr, occ) ? 1 : 0;
		p.t = t;
	}
}

// nerf_renderer.cu:210-262; returns the index of the NeRF whose proxy ray is nearest to the camera (it samples in this pass), or -1
__device__ __forceinline__ int32_t cull_one_ray(uint32_t i, uint32_t n_nerfs, NgpGlobalRay* __restrict__ global_rays, NgpProxyRay* __restrict__ proxy_rays, uint32_t stride, v3 cam_pos,
                                                const NgpNerfProps* __restrict__ props) {
	NgpGlobalRay& g = global_rays[i];
	float min_d2 = 0.0f;
	int32_t active_idx = -1, active_nerf = -1;
	uint32_t n_proxy_alive = 0;
	for (uint32_t n = 0; n < n_nerfs; ++n) {
		const uint32_t pi = i + n * stride;
		NgpProxyRay& p = proxy_rays[pi];
		if (!p.alive) continue;
		++n_proxy_alive;
		const v3 pw = xform_point(props[n].transform, ld3(p.origin) + ld3(p.dir) * p.t);
		const v3 dlt = pw - cam_pos;
		const float d2 = dot(dlt, dlt);
		if (d2 < min_d2 || active_idx == -1) { min_d2 = d2; active_idx = (int32_t)pi; active_nerf = (int32_t)n; }
		p.active = 0;
	}
	if (active_idx >= 0) proxy_rays[active_idx].active = 1;
	if (n_proxy_alive == 0) g.alive = 0;
	return active_nerf;
}

__global__ void cull_rays_kernel(uint32_t n_rays_alive, uint32_t n_nerfs, NgpGlobalRay* __restrict__ global_rays, NgpProxyRay* __restrict__ proxy_rays, uint32_t stride, v3 cam_pos,
                                 const NgpNerfProps* __restrict__ props, uint32_t* __restrict__ active_lists, uint32_t* __restrict__ active_counts) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	int32_t active_nerf = -1;
	if (i < n_rays_alive && global_rays[i].alive) active_nerf = cull_one_ray(i, n_nerfs, global_rays, proxy_rays, stride, cam_pos, props);
	if (!active_lists) return;
	// per-NeRF lists of the rays that sample it in this pass: one atomic per wave and NeRF present in it
	const uint32_t lane = lane_id();
	uint64_t todo = __ballot(active_nerf >= 0);
	while (todo) {
		const int leader = __ffsll((unsigned long long)todo) - 1;
		const int32_t nerf = __shfl(active_nerf, leader, 64);
		const uint64_t same = __ballot(active_nerf == nerf);
		uint32_t base = 0;
		if ((int)lane == leader) base = atomicAdd(&active_counts[nerf], (uint32_t)__popcll(same));
		base = __shfl(base, leader, 64);
		if (active_nerf == nerf) active_lists[(size_t)nerf * stride + base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = i;
		todo &= ~same;
	}
}


// `list` (optional): thread k handles ray list[k] and the network batch holds only those n_elements rays (element k + j * n_elements);
// without it thread k handles ray k as in the reference, inactive rays leaving holes that the network evaluates anyway
__global__ void multi_generate_next_inputs_kernel(uint32_t n_elements, const uint32_t* __restrict__ list, const NgpGlobalRay* __restrict__ global_rays, NgpProxyRay* __restrict__ proxy_rays,
                                                  NgpCoord* __restrict__ network_input, uint32_t n_steps, const NgpNerfProps* __restrict__ props) {
	const uint32_t k = threadIdx.x + blockIdx.x * blockDim.x;
	if (k >= n_elements) return;
	const uint32_t i = list ? list[k] : k;
	if (!global_rays[i].alive) return;
	NgpProxyRay& p = proxy_rays[i];
	if (!p.active) return;
	const v3 origin = ld3(p.origin), dir = ld3(p.dir);
	const v3 idir = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
	const Aabb train_aabb = aabb_of(props->train_aabb);
	const v3 wd = warp_direction(dir);
	float t = p.t;
	float dt = get_dt(t, props->cone_angle, props->min_cone_stepsize, props->max_cone_stepsize);
	// The reference emits sample j, then calls hit_test_and_march (march until occupied or out of the box) and steps.  Flattened into one
	// loop of single DDA iterations per ray — same per-ray sequence, same bits — so that a lane walking ~150 empty voxels to the box
	// boundary does not hold the rest of its wave at every step (see generate_next_inputs_kernel in render.hip).
	const Aabb render_aabb = aabb_of(props->render_aabb);
	const float cone = props->cone_angle, mn = props->min_cone_stepsize, mx = props->max_cone_stepsize;
	const uint8_t* __restrict__ bitfield = props->density_grid_bitfield;
	const uint32_t grid_size = props->grid_size, grid_volume = props->grid_volume, max_cascade = props->nerf_cascades - 1, n_cascades = props->nerf_cascades;
	OccBrick occ;
	uint32_t j = 0;
	bool emit = true;
	while (j < n_steps) {
		if (emit) {
			const v3 wp = aabb_relative_pos(train_aabb, origin + dir * t);
			NgpCoord c;
			c.pos[0] = wp.x; c.pos[1] = wp.y; c.pos[2] = wp.z;
			c.dt = get_warped_dt(dt, mn, n_cascades);
			c.dir[0] = wd.x; c.dir[1] = wd.y; c.dir[2] = wd.z;
			network_input[k + (size_t)j * n_elements] = c;
			emit = false;
		}
		// one iteration of hit_test_and_march (nerf_renderer.cu:148-208)
		const v3 pos = origin + dir * t;
		if (!aabb_contains(render_aabb, pos)) { p.n_steps = (uint16_t)j; return; }
		dt = get_dt(t, cone, mn, mx);
		const int mipi = get_mip_from_dt(dt, pos, grid_size, max_cascade);
		const uint32_t mip = (uint32_t)(mipi < 0 ? 0 : mipi);
		if (!bitfield || get_is_occupied(pos, bitfield, mip, grid_size, grid_volume, occ)) {
			t += dt;
			++j;
			emit = true;
		} else {
			t = get_t_advanced_to_next_voxel(t, cone, pos, dir, idir, grid_size >> mip, mn, mx);
		}
	}
	p.t = t;
	p.n_steps = (uint16_t)n_steps;
}


// ---- one launch at the head of every pass: march_active_rays + cull_global_rays_and_set_proxy_rays_active + compact_rays (nerf_renderer.cu:675-733), plus the
// per-NeRF lists of the rays that sample each NeRF in this pass and the pass's counts posted to the host.
// The reference compacts FIRST (on the alive flags the previous pass's cull left) and marches / culls the survivors; here the ray is marched and culled and THEN
// compacted, so a ray whose last proxy died leaves one pass earlier — it would have idled through that pass (no active proxy, nothing sampled), its colour is the same
// and it reaches the finished list with the same rgba.  The march may REST: after `max_skips` empty voxels the proxy keeps its t (every DDA iteration is a function of t
// alone, so pausing between two iterations changes nothing) and the whole ray sits this pass out — no cull before all its marched proxies stand at their hit points, no
// samples — instead of holding the 63 other lanes of its wave for the ~150 voxels to the far side of the box.  Same per-ray sequence of samples, same pixels.
constexpr uint32_t MULTI_MAX_LISTS = 30;   // NeRFs with a block-aggregated list counter (2 + 30 LDS words); requests with more take the unfused loop

__device__ __forceinline__ int hit_test_and_march_bounded(v3 origin, v3 dir, v3 idir, float& t_io, const NgpNerfProps* __restrict__ props, uint32_t max_skips, OccBrick& occ) {
	const Aabb render_aabb = aabb_of(props->render_aabb);
	const float cone = props->cone_angle, mn = props->min_cone_stepsize, mx = props->max_cone_stepsize;
	const uint8_t* __restrict__ bitfield = props->density_grid_bitfield;
	const uint32_t grid_size = props->grid_size, grid_volume = props->grid_volume, max_cascade = props->nerf_cascades - 1;
	float t = t_io, prev_t = t;
	uint32_t skips = 0;
	while (1) {
		const v3 pos = origin + dir * t;
		if (!aabb_contains(render_aabb, pos)) { t_io = prev_t; return 0; }   // (hit_test_and_march leaves the t before the last advance)
		const float dt = get_dt(t, cone, mn, mx);
		const int mipi = get_mip_from_dt(dt, pos, grid_size, max_cascade);
		const uint32_t mip = (uint32_t)(mipi < 0 ? 0 : mipi);
		if (!bitfield) break;
		if (get_is_occupied(pos, bitfield, mip, grid_size, grid_volume, occ)) break;
		if (max_skips && skips == max_skips) { t_io = t; return 2; }
		++skips;
		prev_t = t;
		t = get_t_advanced_to_next_voxel(t, cone, pos, dir, idir, grid_size >> mip, mn, mx);
	}
	t_io = t;
	return 1;
}

struct MultiAdvanceArgs {
	uint32_t n_prev, n_nerfs, stride, max_skips;
	NgpGlobalRay* g_src; NgpProxyRay* p_src; NgpGlobalRay* g_dst; NgpProxyRay* p_dst; NgpGlobalRay* g_final;   // (the sources are updated in place before they are copied)
	v3 cam_pos; const NgpNerfProps* props;
	uint32_t* counters;        // [0] alive rays, [1 + n] rays that sample NeRF n in this pass; all zero at launch
	uint32_t* next_counters;   // the other set: zeroed by this launch's last workgroup for the next pass
	uint32_t* final_counter;   // finished rays (runs over the whole frame)
	uint32_t* active_lists;    // [n_nerfs][stride] compacted indices
	uint32_t* blocks_done; unsigned long long* host_mailbox; uint32_t sequence;
	uint32_t tile_w, tile_h;   // > 0: thread order = 8 x 8 pixel tiles of a tile_w x tile_h image whose ray i is pixel (i % tile_w, i / tile_w) (the first pass)
};

__global__ void __launch_bounds__(256) multi_advance_kernel(const MultiAdvanceArgs a) {
	__shared__ uint32_t s_cnt[2 + MULTI_MAX_LISTS], s_base[2 + MULTI_MAX_LISTS];
	if (threadIdx.x < 2u + MULTI_MAX_LISTS) s_cnt[threadIdx.x] = 0u;
	__syncthreads();
	const uint32_t tid = threadIdx.x + blockIdx.x * blockDim.x;
	uint32_t i = tid;
	bool in = tid < a.n_prev;
	if (a.tile_w) {
		const uint32_t tpr = (a.tile_w + 7u) >> 3, tile = tid >> 6, within = tid & 63u;
		const uint32_t x = (tile % tpr) * 8u + (within & 7u), y = (tile / tpr) * 8u + (within >> 3);
		in = x < a.tile_w && y < a.tile_h;
		i = x + a.tile_w * y;
		in = in && i < a.n_prev;
	}
	NgpGlobalRay g;
	bool alive = false, fin = false;
	int32_t active_nerf = -1;
	if (in) {
		g = a.g_src[i];
		if (g.alive) {
			// march_active_rays (nerf_renderer.cu:271-314)
			bool rested = false;
			for (uint32_t n = 0; n < a.n_nerfs; ++n) {
				NgpProxyRay p = a.p_src[i + n * a.stride];
				if (p.alive && p.active) {
					const v3 origin = ld3(p.origin), dir = ld3(p.dir);
					const v3 idir = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
					float t = p.t;
					OccBrick occ;
					const int r = hit_test_and_march_bounded(origin, dir, idir, t, a.props + n, a.max_skips, occ);
					a.p_src[i + n * a.stride].t = t;
					if (r == 0) a.p_src[i + n * a.stride].alive = 0;
					rested = rested || r == 2;
				}
			}
			if (rested) alive = true;   // sits this pass out: flags untouched, the marched proxies go on next pass (one that already stands at its hit returns at once)
			else {
				active_nerf = cull_one_ray(i, a.n_nerfs, a.g_src, a.p_src, a.stride, a.cam_pos, a.props);
				alive = active_nerf >= 0;   // cull clears g.alive when no proxy is left
			}
		}
		fin = !alive && g.rgba[3] > 0.001f;
	}
	// compact_rays (nerf_renderer.cu:94-146) + the per-NeRF lists: ranks inside the workgroup by LDS atomics, one global atomic per workgroup and counter
	uint32_t rank = 0, lrank = 0;
	if (alive) rank = atomicAdd(&s_cnt[0], 1u); else if (fin) rank = atomicAdd(&s_cnt[1], 1u);
	if (active_nerf >= 0) lrank = atomicAdd(&s_cnt[2 + active_nerf], 1u);
	__syncthreads();
	if (threadIdx.x < 2u + a.n_nerfs) {
		const uint32_t c = s_cnt[threadIdx.x];
		uint32_t* ctr = threadIdx.x == 0 ? a.counters : threadIdx.x == 1 ? a.final_counter : a.counters + (threadIdx.x - 1u);
		s_base[threadIdx.x] = c ? atomicAdd(ctr, c) : 0u;
	}
	__syncthreads();
	if (alive) {
		const uint32_t idx = s_base[0] + rank;
		g.alive = 1;
		a.g_dst[idx] = g;
		for (uint32_t n = 0; n < a.n_nerfs; ++n) a.p_dst[idx + n * a.stride] = a.p_src[i + n * a.stride];
		if (active_nerf >= 0) a.active_lists[(size_t)active_nerf * a.stride + s_base[2 + active_nerf] + lrank] = idx;
	} else if (fin) {
		g.alive = 0;
		a.g_final[s_base[1] + rank] = g;
	}
	// the last workgroup posts {n_alive, n_active[0 .. n_nerfs)} to the host, one self-tagged 8-byte word each (no ordering between them needed), and clears the
	// counters of the next pass.  (Ticket after this workgroup's counter atomics have returned: see compact_store in render.hip.)
	if (a.host_mailbox) {
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t dep = 0;
			for (uint32_t k = 0; k < 2u + a.n_nerfs; ++k) dep |= s_base[k];
			uint32_t zero;
			asm volatile("v_and_b32 %0, 0, %1" : "=v"(zero) : "v"(dep));
			if (__hip_atomic_fetch_add(a.blocks_done, 1u + zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
				__hip_atomic_store(a.blocks_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				for (uint32_t k = 0; k < 1u + a.n_nerfs; ++k) {
					const uint32_t v = __hip_atomic_load(a.counters + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(a.next_counters + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(a.host_mailbox + k, (unsigned long long)v | ((unsigned long long)a.sequence << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			}
		}
	}
}

typedef uint16_t us4m __attribute__((ext_vector_type(4)));

__global__ void multi_composite_kernel(uint32_t n_global_rays, const uint32_t* __restrict__ list, uint32_t current_step, NgpGlobalRay* __restrict__ global_rays, NgpProxyRay* __restrict__ proxy_rays,
                                       const NgpCoord* __restrict__ network_input, const uint16_t* __restrict__ network_output, uint32_t out_stride, uint32_t n_steps,
                                       int rgb_activation, int density_activation, float min_transmittance, const NgpNerfProps* __restrict__ props) {
	const uint32_t k = threadIdx.x + blockIdx.x * blockDim.x;
	if (k >= n_global_rays) return;
	const uint32_t i = list ? list[k] : k;
	NgpGlobalRay& g = global_rays[i];
	if (!g.alive) return;
	NgpProxyRay& p = proxy_rays[i];
	if (!p.alive || !p.active) return;
	float r = g.rgba[0], gg = g.rgba[1], b = g.rgba[2], a = g.rgba[3];
	const Aabb train_aabb = aabb_of(props->train_aabb);
	const uint32_t actual_n_steps = p.n_steps;
	uint32_t j = 0;
	for (; j < actual_n_steps; ++j) {
		const size_t e = k + (size_t)j * n_global_rays;
		const us4m o = *(const us4m*)(network_output + e * out_stride);
		const NgpCoord& in = network_input[e];
		const v3 pos = unwarp_position(ld3(in.pos), train_aabb);
		const float T = 1.f - a;
		const float dt = get_unwarped_dt(in.dt, props->min_cone_stepsize, props->nerf_cascades);
		const float alpha = 1.f - __expf(-network_to_density(h2f(o[3]), density_activation) * dt);
		float weight = alpha * T;
		const float cr = network_to_rgb(h2f(o[0]), rgb_activation), cg = network_to_rgb(h2f(o[1]), rgb_activation), cb = network_to_rgb(h2f(o[2]), rgb_activation);
		float mask_weight = 1.f;
		for (uint32_t k = 0; k < props->n_masks; ++k) mask_weight = clampf(mask_weight + mask_sample(props->masks[k], pos), 0.0f, 1.0f);
		weight *= mask_weight;
		weight *= props->opacity;
		r += cr * weight; gg += cg * weight; b += cb * weight; a += weight;
		if (a > (1.0f - min_transmittance)) { r /= a; gg /= a; b /= a; a /= a; break; }
	}
	if (j < n_steps) { p.alive = 0; p.n_steps = (uint16_t)(j + current_step); }
	g.rgba[0] = r; g.rgba[1] = gg; g.rgba[2] = b; g.rgba[3] = a;
}

__global__ void multi_shade_kernel(uint32_t n_rays, const NgpGlobalRay* __restrict__ rays, bool train_in_linear_colors, float4* __restrict__ frame_buffer, float* __restrict__ depth_buffer,
                                   const NgpDownsampleInfo ds, bool flip_y) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_rays) return;
	const NgpGlobalRay& ray = rays[i];
	const uint32_t x = (uint32_t)ds.skip[0] * (ray.idx % (uint32_t)ds.scaled_res[0]);
	uint32_t y = (uint32_t)ds.skip[1] * (ray.idx / (uint32_t)ds.scaled_res[0]);
	if (flip_y) y = (uint32_t)ds.max_res[1] - y - 1;
	float4 tmp = make_float4(ray.rgba[0], ray.rgba[1], ray.rgba[2], ray.rgba[3]);
	if (!train_in_linear_colors) { tmp.x = srgb_to_linear(tmp.x); tmp.y = srgb_to_linear(tmp.y); tmp.z = srgb_to_linear(tmp.z); }
	for (uint32_t u = 0; u < (uint32_t)ds.skip[0]; ++u) {
		for (uint32_t v = 0; v < (uint32_t)ds.skip[1]; ++v) {
			const uint32_t idx = (x + u) + (y + v) * (uint32_t)ds.max_res[0];
			if (idx >= ds.max_pixels) continue;
			const float4 f = frame_buffer[idx];
			const float k = 1.0f - tmp.w;
			frame_buffer[idx] = make_float4(tmp.x + f.x * k, tmp.y + f.y * k, tmp.z + f.z * k, tmp.w + f.w * k);
			if (tmp.w > 0.2f) depth_buffer[idx] = ray.depth;
		}
	}
}

} // namespace ngp

using namespace ngp;

extern "C" {

int ngp_hip_multi_init_global_rays(void* stream, uint32_t sample_index, NgpGlobalRay* rays, float* depthbuffer, const NgpDownsampleInfo* ds, const NgpRenderCamera* camera) {
	if (ds->scaled_res[0] <= 0 || ds->scaled_res[1] <= 0) return 0;
	const dim3 threads(16, 8, 1), blocks(div_up((uint32_t)ds->scaled_res[0], 16u), div_up((uint32_t)ds->scaled_res[1], 8u), 1);
	hipLaunchKernelGGL(init_global_rays_kernel, blocks, threads, 0, (hipStream_t)stream, sample_index, rays, depthbuffer, *ds, *camera);
	NGP_LAUNCH_CHECK("init_global_rays_kernel");
	return 0;
}

int ngp_hip_multi_init_proxy_rays(void* stream, uint32_t n_elements, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, const NgpNerfProps* props) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(init_proxy_rays_kernel, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, global_rays, proxy_rays, props);
	NGP_LAUNCH_CHECK("init_proxy_rays_kernel");
	return 0;
}

int ngp_hip_multi_compact_rays(void* stream, uint32_t n_elements, const NgpGlobalRay* global_src, NgpGlobalRay* global_dst, const NgpProxyRay* proxy_src, NgpProxyRay* proxy_dst,
                               uint32_t n_nerfs, uint32_t stride, NgpGlobalRay* global_final, uint32_t* alive_counter, uint32_t* final_counter) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(multi_compact_rays_kernel, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, global_src, global_dst, proxy_src, proxy_dst, n_nerfs, stride,
	                   global_final, alive_counter, final_counter);
	NGP_LAUNCH_CHECK("multi_compact_rays_kernel");
	return 0;
}

int ngp_hip_multi_march_active_rays(void* stream, uint32_t n_rays_alive, uint32_t n_nerfs, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, uint32_t stride, const NgpNerfProps* props) {
	if (!n_rays_alive) return 0;
	hipLaunchKernelGGL(march_active_rays_kernel, dim3(div_up(n_rays_alive, 128)), dim3(128), 0, (hipStream_t)stream, n_rays_alive, n_nerfs, global_rays, proxy_rays, stride, props);
	NGP_LAUNCH_CHECK("march_active_rays_kernel");
	return 0;
}

int ngp_hip_multi_cull_rays(void* stream, uint32_t n_rays_alive, uint32_t n_nerfs, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, uint32_t stride, const float* cam_pos, const NgpNerfProps* props) {
	if (!n_rays_alive) return 0;
	v3 cp; cp.x = cam_pos[0]; cp.y = cam_pos[1]; cp.z = cam_pos[2];
	hipLaunchKernelGGL(cull_rays_kernel, dim3(div_up(n_rays_alive, 128)), dim3(128), 0, (hipStream_t)stream, n_rays_alive, n_nerfs, global_rays, proxy_rays, stride, cp, props, (uint32_t*)nullptr, (uint32_t*)nullptr);
	NGP_LAUNCH_CHECK("cull_rays_kernel");
	return 0;
}

int ngp_hip_multi_cull_rays_collect(void* stream, uint32_t n_rays_alive, uint32_t n_nerfs, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, uint32_t stride, const float* cam_pos, const NgpNerfProps* props,
                                    uint32_t* active_lists, uint32_t* active_counts) {
	if (!active_lists || !active_counts) { set_last_error("ngp_hip_multi_cull_rays_collect: active_lists / active_counts missing", hipErrorInvalidValue); return -1; }
	NGP_HIP_TRY(hipMemsetAsync(active_counts, 0, (size_t)n_nerfs * 4, (hipStream_t)stream));
	if (!n_rays_alive) return 0;
	v3 cp; cp.x = cam_pos[0]; cp.y = cam_pos[1]; cp.z = cam_pos[2];
	hipLaunchKernelGGL(cull_rays_kernel, dim3(div_up(n_rays_alive, 128)), dim3(128), 0, (hipStream_t)stream, n_rays_alive, n_nerfs, global_rays, proxy_rays, stride, cp, props, active_lists, active_counts);
	NGP_LAUNCH_CHECK("cull_rays_kernel");
	return 0;
}

int ngp_hip_multi_generate_next_inputs(void* stream, uint32_t n_elements, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, NgpCoord* network_input, uint32_t n_steps, const NgpNerfProps* props) {
	return ngp_hip_multi_generate_next_inputs_list(stream, n_elements, nullptr, global_rays, proxy_rays, network_input, n_steps, props);
}
int ngp_hip_multi_generate_next_inputs_list(void* stream, uint32_t n_elements, const uint32_t* list, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, NgpCoord* network_input, uint32_t n_steps,
                                            const NgpNerfProps* props) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(multi_generate_next_inputs_kernel, dim3(div_up(n_elements, 128)), dim3(128), 0, (hipStream_t)stream, n_elements, list, global_rays, proxy_rays, network_input, n_steps, props);
	NGP_LAUNCH_CHECK("multi_generate_next_inputs_kernel");
	return 0;
}

int ngp_hip_multi_composite(void* stream, uint32_t n_global_rays, uint32_t current_step, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, const NgpCoord* network_input,
                            const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance, const NgpNerfProps* props) {
	return ngp_hip_multi_composite_list(stream, n_global_rays, nullptr, current_step, global_rays, proxy_rays, network_input, network_output, out_stride, n_steps, rgb_activation, density_activation,
	                                    min_transmittance, props);
}
int ngp_hip_multi_composite_list(void* stream, uint32_t n_global_rays, const uint32_t* list, uint32_t current_step, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, const NgpCoord* network_input,
                                 const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance, const NgpNerfProps* props) {
	if (!n_global_rays) return 0;
	if (out_stride < 4 || (out_stride & 3)) { set_last_error("ngp_hip_multi_composite: out_stride must be a multiple of 4", hipErrorInvalidValue); return -1; }
	hipLaunchKernelGGL(multi_composite_kernel, dim3(div_up(n_global_rays, 128)), dim3(128), 0, (hipStream_t)stream, n_global_rays, list, current_step, global_rays, proxy_rays, network_input,
	                   network_output, out_stride, n_steps, rgb_activation, density_activation, min_transmittance, props);
	NGP_LAUNCH_CHECK("multi_composite_kernel");
	return 0;
}

int ngp_hip_multi_advance(void* stream, uint32_t n_prev, uint32_t n_nerfs, const NgpGlobalRay* global_src, NgpProxyRay* proxy_src, NgpGlobalRay* global_dst, NgpProxyRay* proxy_dst,
                          uint32_t stride, NgpGlobalRay* global_final, const float* cam_pos, const NgpNerfProps* props, uint32_t max_skips, uint32_t* counters, uint32_t* next_counters,
                          uint32_t* final_counter, uint32_t* active_lists, uint32_t* blocks_done, uint64_t* host_mailbox, uint32_t sequence, uint32_t tile_w, uint32_t tile_h) {
	if (!n_prev) return 0;
	if (n_nerfs == 0 || n_nerfs > MULTI_MAX_LISTS) { set_last_error("ngp_hip_multi_advance: between 1 and 30 NeRFs", hipErrorInvalidValue); return -1; }
	if (!counters || !final_counter || !active_lists || (host_mailbox && (!blocks_done || !next_counters))) { set_last_error("ngp_hip_multi_advance: counters / lists missing", hipErrorInvalidValue); return -1; }
	MultiAdvanceArgs a;
	a.n_prev = n_prev; a.n_nerfs = n_nerfs; a.stride = stride; a.max_skips = max_skips;
	a.g_src = const_cast<NgpGlobalRay*>(global_src); a.p_src = proxy_src; a.g_dst = global_dst; a.p_dst = proxy_dst; a.g_final = global_final;
	a.cam_pos.x = cam_pos[0]; a.cam_pos.y = cam_pos[1]; a.cam_pos.z = cam_pos[2]; a.props = props;
	a.counters = counters; a.next_counters = next_counters; a.final_counter = final_counter; a.active_lists = active_lists;
	a.blocks_done = blocks_done; a.host_mailbox = (unsigned long long*)host_mailbox; a.sequence = sequence; a.tile_w = tile_w; a.tile_h = tile_h;
	uint32_t n_threads = n_prev;
	if (tile_w) {
		if ((uint64_t)tile_w * tile_h < n_prev) { set_last_error("ngp_hip_multi_advance: tile_w x tile_h smaller than n_prev", hipErrorInvalidValue); return -1; }
		n_threads = ((tile_w + 7u) / 8u) * ((tile_h + 7u) / 8u) * 64u;
	}
	hipLaunchKernelGGL(multi_advance_kernel, dim3(div_up(n_threads, 256)), dim3(256), 0, (hipStream_t)stream, a);
	NGP_LAUNCH_CHECK("multi_advance_kernel");
	return 0;
}

int ngp_hip_multi_shade(void* stream, uint32_t n_rays, const NgpGlobalRay* rays, int train_in_linear_colors, float* frame_buffer, float* depth_buffer, const NgpDownsampleInfo* ds, int flip_y) {
	if (!n_rays) return 0;
	hipLaunchKernelGGL(multi_shade_kernel, dim3(div_up(n_rays, 128)), dim3(128), 0, (hipStream_t)stream, n_rays, rays, train_in_linear_colors != 0, (float4*)frame_buffer, depth_buffer, *ds, flip_y != 0);
	NGP_LAUNCH_CHECK("multi_shade_kernel");
	return 0;
}

} // extern "C"
