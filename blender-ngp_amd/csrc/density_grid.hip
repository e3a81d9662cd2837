// density_grid.hip — occupancy ("density") grid maintenance for gfx950.
// Replaces src/testbed_nerf.cu:369-416 (mark_untrained_density_grid), 465-494 (generate_grid_samples_nerf_nonuniform),
// 496-512 (splat), 532-555 (ema), 563-610 (grid_to_bitfield, bitfield_max_pool), 2851-2852 (mean via reduce_sum).
// All index math is exact (built -ffp-contract=off): bit-exact against the CPU oracle.
#include "ngp_device.cuh"
#include <string.h>
#include <stdio.h>

namespace ngp {

static thread_local char g_last_error[512] = "";
void set_last_error(const char* what, hipError_t e) {
	snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", what, e == hipSuccess ? "ok" : hipGetErrorString(e), (int)e);
}

__global__ void mark_untrained_kernel(uint32_t n_elements, float* __restrict__ grid_out, uint32_t n_training_images,
                                      const NgpImageMeta* __restrict__ metadata, const NgpXForm* __restrict__ xforms, bool clear_visible_voxels) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint32_t level = i / NGP_NERF_GRID_N_CELLS;
	const uint32_t pos_idx = i % NGP_NERF_GRID_N_CELLS;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float s = __uint_as_float((127u + level) << 23); // scalbnf(1, level)
	const v3 pos = mk((((float)x + 0.5f) / 128.0f - 0.5f) * s + 0.5f, (((float)y + 0.5f) / 128.0f - 0.5f) * s + 0.5f, (((float)z + 0.5f) / 128.0f - 0.5f) * s + 0.5f);
	const float voxel_radius = 0.5f * SQRT3() * s / 128.0f;
	int count = 0;
	for (uint32_t j = 0; j < n_training_images; ++j) {
		const NgpImageMeta& md = metadata[j];
		if (md.lens_mode == 2 || md.lens_mode == 3) { count++; break; }
		const float half_resx = (float)md.res[0] * 0.5f, half_resy = (float)md.res[1] * 0.5f;
		const float* xf = xforms[j].start;
		const v3 ploc = pos - col(xf, 3);
		const float px = dot(ploc, col(xf, 0)), py = dot(ploc, col(xf, 1)), pz = dot(ploc, col(xf, 2));
		if (pz > 0.f) {
			if (fabsf(px) - voxel_radius < pz / md.focal_length[0] * half_resx && fabsf(py) - voxel_radius < pz / md.focal_length[1] * half_resy) {
				count++;
				break;
			}
		}
	}
	if (clear_visible_voxels || (grid_out[i] < 0) != (count <= 0)) grid_out[i] = (count > 0) ? 0.f : -1.f;
}

__global__ void grid_samples_nonuniform_kernel(uint32_t n_elements, Pcg32 rng, uint32_t step, Aabb aabb, const float* __restrict__ grid_in,
                                               float* __restrict__ out_pos, uint32_t* __restrict__ indices, uint32_t n_cascades, float thresh) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	rng.advance((uint64_t)(i * 4u));
	const uint32_t level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
	uint32_t idx = 0;
	for (uint32_t j = 0; j < 10; ++j) {
		idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % NGP_NERF_GRID_N_CELLS;
		idx += level * NGP_NERF_GRID_N_CELLS;
		if (grid_in[idx] > thresh) break;
	}
	const uint32_t pos_idx = idx % NGP_NERF_GRID_N_CELLS;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
	const float s = __uint_as_float((127u + level) << 23);
	const v3 pos = mk((((float)x + rx) / 128.0f - 0.5f) * s + 0.5f, (((float)y + ry) / 128.0f - 0.5f) * s + 0.5f, (((float)z + rz) / 128.0f - 0.5f) * s + 0.5f);
	const v3 w = aabb_relative_pos(aabb, pos);
	out_pos[3 * i + 0] = w.x; out_pos[3 * i + 1] = w.y; out_pos[3 * i + 2] = w.z;
	indices[i] = idx;
}

// ---- the same samples, generated in MORTON ORDER of the cell their first try lands in --------------------------------------------------------------------------
// The generator above hands consecutive samples i, i + 1 cells that a multiplicative hash scatters over the whole grid, so the density pass that follows gathers
// 16 levels x 4 line pairs per sample with nothing shared between the lanes of a wave, and the splat's atomicMax hits 64 unrelated lines per instruction.  But the
// first try's map  cell_0(i) = ((i + step n) * 56924617 + 96925573) mod 2^21  is a bijection of Z / 2^21 (odd multiplier): thread c of THIS kernel stands on cell c —
// threads in Morton order, 256 consecutive cells per workgroup — solves the map for i (multiplicative inverse mod 2^21), and if i < n runs the reference's body for
// sample i unchanged (same generator skip-ahead, same ten tries, same position).  Every i in [0, n) is produced exactly once; what changes is only WHERE in the
// output it lands: the samples of a workgroup's 1024 cells take one contiguous slot range, in cell order, so neighbouring output slots hold samples of neighbouring cells.  Samples whose first
// try is accepted (the uniform half: all cells the cameras see; the non-uniform half: the occupied ones) are spatially ordered; the others land where the later
// tries send them, as before.  The multiset of (position, index) pairs is the reference's, so the grid the update produces is the same bit for bit.
// (Measured and rejected first: ordering the forward kernel's output by 4^3-cell brick with a histogram / scan / scatter of global atomics — 190 + 230 + 250 us on
// the second stream for the fox scene's 3.1 M samples, more than the ordered density pass wins back: profiles/r06_experiments.md.)
__host__ __device__ constexpr uint32_t inverse_mod_2_32(uint32_t a) {   // a odd.  Newton: x <- x (2 - a x) doubles the correct low bits (3 -> 6 -> 12 -> 24 -> 48)
	uint32_t x = a;
	x *= 2u - a * x; x *= 2u - a * x; x *= 2u - a * x; x *= 2u - a * x;
	return x;
}
static_assert(inverse_mod_2_32(56924617u) * 56924617u == 1u, "inverse of the grid sampler's multiplier");

constexpr uint32_t GSM_CELLS = 1024;                                  // cells (threads) per workgroup
constexpr uint32_t GSM_GROUPS = NGP_NERF_GRID_N_CELLS / GSM_CELLS;     // 2048 workgroups cover the 2^21 first-try cells
constexpr uint32_t GSM_MAX_PER_CELL = 8;                               // n <= 2^24 = 8 cascades x 2^21 (the full-grid updates of the first 256 steps)

// sample numbers whose first try lands on cell c: i0, i0 + 2^21, ... below n
__device__ __forceinline__ uint32_t gsm_first_sample(uint32_t c, uint32_t step, uint32_t n_elements) {
	// cell_0(i) = c  <=>  i + step * n = (c - 96925573) * 56924617^-1   (mod 2^21; the reference computes mod 2^32 and reduces, which is the same thing)
	return (((c - 96925573u) * inverse_mod_2_32(56924617u)) - step * n_elements) & (NGP_NERF_GRID_N_CELLS - 1u);
}
__device__ __forceinline__ uint32_t gsm_count(uint32_t i0, uint32_t n_elements) { return i0 < n_elements ? (n_elements - i0 + NGP_NERF_GRID_N_CELLS - 1u) / NGP_NERF_GRID_N_CELLS : 0u; }

// pass 1: how many samples each workgroup of pass 2 will write (its 1024 cells' counts summed).  No atomics anywhere: the slot of every sample is a pure function
// of (n, step), so two runs give the same buffers bit for bit.
__global__ void __launch_bounds__(GSM_CELLS) grid_samples_morton_count_kernel(uint32_t n_elements, uint32_t step, uint32_t* __restrict__ totals) {
	__shared__ uint32_t s_wave[16];
	uint32_t v = gsm_count(gsm_first_sample(blockIdx.x * GSM_CELLS + threadIdx.x, step, n_elements), n_elements);
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	if ((threadIdx.x & 63u) == 0u) s_wave[threadIdx.x >> 6] = v;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += s_wave[w]; totals[blockIdx.x] = t; }
}

__global__ void __launch_bounds__(GSM_CELLS) grid_samples_morton_kernel(uint32_t n_elements, Pcg32 rng0, uint32_t step, Aabb aabb, const float* __restrict__ grid_in,
                                                                        float* __restrict__ out_pos, uint32_t* __restrict__ indices, uint32_t n_cascades, float thresh, const uint32_t* __restrict__ totals) {
	__shared__ uint32_t s_wave[16], s_part[16];
	__shared__ uint32_t s_list[GSM_CELLS * GSM_MAX_PER_CELL];   // this workgroup's sample numbers in cell order: the body below runs on DENSE waves (1 cell in 4 has a sample when n = 2^19)
	const uint32_t c = blockIdx.x * GSM_CELLS + threadIdx.x;    // the cell of the first try; the launch covers all 2^21 of them, threads in Morton order
	const uint32_t i0 = gsm_first_sample(c, step, n_elements);
	const uint32_t mine = gsm_count(i0, n_elements);
	// first output slot of this workgroup: the totals of the workgroups in front of it (2048 words: two per thread)
	uint32_t before = 0;
	for (uint32_t w = threadIdx.x; w < blockIdx.x; w += GSM_CELLS) before += totals[w];
	for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
	// exclusive scan of `mine` over the workgroup
	uint32_t incl = mine;
	for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if ((int)(threadIdx.x & 63u) >= off) incl += v; }
	if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = incl;
	if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = before;
	__syncthreads();
	uint32_t local = incl - mine, total = 0, base = 0;
	for (uint32_t w = 0; w < 16; ++w) { if (w < (threadIdx.x >> 6)) local += s_wave[w]; total += s_wave[w]; base += s_part[w]; }
	for (uint32_t k = 0; k < mine; ++k) s_list[local + k] = i0 + k * NGP_NERF_GRID_N_CELLS;
	__syncthreads();
	for (uint32_t e = threadIdx.x; e < total; e += GSM_CELLS) {
		const uint32_t i = s_list[e], slot = base + e;
		Pcg32 rng = rng0;
		rng.advance((uint64_t)(i * 4u));
		const uint32_t level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
		uint32_t idx = 0;
		for (uint32_t j = 0; j < 10; ++j) {
			idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % NGP_NERF_GRID_N_CELLS;
			idx += level * NGP_NERF_GRID_N_CELLS;
			if (grid_in[idx] > thresh) break;
		}
		const uint32_t pos_idx = idx % NGP_NERF_GRID_N_CELLS;
		const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
		const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
		const float s = __uint_as_float((127u + level) << 23);
		const v3 pos = mk((((float)x + rx) / 128.0f - 0.5f) * s + 0.5f, (((float)y + ry) / 128.0f - 0.5f) * s + 0.5f, (((float)z + rz) / 128.0f - 0.5f) * s + 0.5f);
		const v3 w = aabb_relative_pos(aabb, pos);
		out_pos[3 * slot + 0] = w.x; out_pos[3 * slot + 1] = w.y; out_pos[3 * slot + 2] = w.z;
		indices[slot] = idx;
	}
}

__global__ void splat_max_kernel(uint32_t n_elements, const uint32_t* __restrict__ indices, const uint16_t* __restrict__ network_output,
                                 float* __restrict__ grid_out, int density_activation) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const float mlp = network_to_density(h2f(network_output[i]), density_activation);
	const float optical_thickness = mlp * MIN_CONE_STEPSIZE();
	// positive floats order like their bit patterns: uint atomicMax (testbed_nerf.cu:509-511)
	atomicMax((uint32_t*)&grid_out[indices[i]], __float_as_uint(optical_thickness));
}

__global__ void ema_kernel(uint32_t n_elements, float decay, float* __restrict__ grid_out, const float* __restrict__ grid_in) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const float prev_val = grid_out[i];
	grid_out[i] = (prev_val < 0.f) ? prev_val : fmaxf(prev_val * decay, grid_in[i]);
}

// sum_i f(in[i]) with f = max(v,0)/n (MODE 0) or identity (MODE 1); wave shuffle -> LDS -> one atomicAdd per block
template <int MODE>
__global__ void __launch_bounds__(256) reduce_sum_kernel(const float* __restrict__ in, uint32_t n, float scale, float* __restrict__ out) {
	__shared__ float partial[4];
	float acc = 0.0f;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const float v = in[i];
		acc += MODE == 0 ? fmaxf(v, 0.f) * scale : v;
	}
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
	if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(out, partial[0] + partial[1] + partial[2] + partial[3]);
}

__global__ void grid_to_bitfield_kernel(uint32_t n_elements, uint32_t n_nonzero_elements, const float* __restrict__ grid,
                                        uint8_t* __restrict__ bitfield, const float* __restrict__ mean_density_ptr) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	if (i >= n_nonzero_elements) { bitfield[i] = 0; return; }
	const float thresh = fminf(MIN_OPTICAL_THICKNESS(), *mean_density_ptr);
	const float4 a = ((const float4*)grid)[2 * i], b = ((const float4*)grid)[2 * i + 1];
	uint8_t bits = 0;
	bits |= a.x > thresh ? 1 : 0; bits |= a.y > thresh ? 2 : 0; bits |= a.z > thresh ? 4 : 0; bits |= a.w > thresh ? 8 : 0;
	bits |= b.x > thresh ? 16 : 0; bits |= b.y > thresh ? 32 : 0; bits |= b.z > thresh ? 64 : 0; bits |= b.w > thresh ? 128 : 0;
	bitfield[i] = bits;
}

// testbed_nerf.cu:589-610.  Thread i reads 8 bytes of the finer level (64 cells = 8 coarse cells) and produces ONE byte of the coarser
// level: bit j is set if fine byte j has any cell set.  The target is the byte whose (x, y, z) byte-coordinates are those of i shifted
// by 16 (the finer cascade covers the central half of the coarser one); targets are distinct across threads, so the `|=` is race-free.
__global__ void bitfield_max_pool_kernel(uint32_t n_elements, const uint8_t* __restrict__ prev_level, uint8_t* __restrict__ next_level) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint64_t eight = ((const uint64_t*)prev_level)[i];
	uint8_t bits = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) bits |= ((eight >> (8 * j)) & 0xffu) ? (uint8_t)(1u << j) : 0;
	const uint32_t x = morton3D_invert(i >> 0) + 16, y = morton3D_invert(i >> 1) + 16, z = morton3D_invert(i >> 2) + 16;
	next_level[morton3D(x, y, z)] |= bits;
}

// The levels above the last cascade that has cells of its own are pure max-pools of it (own bits = 0: grid_to_bitfield zeroed them): one launch builds them all instead
// of one launch per level (seven ~4.7 us launches in stream order for a one-cascade scene, each over 32768 threads that mostly read zeros).  A workgroup takes 256
// pooling units (2048 bytes) of the source level and follows them up: 256 bytes of the next level, 32 of the one above, 4 of the third — whole bytes that no other
// workgroup touches — and from the fourth level on single bits, OR-ed into the (zeroed) target with a word atomic.  f() is the reference's target map (:600-609):
// unit i = 8 bytes = a 2 x 2 x 2 block of bytes at byte coordinates morton3D_invert(i); the target byte sits at those coordinates + 16 in the next level.
__device__ __forceinline__ uint32_t pool_target(uint32_t unit) { return morton3D(morton3D_invert(unit >> 0) + 16u, morton3D_invert(unit >> 1) + 16u, morton3D_invert(unit >> 2) + 16u); }

__global__ void __launch_bounds__(256) bitfield_pool_levels_kernel(const uint8_t* __restrict__ src_level, uint8_t* __restrict__ next_levels, uint32_t n_levels) {
	__shared__ uint8_t s1[256], s2[32], s3[4];
	constexpr uint32_t LEVEL_BYTES = NGP_NERF_GRID_N_CELLS / 8u;
	const uint32_t unit0 = blockIdx.x * 256u + threadIdx.x;
	const uint64_t eight = ((const uint64_t*)src_level)[unit0];
	uint8_t bits = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) bits |= ((eight >> (8 * j)) & 0xffu) ? (uint8_t)(1u << j) : 0;
	const uint32_t t1 = pool_target(unit0);
	next_levels[t1] = bits;
	if (n_levels == 1u) return;
	s1[threadIdx.x] = bits;   // (t1 = t1 of the workgroup's first unit + threadIdx.x: the + 16 only rewrites coordinate bit 4, i.e. Morton bits >= 12, and a workgroup's units share those)
	__syncthreads();
	const uint32_t t1_base = pool_target(blockIdx.x * 256u);
	uint32_t t2_base = 0;
	if (threadIdx.x < 32u) {
		uint8_t b = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) b |= s1[8 * threadIdx.x + j] ? (uint8_t)(1u << j) : 0;
		next_levels[LEVEL_BYTES + pool_target((t1_base >> 3) + threadIdx.x)] = b;
		s2[threadIdx.x] = b;
	}
	if (n_levels == 2u) return;
	__syncthreads();
	t2_base = pool_target(t1_base >> 3);
	if (threadIdx.x < 4u) {
		uint8_t b = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) b |= s2[8 * threadIdx.x + j] ? (uint8_t)(1u << j) : 0;
		next_levels[2u * LEVEL_BYTES + pool_target((t2_base >> 3) + threadIdx.x)] = b;
		s3[threadIdx.x] = b;
	}
	if (n_levels == 3u) return;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t_prev = pool_target(t2_base >> 3);   // this workgroup's first byte of the third level: bytes t_prev .. t_prev + 3, half a pooling unit
		uint32_t contribution = 0;
		for (uint32_t v = 0; v < 4u; ++v) if (s3[v]) contribution |= 1u << ((t_prev & 7u) + v);
		for (uint32_t level = 3u; level < n_levels && contribution; ++level) {
			const uint32_t t = pool_target(t_prev >> 3);
			uint8_t* byte = next_levels + (size_t)level * LEVEL_BYTES + t;
			atomicOr((uint32_t*)((uintptr_t)byte & ~(uintptr_t)3), contribution << (8u * (uint32_t)((uintptr_t)byte & 3u)));
			contribution = 1u << (t & 7u);   // towards the level above: this byte is non-zero, it is byte (t & 7) of its unit
			t_prev = t;
		}
	}
}

// The tail of an update — ema (:532-555), mean of cascade 0 (:2851-2852: reduce_sum of max(v, 0) / n), bitfield (:563-587) — in TWO passes over the grid instead of a
// memset and three: the ema pass leaves one partial sum of the mean per workgroup, and every workgroup of the bitfield pass adds the <= 1024 partials up itself, in one
// fixed order (so the mean is the same in every workgroup, and the same run to run — which the block-wise atomicAdd of reduce_sum_kernel is not); workgroup 0 also
// stores it for the readers of density_grid_mean.  No atomics: a ticket that elects a last workgroup costs ~30 ns per workgroup on one address, 60-200 us here (measured).
constexpr uint32_t EMA_MAX_BLOCKS = 1024;
__global__ void __launch_bounds__(256) ema_partial_mean_kernel(uint32_t n_elements, float decay, float* __restrict__ grid_out, const float* __restrict__ grid_in, uint32_t n_mean, float scale,
                                                               float* __restrict__ partials) {
	__shared__ float s_part[4];
	float acc = 0.0f;
	for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 4u; i < n_elements; i += gridDim.x * 1024u) {   // four cells per thread and trip (n_elements is a multiple of 2^21)
		const float4 prev = *(const float4*)(grid_out + i), in = *(const float4*)(grid_in + i);
		float4 v;
		v.x = (prev.x < 0.f) ? prev.x : fmaxf(prev.x * decay, in.x); v.y = (prev.y < 0.f) ? prev.y : fmaxf(prev.y * decay, in.y);
		v.z = (prev.z < 0.f) ? prev.z : fmaxf(prev.z * decay, in.z); v.w = (prev.w < 0.f) ? prev.w : fmaxf(prev.w * decay, in.w);
		*(float4*)(grid_out + i) = v;
		if (i < n_mean) acc += ((fmaxf(v.x, 0.f) * scale + fmaxf(v.y, 0.f) * scale) + fmaxf(v.z, 0.f) * scale) + fmaxf(v.w, 0.f) * scale;
	}
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
	if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) partials[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ void __launch_bounds__(256) grid_to_bitfield_mean_kernel(uint32_t n_elements, uint32_t n_nonzero_elements, const float* __restrict__ grid, uint8_t* __restrict__ bitfield,
                                                                    const float* __restrict__ partials, uint32_t n_partials, float* __restrict__ mean_out) {
	__shared__ float s_part[4];
	float sum = 0.0f;
	for (uint32_t b = threadIdx.x; b < n_partials; b += 256u) sum += partials[b];
	for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
	if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = sum;
	__syncthreads();
	const float mean = s_part[0] + s_part[1] + s_part[2] + s_part[3];
	if (blockIdx.x == 0 && threadIdx.x == 0) *mean_out = mean;
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	if (i >= n_nonzero_elements) { bitfield[i] = 0; return; }
	const float thresh = fminf(MIN_OPTICAL_THICKNESS(), mean);
	const float4 a = ((const float4*)grid)[2 * i], b = ((const float4*)grid)[2 * i + 1];
	uint8_t bits = 0;
	bits |= a.x > thresh ? 1 : 0; bits |= a.y > thresh ? 2 : 0; bits |= a.z > thresh ? 4 : 0; bits |= a.w > thresh ? 8 : 0;
	bits |= b.x > thresh ? 16 : 0; bits |= b.y > thresh ? 32 : 0; bits |= b.z > thresh ? 64 : 0; bits |= b.w > thresh ? 128 : 0;
	bitfield[i] = bits;
}

} // namespace ngp

using namespace ngp;

extern "C" {

int ngp_hip_abi_version(void) { return NGP_HIP_ABI_VERSION; }
const char* ngp_hip_last_error(void) { return g_last_error; }

int ngp_hip_mark_untrained_density_grid(void* stream, uint32_t n_elements, float* grid_out, uint32_t n_training_images,
                                        const NgpImageMeta* metadata, const NgpXForm* xforms, int clear_visible_voxels) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(mark_untrained_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, grid_out, n_training_images, metadata, xforms, clear_visible_voxels != 0);
	NGP_LAUNCH_CHECK("mark_untrained_kernel");
	return 0;
}

int ngp_hip_generate_grid_samples_nonuniform(void* stream, uint32_t n_elements, uint64_t rng_state, uint64_t rng_inc, uint32_t step,
                                             const NgpAabb* aabb_host, const float* grid_in, float* out_pos, uint32_t* indices, uint32_t n_cascades, float thresh) {
	if (!n_elements) return 0;
	Pcg32 rng; rng.state = rng_state; rng.inc = rng_inc;
	hipLaunchKernelGGL(grid_samples_nonuniform_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, rng, step, aabb_from_host(aabb_host), grid_in, out_pos, indices, n_cascades, thresh);
	NGP_LAUNCH_CHECK("grid_samples_nonuniform_kernel");
	return 0;
}

// the samples of ngp_hip_generate_grid_samples_nonuniform — the same (position, index) pairs — in another ORDER: slots follow the Morton order of the cell each
// sample's first try lands in (kernel comment above); deterministic.  `workspace`: ngp_hip_generate_grid_samples_morton_workspace_bytes() of device scratch.
uint64_t ngp_hip_generate_grid_samples_morton_workspace_bytes(void) { return (uint64_t)GSM_GROUPS * 4u; }
int ngp_hip_generate_grid_samples_morton(void* stream, uint32_t n_elements, uint64_t rng_state, uint64_t rng_inc, uint32_t step, const NgpAabb* aabb_host, const float* grid_in,
                                         float* out_pos, uint32_t* indices, uint32_t n_cascades, float thresh, uint32_t* workspace) {
	if (!n_elements) return 0;
	if (!workspace || n_elements > GSM_MAX_PER_CELL * NGP_NERF_GRID_N_CELLS) { set_last_error("ngp_hip_generate_grid_samples_morton: no workspace, or more than 2^24 samples", hipErrorInvalidValue); return -1; }
	Pcg32 rng; rng.state = rng_state; rng.inc = rng_inc;
	hipLaunchKernelGGL(grid_samples_morton_count_kernel, dim3(GSM_GROUPS), dim3(GSM_CELLS), 0, (hipStream_t)stream, n_elements, step, workspace);
	NGP_LAUNCH_CHECK("grid_samples_morton_count_kernel");
	hipLaunchKernelGGL(grid_samples_morton_kernel, dim3(GSM_GROUPS), dim3(GSM_CELLS), 0, (hipStream_t)stream, n_elements, rng, step, aabb_from_host(aabb_host), grid_in, out_pos, indices, n_cascades, thresh, workspace);
	NGP_LAUNCH_CHECK("grid_samples_morton_kernel");
	return 0;
}

int ngp_hip_splat_grid_samples_max(void* stream, uint32_t n_elements, const uint32_t* indices, const uint16_t* network_output, float* grid_out, int density_activation) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(splat_max_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, indices, network_output, grid_out, density_activation);
	NGP_LAUNCH_CHECK("splat_max_kernel");
	return 0;
}

int ngp_hip_ema_grid_samples(void* stream, uint32_t n_elements, float decay, float* grid_out, const float* grid_in) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(ema_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, decay, grid_out, grid_in);
	NGP_LAUNCH_CHECK("ema_kernel");
	return 0;
}

int ngp_hip_density_grid_mean(void* stream, const float* grid, uint32_t n_elements, float* mean_out) {
	NGP_HIP_TRY(hipMemsetAsync(mean_out, 0, sizeof(float), (hipStream_t)stream));
	uint32_t blocks = div_up(n_elements, 256 * 16); blocks = blocks > 1024 ? 1024 : (blocks ? blocks : 1);
	hipLaunchKernelGGL(reduce_sum_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grid, n_elements, 1.0f / (float)n_elements, mean_out);
	NGP_LAUNCH_CHECK("reduce_sum_kernel<0>");
	return 0;
}

int ngp_hip_reduce_sum_f32(void* stream, const float* in, uint32_t n, float* out) {
	NGP_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream));
	if (!n) return 0;
	uint32_t blocks = div_up(n, 256 * 16); blocks = blocks > 1024 ? 1024 : (blocks ? blocks : 1);
	hipLaunchKernelGGL(reduce_sum_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, n, 1.0f, out);
	NGP_LAUNCH_CHECK("reduce_sum_kernel<1>");
	return 0;
}

__global__ void gather_words_kernel(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* dst) {
	if (threadIdx.x == 0) {
		dst[0] = a ? *a : 0u; dst[1] = b ? *b : 0u; dst[2] = c ? *c : 0u; dst[3] = d ? *d : 0u;
		__threadfence_system();
	}
}

__global__ void post_words_kernel(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst, uint32_t* zero_words, uint32_t n_zero_words, double* sum3_dev) {
	if (threadIdx.x == 0) {
		const uint32_t va = a ? *a : 0u, vb = b ? *b : 0u, vc = c ? *c : 0u;
		dst[0] = va; dst[1] = vb; dst[2] = vc;
		if (sum3_dev) { sum3_dev[0] = (double)va; sum3_dev[1] = (double)vb; sum3_dev[2] = (double)__uint_as_float(vc); }
		for (uint32_t k = 0; k < n_zero_words; ++k) zero_words[k] = 0u;
		__threadfence_system();
		__hip_atomic_store(&dst[3], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

int ngp_hip_post_words(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst4, uint32_t* zero_words, uint32_t n_zero_words, double* sum3_dev) {
	hipLaunchKernelGGL(post_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, tag, dst4, zero_words, zero_words ? n_zero_words : 0u, sum3_dev);
	NGP_LAUNCH_CHECK("post_words_kernel");
	return 0;
}

int ngp_hip_gather_words(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* dst4) {
	hipLaunchKernelGGL(gather_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, d, dst4);
	NGP_LAUNCH_CHECK("gather_words_kernel");
	return 0;
}

static int pool_levels(void* stream, uint32_t n_cascades_used, uint8_t* bitfield);

int ngp_hip_grid_to_bitfield_and_pool(void* stream, const float* grid, uint32_t n_cascades_used, const float* mean_density, uint8_t* bitfield) {
	const uint32_t n = NGP_NERF_GRID_N_CELLS;
	hipLaunchKernelGGL(grid_to_bitfield_kernel, dim3(div_up(n / 8 * NGP_NERF_CASCADES, 256)), dim3(256), 0, (hipStream_t)stream, n / 8 * NGP_NERF_CASCADES, n / 8 * n_cascades_used, grid, bitfield, mean_density);
	NGP_LAUNCH_CHECK("grid_to_bitfield_kernel");
	return pool_levels(stream, n_cascades_used, bitfield);
}

// The tail of update_density_grid_nerf in one call (:2838 ema_grid_samples_nerf, then update_density_grid_mean_and_bitfield :2844-2859): grid_out = ema(grid_out, grid_in)
// over n_cascades_used x 2^21 cells, *mean_out = mean of max(grid_out, 0) over cascade 0, bitfield + pooled levels from them — two passes over the grid and the pooling
// launches, no memset, no atomics; the mean is summed in a fixed order.  workspace: ngp_hip_density_grid_tail_workspace_bytes() of scratch.
uint64_t ngp_hip_density_grid_tail_workspace_bytes(void) { return (uint64_t)EMA_MAX_BLOCKS * 4u; }
int ngp_hip_density_grid_ema_mean_bitfield(void* stream, uint32_t n_cascades_used, float decay, float* grid_out, const float* grid_in, float* mean_out, uint8_t* bitfield, void* workspace) {
	if (!grid_out || !grid_in || !mean_out || !bitfield || !workspace || n_cascades_used < 1u || n_cascades_used > NGP_NERF_CASCADES) {
		set_last_error("ngp_hip_density_grid_ema_mean_bitfield: null buffer or a cascade count outside 1 .. 8", hipErrorInvalidValue); return -1;
	}
	const uint32_t n = NGP_NERF_GRID_N_CELLS, n_elements = n * n_cascades_used;
	uint32_t blocks = n_elements / 1024u; blocks = blocks > EMA_MAX_BLOCKS ? EMA_MAX_BLOCKS : blocks;
	hipLaunchKernelGGL(ema_partial_mean_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n_elements, decay, grid_out, grid_in, n, 1.0f / (float)n, (float*)workspace);
	NGP_LAUNCH_CHECK("ema_partial_mean_kernel");
	hipLaunchKernelGGL(grid_to_bitfield_mean_kernel, dim3(div_up(n / 8 * NGP_NERF_CASCADES, 256)), dim3(256), 0, (hipStream_t)stream, n / 8 * NGP_NERF_CASCADES, n / 8 * n_cascades_used, grid_out, bitfield,
	                   (const float*)workspace, blocks, mean_out);
	NGP_LAUNCH_CHECK("grid_to_bitfield_mean_kernel");
	return pool_levels(stream, n_cascades_used, bitfield);
}

static int pool_levels(void* stream, uint32_t n_cascades_used, uint8_t* bitfield) {
	const uint32_t n = NGP_NERF_GRID_N_CELLS;
	// levels that have cells of their own take the finer level's pool on top of them, one launch each (:2856-2858) ...
	const uint32_t used = n_cascades_used < 1u ? 1u : (n_cascades_used > NGP_NERF_CASCADES ? NGP_NERF_CASCADES : n_cascades_used);
	for (uint32_t level = 1; level < used; ++level) {
		hipLaunchKernelGGL(bitfield_max_pool_kernel, dim3(div_up(n / 64, 256)), dim3(256), 0, (hipStream_t)stream, n / 64, bitfield + (size_t)(n / 8) * (level - 1), bitfield + (size_t)(n / 8) * level);
		NGP_LAUNCH_CHECK("bitfield_max_pool_kernel");
	}
	// ... the levels above the last such cascade are pools of pools of it: all of them in one launch
	if (used < NGP_NERF_CASCADES) {
		hipLaunchKernelGGL(bitfield_pool_levels_kernel, dim3(n / 64 / 256), dim3(256), 0, (hipStream_t)stream, bitfield + (size_t)(n / 8) * (used - 1), bitfield + (size_t)(n / 8) * used, NGP_NERF_CASCADES - used);
		NGP_LAUNCH_CHECK("bitfield_pool_levels_kernel");
	}
	return 0;
}

} // extern "C"
